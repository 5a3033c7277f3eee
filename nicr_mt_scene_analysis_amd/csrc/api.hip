// api.hip — version / error strings of libnmsa_hip.so.
#include "nmsa_common.hpp"

#include <map>
#include <mutex>
#include <utility>

namespace nmsa {
thread_local int g_last_hip_error = 0;

int allow_dynamic_lds_impl(const void* kernel, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> granted;       // bytes; 0 = denied
    int dev = 0;
    if (check_hip(hipGetDevice(&dev))) return NMSA_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(kernel, dev);
    const auto it = granted.find(key);
    if (it != granted.end() && it->second >= bytes) return NMSA_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)bytes);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        (void)hipGetLastError();                    // do not poison the next launch check
        return NMSA_ERR_LAUNCH;
    }
    granted[key] = bytes;
    return NMSA_OK;
}
}  // namespace nmsa

extern "C" int nmsa_version(void) { return 100; /* 0.1.0 */ }

extern "C" int nmsa_last_hip_error(void) { return nmsa::g_last_hip_error; }

extern "C" const char* nmsa_strerror(int code)
{
    switch (code) {
        case NMSA_OK: return "ok";
        case NMSA_ERR_ARG: return "invalid argument";
        case NMSA_ERR_LAUNCH: return "HIP runtime / kernel launch error";
        case NMSA_ERR_WORKSPACE: return "workspace too small";
        case NMSA_ERR_UNSUPPORTED: return "not supported by the HIP path";
        default: return "unknown error";
    }
}
