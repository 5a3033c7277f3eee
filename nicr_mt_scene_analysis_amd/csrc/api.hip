// api.hip — version / error strings, per-device launch geometry and LDS grants of libnmsa_hip.so.
#include "nmsa_common.hpp"

#include <stdlib.h>

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

namespace {
DeviceGeometry query_geometry(int dev)
{
    DeviceGeometry g{256, 8, (size_t)160 * 1024, (size_t)160 * 1024};        // MI355X (SPX mode)
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) g.cus = v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess && v > 0) g.xcds = v;
    else g.xcds = g.cus >= 32 ? g.cus / 32 : 1;                              // 32 CUs per XCD on gfx950
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && v > 0)
        g.lds_per_cu = (size_t)v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0)
        g.lds_per_block = (size_t)v;
    if (g.lds_per_block > g.lds_per_cu) g.lds_per_block = g.lds_per_cu;
    (void)hipGetLastError();
    return g;
}
}  // namespace

DeviceGeometry device_geometry()
{
    static std::mutex mu;
    static std::map<int, DeviceGeometry> known;
    DeviceGeometry g{256, 8, (size_t)160 * 1024, (size_t)160 * 1024};
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        std::lock_guard<std::mutex> lock(mu);
        const auto it = known.find(dev);
        if (it != known.end()) g = it->second;
        else g = known[dev] = query_geometry(dev);
    } else {
        (void)hipGetLastError();                    // no device (CPU-only symbol / workspace checks)
    }
    const char* e = getenv("NMSA_ASSUME_CUS");      // read at every call: tests switch it
    if (e && atoi(e) > 0) g.cus = atoi(e);
    e = getenv("NMSA_ASSUME_XCDS");
    if (e && atoi(e) > 0) g.xcds = atoi(e);
    if (g.xcds > g.cus) g.xcds = g.cus;
    return g;
}
}  // namespace nmsa

extern "C" int nmsa_device_geometry(int* cus, int* xcds, size_t* lds_per_cu)
{
    const nmsa::DeviceGeometry g = nmsa::device_geometry();
    if (cus) *cus = g.cus;
    if (xcds) *xcds = g.xcds;
    if (lds_per_cu) *lds_per_cu = g.lds_per_cu;
    return NMSA_OK;
}

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
