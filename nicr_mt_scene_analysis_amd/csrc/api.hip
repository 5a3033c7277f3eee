// api.hip — version / error strings of libnmsa_hip.so.
#include "nmsa_common.hpp"

namespace nmsa {
thread_local int g_last_hip_error = 0;
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
