// Library-level entry points: ABI version, error strings, HIP error latch.
#include <string.h>

#include "common.h"

namespace npcd {
static thread_local char g_hip_error[256] = "";
void set_hip_error(hipError_t e) {
    strncpy(g_hip_error, hipGetErrorString(e), sizeof(g_hip_error) - 1);
    g_hip_error[sizeof(g_hip_error) - 1] = 0;
}
}  // namespace npcd

extern "C" int npcd_abi_version(void) { return 9; }

extern "C" const char* npcd_error_string(int code) {
    switch (code) {
        case NPCD_OK: return "ok";
        case NPCD_ERR_ARG: return "invalid argument";
        case NPCD_ERR_UNSUPPORTED: return "unsupported shape or dtype";
        case NPCD_ERR_HIP: return "HIP runtime error";
        default: return "unknown error";
    }
}

extern "C" const char* npcd_last_hip_error(void) { return npcd::g_hip_error; }
