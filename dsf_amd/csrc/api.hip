#include "common.h"

extern "C" int dsf_abi_version(void) { return 1; }

extern "C" const char* dsf_status_string(int s) {
    switch (s) {
        case DSF_OK: return "ok";
        case DSF_ERR_INVALID_ARG: return "invalid argument";
        case DSF_ERR_UNSUPPORTED: return "unsupported configuration";
        case DSF_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}
