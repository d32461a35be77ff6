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

// ---- deterministic mode --------------------------------------------------------------------------------------------
// Off: backward accumulations use float atomics (LDS first, one global flush per workgroup) and the convolutions may split
// their reduction over workgroups that meet in the output by float atomics -- results agree to ~1e-7 relative, not bit for
// bit.  On: fixed-point accumulators in the geometry backward kernels (common.h Acc<true>), no split-K in the forward-type
// convolutions, backward-weights through per-split partial tiles summed in a fixed order.
#include <stdlib.h>
static int g_deterministic = [] { const char* e = getenv("DSF_DETERMINISTIC"); return (e && atoi(e) != 0) ? 1 : 0; }();
int dsf_deterministic() { return g_deterministic; }
extern "C" int dsf_set_deterministic(int on) { const int old = g_deterministic; g_deterministic = on ? 1 : 0; return old; }
extern "C" int dsf_get_deterministic(void) { return g_deterministic; }
