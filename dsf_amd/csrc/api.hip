#include "common.h"

extern "C" int dsf_abi_version(void) { return 5; }      // 2: dsf_mano_forward needs `save`, dsf_mano_backward takes `scratch`; 3: the exports of rounds 5-6 (.._plan, .._wrw_bias, .._pair, ..); 4: dsf_offset2joint_*_cl; 5: dsf_bn_relu_pool_*, dsf_conv_c1_forward_bn_acc, dsf_conv_c1_wrw_bn, dsf_cat_channels_nhwc

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

namespace {
template <typename T>
__global__ __launch_bounds__(256) void zero_kernel(T* __restrict__ p, size_t n) {
    T z; __builtin_memset(&z, 0, sizeof(T));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = z;
}
}  // namespace

hipError_t dsf_zero_async(void* ptr, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    if (!ptr) return hipErrorInvalidValue;
    auto grid = [](size_t n) { const size_t g = (n + 255) / 256; return dim3((unsigned)(g < 4096 ? g : 4096)); };
    if ((((uintptr_t)ptr | bytes) & 15) == 0)
        hipLaunchKernelGGL(zero_kernel<uint4>, grid(bytes / 16), dim3(256), 0, stream, (uint4*)ptr, bytes / 16);
    else if ((((uintptr_t)ptr | bytes) & 3) == 0)
        hipLaunchKernelGGL(zero_kernel<uint32_t>, grid(bytes / 4), dim3(256), 0, stream, (uint32_t*)ptr, bytes / 4);
    else
        hipLaunchKernelGGL(zero_kernel<uint8_t>, grid(bytes), dim3(256), 0, stream, (uint8_t*)ptr, bytes);
    return hipGetLastError();
}
