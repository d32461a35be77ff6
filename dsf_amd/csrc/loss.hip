// K10: Huber loss with delta (the reference's SmoothL1Loss, metric/losses.py:6-30) as one fused reduction and one
// fused backward instead of ~15 elementwise / reduce kernels over the (B,84,64,64) offset maps:
//   loss = scale * sum_i h(x_i - y_i),  h(z) = 0.5 z^2 (|z| < delta)  else  delta (|z| - delta / 2)
//   dloss/dx_i = g * scale * (z (|z| < delta) else delta sign(z))
// (mean over the last dim followed by mean / sum over the rest is one scale factor because every row has the same
// length).  x and y are walked in MEMORY order, so any pair of tensors with identical dense strides qualifies
// (NCHW or channels-last).  The reduction is deterministic: fixed block partials, summed in a fixed order.
#include "common.h"

namespace {

constexpr int MAX_PARTIALS = 1024;

__device__ __forceinline__ float huber(float z, float d) {
    const float a = fabsf(z);
    return a < d ? (0.5f * z) * z : d * (a - 0.5f * d);
}
__device__ __forceinline__ float huber_grad(float z, float d) {
    return fabsf(z) < d ? z : copysignf(d, z);
}

__device__ __forceinline__ float block_total(float v, float* s) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ __launch_bounds__(256) void huber_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            int64_t n, float delta, float scale, int vec,
                                                            float* __restrict__ out) {
    __shared__ float s[4];
    float acc = 0.f;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (int64_t)gridDim.x * 256;
    if (vec) {
        const int64_t n4 = n >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* y4 = reinterpret_cast<const float4*>(y);
        for (int64_t i = tid; i < n4; i += nthreads) {
            const float4 a = x4[i], b = y4[i];
            acc += (huber(a.x - b.x, delta) + huber(a.y - b.y, delta)) + (huber(a.z - b.z, delta) + huber(a.w - b.w, delta));
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nthreads) acc += huber(x[i] - y[i], delta);
    } else {
        for (int64_t i = tid; i < n; i += nthreads) acc += huber(x[i] - y[i], delta);
    }
    acc = block_total(acc, s);
    if (threadIdx.x == 0) out[blockIdx.x] = gridDim.x == 1 ? acc * scale : acc;
}

__global__ __launch_bounds__(256) void huber_final_kernel(const float* __restrict__ partials, int np, float scale,
                                                          float* __restrict__ loss) {
    __shared__ float s[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < np; i += 256) acc += partials[i];
    acc = block_total(acc, s);
    if (threadIdx.x == 0) loss[0] = acc * scale;
}

__global__ __launch_bounds__(256) void huber_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ g, int64_t n, float delta, float scale,
                                                        int vec, float* __restrict__ gx) {
    const float k = g[0] * scale;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (int64_t)gridDim.x * 256;
    if (vec) {
        const int64_t n4 = n >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* y4 = reinterpret_cast<const float4*>(y);
        float4* g4 = reinterpret_cast<float4*>(gx);
        for (int64_t i = tid; i < n4; i += nthreads) {
            const float4 a = x4[i], b = y4[i];
            g4[i] = make_float4(k * huber_grad(a.x - b.x, delta), k * huber_grad(a.y - b.y, delta),
                                k * huber_grad(a.z - b.z, delta), k * huber_grad(a.w - b.w, delta));
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nthreads) gx[i] = k * huber_grad(x[i] - y[i], delta);
    } else {
        for (int64_t i = tid; i < n; i += nthreads) gx[i] = k * huber_grad(x[i] - y[i], delta);
    }
}

inline int aligned16(const void* a, const void* b, const void* c) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

}  // namespace

extern "C" int dsf_huber_mean_forward(const float* x, const float* y, int64_t n, float delta, float scale, float* loss,
                                      float* workspace, dsf_stream_t stream) {
    DSF_CHECK_ARG(loss && n >= 0 && delta > 0.f);
    if (n == 0) return dsf_zero_async(loss, sizeof(float), (hipStream_t)stream) == hipSuccess ? DSF_OK : DSF_ERR_LAUNCH;
    DSF_CHECK_ARG(x && y);
    int blocks = (int)((n + 4095) / 4096);                     // >= 16 elements per thread
    if (blocks > MAX_PARTIALS) blocks = MAX_PARTIALS;
    DSF_CHECK_ARG(blocks == 1 || workspace);
    const int vec = aligned16(x, y, x);
    hipLaunchKernelGGL(huber_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, n, delta, scale, vec,
                       blocks == 1 ? loss : workspace);
    if (blocks > 1)
        hipLaunchKernelGGL(huber_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, blocks, scale, loss);
    return dsf_launch_status();
}

extern "C" int dsf_huber_mean_backward(const float* x, const float* y, const float* grad_loss, int64_t n, float delta,
                                       float scale, float* grad_x, dsf_stream_t stream) {
    DSF_CHECK_ARG(n >= 0 && delta > 0.f);
    if (n == 0) return DSF_OK;
    DSF_CHECK_ARG(x && y && grad_loss && grad_x);
    int64_t blocks = (n + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(huber_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, grad_loss, n,
                       delta, scale, aligned16(x, y, grad_x), grad_x);
    return dsf_launch_status();
}
