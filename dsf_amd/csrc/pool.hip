// NHWC max pooling, forward and backward (nn.MaxPool2d(3, 2, 1) of the backbone stem, model/backbone.py:200-204 in the
// reference; nn.MaxPool2d(2, 2) of the hourglass, model/hourglass.py:131).
// HBM-bound streaming kernels: a lane owns 4 consecutive channels (16-byte accesses) of one pixel.
//  * forward: reads the k x k window, writes the maximum and ONE byte per element holding the window position of the
//    argmax (first maximum in (kh, kw) scan order, NaN wins -- torch's rule), instead of torch's 8-byte index;
//  * backward: a GATHER over the <= ceil(k/s)^2 windows that contain an input pixel (fixed order: deterministic, no
//    atomics, no zero fill): B=32 64x128x128 stem 134 MB written once at stream rate.
#include "common.h"

namespace {

struct PoolP { int B, Hi, Wi, C, Ho, Wo, k, stride, pad; };

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ arg, PoolP p, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c4 = p.C >> 2;
    const int c = (int)(i % c4) * 4;
    int64_t q = i / c4;
    const int ox = (int)(q % p.Wo); q /= p.Wo;
    const int oy = (int)(q % p.Ho); const int b = (int)(q / p.Ho);
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int where[4] = {-1, -1, -1, -1};
    for (int kh = 0; kh < p.k; ++kh) {
        const int iy = oy * p.stride - p.pad + kh;
        if ((unsigned)iy >= (unsigned)p.Hi) continue;
        for (int kw = 0; kw < p.k; ++kw) {
            const int ix = ox * p.stride - p.pad + kw;
            if ((unsigned)ix >= (unsigned)p.Wi) continue;
            const float4 v = *reinterpret_cast<const float4*>(x + (((int64_t)b * p.Hi + iy) * p.Wi + ix) * p.C + c);
            const float ve[4] = {v.x, v.y, v.z, v.w};
            const int pos = kh * p.k + kw;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // torch's rule (max_pool2d_with_indices): start at -inf / the first valid tap, replace on `val > max || isnan(val)`
                if (where[e] < 0) where[e] = pos;
                if (ve[e] > best[e] || ve[e] != ve[e]) { best[e] = ve[e]; where[e] = pos; }
            }
        }
    }
    *reinterpret_cast<float4*>(y + i * 4) = make_float4(best[0], best[1], best[2], best[3]);
    if (arg) *reinterpret_cast<uchar4*>(arg + i * 4) = make_uchar4((uint8_t)where[0], (uint8_t)where[1], (uint8_t)where[2], (uint8_t)where[3]);
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ gy, const uint8_t* __restrict__ arg,
                                                          float* __restrict__ gx, PoolP p, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c4 = p.C >> 2;
    const int c = (int)(i % c4) * 4;
    int64_t q = i / c4;
    const int ix = (int)(q % p.Wi); q /= p.Wi;
    const int iy = (int)(q % p.Hi); const int b = (int)(q / p.Hi);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // output windows containing (iy, ix): oy * stride - pad <= iy < oy * stride - pad + k
    const int oy_hi = min((iy + p.pad) / p.stride, p.Ho - 1), ox_hi = min((ix + p.pad) / p.stride, p.Wo - 1);
    const int oy_lo = max(0, (iy + p.pad - p.k + p.stride) / p.stride), ox_lo = max(0, (ix + p.pad - p.k + p.stride) / p.stride);
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const int me = (iy - (oy * p.stride - p.pad)) * p.k + (ix - (ox * p.stride - p.pad));
            const int64_t o = (((int64_t)b * p.Ho + oy) * p.Wo + ox) * p.C + c;
            const uchar4 a = *reinterpret_cast<const uchar4*>(arg + o);
            const float4 g = *reinterpret_cast<const float4*>(gy + o);
            acc[0] += (a.x == me) ? g.x : 0.f; acc[1] += (a.y == me) ? g.y : 0.f;
            acc[2] += (a.z == me) ? g.z : 0.f; acc[3] += (a.w == me) ? g.w : 0.f;
        }
    *reinterpret_cast<float4*>(gx + i * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

}  // namespace

extern "C" {

int dsf_maxpool_forward(const float* x, float* y, uint8_t* argmax, int B, int Hi, int Wi, int C, int Ho, int Wo, int k,
                        int stride, int pad, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && y && B >= 0 && Hi > 0 && Wi > 0 && C > 0 && (C & 3) == 0 && Ho > 0 && Wo > 0);
    DSF_CHECK_ARG(k >= 1 && k <= 15 && stride >= 1 && pad >= 0 && 2 * pad <= k);
    if (B == 0) return DSF_OK;
    PoolP p = {B, Hi, Wi, C, Ho, Wo, k, stride, pad};
    const int64_t n4 = (int64_t)B * Ho * Wo * (C >> 2);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, argmax, p, n4);
    return dsf_launch_status();
}

int dsf_maxpool_backward(const float* grad_y, const uint8_t* argmax, float* grad_x, int B, int Hi, int Wi, int C, int Ho,
                         int Wo, int k, int stride, int pad, dsf_stream_t stream) {
    DSF_CHECK_ARG(grad_y && argmax && grad_x && B >= 0 && Hi > 0 && Wi > 0 && C > 0 && (C & 3) == 0 && Ho > 0 && Wo > 0);
    DSF_CHECK_ARG(k >= 1 && k <= 15 && stride >= 1 && pad >= 0 && 2 * pad <= k);
    if (B == 0) return DSF_OK;
    PoolP p = {B, Hi, Wi, C, Ho, Wo, k, stride, pad};
    const int64_t n4 = (int64_t)B * Hi * Wi * (C >> 2);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_y, argmax, grad_x, p, n4);
    return dsf_launch_status();
}

}  // extern "C"
