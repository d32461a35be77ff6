// Fused training-mode BatchNorm (+ residual add) (+ ReLU) for NHWC activations, forward and backward.
// Replaces the 3 MIOpen BN kernels + add + ReLU (and their 5 backward kernels) that PyTorch launches
// per BasicBlock / Bottleneck stage (model/resnet.py:18-98 in the reference) with 2 + 2 HBM passes.
//
// x is viewed as (M = B*H*W rows, C channels) row-major.  Pass 1 accumulates per-channel sum / sum of
// squares: each lane owns 4 consecutive channels (16-byte loads), a workgroup walks a slab of rows, the
// row-lanes are combined through LDS and the slab totals go to global double accumulators (one f64 atomic
// per channel per workgroup; double keeps E[x^2]-E[x]^2 safe).  Pass 2 is a pure 16-byte-per-lane stream.
// HBM-bound: forward reads x twice and writes y once, backward reads (gy, y, x) twice and writes dx.
#include "common.h"

namespace {

// ---- pass 1: per-channel sums of a (M, C) matrix: sum(a*m), sum(a*m*b_hat) style reductions --------
// MODE 0: stats of x:          s0 = sum x,            s1 = sum x^2
// MODE 1: backward reductions: s0 = sum g,            s1 = sum g * xhat      (g = gy * (y > 0 if relu))
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                        const float* __restrict__ y, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, int64_t M, int C, int relu,
                                                        int rows_per_wg, float* __restrict__ part) {
    __shared__ float s_part[2][256 * 4];
    const int t = threadIdx.x;
    const int c4n = C >> 2;                       // float4 columns
    const int col = t % c4n, rl = t / c4n;        // requires c4n <= 256 and 256 % c4n == 0
    const int rlanes = 256 / c4n;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = (r0 + rows_per_wg < M) ? r0 + rows_per_wg : M;
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {1.f, 1.f, 1.f, 1.f};
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { mu[k] = mean[col * 4 + k]; is[k] = invstd[col * 4 + k]; }
    }
    for (int64_t r = r0 + rl; r < r1; r += rlanes) {
        const float4 xv = *reinterpret_cast<const float4*>(x + r * C + col * 4);
        const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0[k] += xe[k]; a1[k] = fmaf(xe[k], xe[k], a1[k]); }
        } else {
            const float4 gv = *reinterpret_cast<const float4*>(gy + r * C + col * 4);
            float ge[4] = {gv.x, gv.y, gv.z, gv.w};
            if (relu) {
                const float4 yv = *reinterpret_cast<const float4*>(y + r * C + col * 4);
                const float ye[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) ge[k] = (ye[k] > 0.f) ? ge[k] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0[k] += ge[k]; a1[k] = fmaf(ge[k], (xe[k] - mu[k]) * is[k], a1[k]); }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s_part[0][t * 4 + k] = a0[k]; s_part[1][t * 4 + k] = a1[k]; }
    __syncthreads();
    // fold the row-lanes; one partial row per workgroup (no atomics: 1000 workgroups on 2C addresses
    // serialise at the memory side), combined in double by bn_combine_kernel
    float* out = part + (int64_t)blockIdx.x * 2 * C;
    for (int ch = t; ch < C; ch += 256) {
        const int cc = ch >> 2, kk = ch & 3;
        float d0 = 0.f, d1 = 0.f;
        for (int q = 0; q < rlanes; ++q) { d0 += s_part[0][(q * c4n + cc) * 4 + kk]; d1 += s_part[1][(q * c4n + cc) * 4 + kk]; }
        out[ch] = d0;
        out[C + ch] = d1;
    }
}

// sums[2][C] (double accumulate) over the per-workgroup partials; 64 channels x 4 partial-lanes per block
__global__ __launch_bounds__(256) void bn_combine_kernel(const float* __restrict__ part, int wgs, int C,
                                                         double* __restrict__ sums) {
    __shared__ double s[2][256];
    const int t = threadIdx.x, cl = t & 63, pl = t >> 6;
    const int c = blockIdx.x * 64 + cl;
    double d0 = 0.0, d1 = 0.0;
    if (c < C)
        for (int w = pl; w < wgs; w += 4) { d0 += part[(int64_t)w * 2 * C + c]; d1 += part[(int64_t)w * 2 * C + C + c]; }
    s[0][t] = d0; s[1][t] = d1;
    __syncthreads();
    if (pl == 0 && c < C) {
        sums[c] = s[0][cl] + s[0][64 + cl] + s[0][128 + cl] + s[0][192 + cl];
        sums[C + c] = s[1][cl] + s[1][64 + cl] + s[1][128 + cl] + s[1][192 + cl];
    }
}

// finalize forward statistics: mean, invstd; running stats with momentum (unbiased variance)
__global__ void bn_finalize_kernel(const double* __restrict__ acc, int64_t M, int C, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ running_mean,
                                   float* __restrict__ running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mu = acc[c] / (double)M;
    double var = acc[C + c] / (double)M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = (M > 1) ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// ---- pass 2 forward: y = (x - mean) * invstd * gamma + beta (+ residual) (relu) ---------------------
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int64_t n4, int C, int relu, float* __restrict__ y) {
    const int c4n = C >> 2;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % c4n) * 4;
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sc = invstd[c + k] * (gamma ? gamma[c + k] : 1.f);
            o[k] = (xe[k] - mean[c + k]) * sc + (beta ? beta[c + k] : 0.f);
        }
        if (res) {
            const float4 rv = reinterpret_cast<const float4*>(res)[i];
            o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w;
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = fmaxf(o[k], 0.f);
        }
        reinterpret_cast<float4*>(y)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---- pass 2 backward: dx = gamma*invstd*(g - sum_g/M - xhat*sum_gx/M); dres = g -----------------------
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                           const float* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const double* __restrict__ acc, int64_t M, int64_t n4, int C,
                                                           int relu, float* __restrict__ dx, float* __restrict__ dres) {
    const int c4n = C >> 2;
    const float invM = 1.0f / (float)M;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % c4n) * 4;
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        const float4 gv = reinterpret_cast<const float4*>(gy)[i];
        const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
        float ge[4] = {gv.x, gv.y, gv.z, gv.w};
        if (relu) {
            const float4 yv = reinterpret_cast<const float4*>(y)[i];
            const float ye[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) ge[k] = (ye[k] > 0.f) ? ge[k] : 0.f;
        }
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xe[k] - mean[c + k]) * invstd[c + k];
            const float sg = (float)acc[c + k] * invM, sgx = (float)acc[C + c + k] * invM;
            o[k] = (gamma ? gamma[c + k] : 1.f) * invstd[c + k] * (ge[k] - sg - xh * sgx);
        }
        reinterpret_cast<float4*>(dx)[i] = make_float4(o[0], o[1], o[2], o[3]);
        if (dres) reinterpret_cast<float4*>(dres)[i] = make_float4(ge[0], ge[1], ge[2], ge[3]);
    }
}

__global__ void bn_param_grads_kernel(const double* __restrict__ acc, int C, float* __restrict__ dgamma,
                                      float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (dbeta) dbeta[c] = (float)acc[c];
    if (dgamma) dgamma[c] = (float)acc[C + c];
}

// Per-channel sums of an (M rows, C channels) row-major matrix (bias gradients of NHWC activations).
// Each lane owns one float4 column group, the 256 / (C/4) row-lanes of a workgroup walk rows_per_wg rows with 8
// independent 16-byte loads in flight per lane; row-lanes fold through LDS and one float atomic per channel per
// workgroup lands in `out` (zeroed by the caller).  ~2048 workgroups keep every CU's memory pipeline full.
__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ x, int64_t M, int C, int c4n,
                                                      int rows_per_wg, float* __restrict__ out) {
    __shared__ float4 s[256];
    const int t = threadIdx.x;
    const int rlanes = 256 / c4n;                  // c4n <= 256 column groups per pass
    const int col = t % c4n, rl = t / c4n;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = (r0 + rows_per_wg < M) ? r0 + rows_per_wg : M;
    const int C4 = C >> 2;
    for (int cb = 0; cb < C4; cb += c4n) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rl < rlanes && cb + col < C4) {
            const float4* base = reinterpret_cast<const float4*>(x) + cb + col;
            int64_t r = r0 + rl;
            for (; r + 7 * rlanes < r1; r += 8 * rlanes) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = base[(r + u * rlanes) * C4];
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            for (; r < r1; r += rlanes) {
                const float4 v = base[r * C4];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        s[t] = acc;
        __syncthreads();
        if (rl == 0 && cb + col < C4) {
            float4 tot = s[col];
            for (int q = 1; q < rlanes; ++q) { const float4 v = s[q * c4n + col]; tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w; }
            float* o = out + (cb + col) * 4;
            atomicAdd(o, tot.x); atomicAdd(o + 1, tot.y); atomicAdd(o + 2, tot.z); atomicAdd(o + 3, tot.w);
        }
        __syncthreads();
    }
}

// scalar-column variant for channel counts that are not a multiple of 4
__global__ __launch_bounds__(256) void col_sum_scalar_kernel(const float* __restrict__ x, int64_t M, int C, int cpad,
                                                             int rows_per_wg, float* __restrict__ out) {
    __shared__ float s[256];
    const int t = threadIdx.x;
    const int c = t % cpad, rl = t / cpad, rlanes = 256 / cpad;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = (r0 + rows_per_wg < M) ? r0 + rows_per_wg : M;
    for (int cb = 0; cb < C; cb += cpad) {
        float acc = 0.f;
        if (cb + c < C)
            for (int64_t r = r0 + rl; r < r1; r += rlanes) acc += x[r * C + cb + c];
        s[t] = acc;
        __syncthreads();
        if (rl == 0 && cb + c < C) {
            float tot = 0.f;
            for (int q = 0; q < rlanes; ++q) tot += s[q * cpad + c];
            atomicAdd(out + cb + c, tot);
        }
        __syncthreads();
    }
}

inline bool bn_shape_ok(int C) { return C >= 4 && (C & 3) == 0 && (C >> 2) <= 256 && 256 % (C >> 2) == 0; }

constexpr int BN_MAX_WGS = 512;
inline int bn_rows_per_wg(int64_t M) {
    int64_t r = (M + BN_MAX_WGS - 1) / BN_MAX_WGS;
    if (r < 32) r = 32;
    return (int)r;
}
// workspace layout (doubles): [0, 2C) channel sums; then BN_MAX_WGS * 2C floats of per-workgroup partials

}  // namespace

extern "C" int dsf_bn_forward(const float* x, const float* residual, const float* gamma, const float* beta, int64_t M,
                              int C, float eps, float momentum, int relu, float* running_mean, float* running_var,
                              float* y, float* save_mean, float* save_invstd, double* workspace, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && y && save_mean && save_invstd && workspace && M > 0);
    if (!bn_shape_ok(C)) return DSF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int rows = bn_rows_per_wg(M);
    const int wgs = (int)((M + rows - 1) / rows);
    float* part = reinterpret_cast<float*>(workspace + 2 * C);
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(wgs), dim3(256), 0, st, x, nullptr, nullptr, nullptr, nullptr, M, C, 0, rows,
                       part);
    hipLaunchKernelGGL(bn_combine_kernel, dim3((C + 63) / 64), dim3(256), 0, st, part, wgs, C, workspace);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, workspace, M, C, eps, momentum, save_mean,
                       save_invstd, running_mean, running_var);
    const int64_t n4 = M * (C >> 2);
    const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), 0, st, x, residual, save_mean, save_invstd, gamma, beta, n4, C,
                       relu, y);
    return dsf_launch_status();
}

// inference / frozen statistics: y = (x - mean) * invstd * gamma + beta with given mean / invstd
extern "C" int dsf_bn_apply(const float* x, const float* residual, const float* gamma, const float* beta,
                            const float* mean, const float* invstd, int64_t M, int C, int relu, float* y,
                            dsf_stream_t stream) {
    DSF_CHECK_ARG(x && y && mean && invstd && M > 0);
    if (C < 4 || (C & 3)) return DSF_ERR_UNSUPPORTED;
    const int64_t n4 = M * (C >> 2);
    const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, residual, mean, invstd, gamma, beta,
                       n4, C, relu, y);
    return dsf_launch_status();
}

extern "C" int dsf_bn_backward(const float* x, const float* grad_y, const float* y, const float* gamma,
                               const float* save_mean, const float* save_invstd, int64_t M, int C, int relu,
                               float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta,
                               double* workspace, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && grad_y && save_mean && save_invstd && grad_x && workspace && M > 0 && (!relu || y));
    if (!bn_shape_ok(C)) return DSF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int rows = bn_rows_per_wg(M);
    const int wgs = (int)((M + rows - 1) / rows);
    float* part = reinterpret_cast<float*>(workspace + 2 * C);
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3(wgs), dim3(256), 0, st, x, grad_y, y, save_mean, save_invstd, M, C, relu,
                       rows, part);
    hipLaunchKernelGGL(bn_combine_kernel, dim3((C + 63) / 64), dim3(256), 0, st, part, wgs, C, workspace);
    const int64_t n4 = M * (C >> 2);
    const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 0, st, x, grad_y, y, save_mean, save_invstd, gamma,
                       workspace, M, n4, C, relu, grad_x, grad_residual);
    if (grad_gamma || grad_beta)
        hipLaunchKernelGGL(bn_param_grads_kernel, dim3((C + 255) / 256), dim3(256), 0, st, workspace, C, grad_gamma, grad_beta);
    return dsf_launch_status();
}

extern "C" int dsf_col_sum(const float* x, int64_t M, int C, float* out, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && out && M > 0 && C > 0);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float) * C, st) != hipSuccess) return DSF_ERR_LAUNCH;
    if ((C & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        int c4n = 1;
        while (c4n < (C >> 2) && c4n < 256) c4n <<= 1;        // float4 column groups per pass (power of two <= 256)
        const int rlanes = 256 / c4n;
        int64_t rows = (M + 2047) / 2048;                      // ~2048 workgroups
        const int64_t min_rows = (int64_t)rlanes * 8;          // at least one unrolled trip per lane
        if (rows < min_rows) rows = min_rows;
        const int wgs = (int)((M + rows - 1) / rows);
        hipLaunchKernelGGL(col_sum_kernel, dim3(wgs), dim3(256), 0, st, x, M, C, c4n, (int)rows, out);
        return dsf_launch_status();
    }
    int cpad = 1;
    while (cpad < C && cpad < 256) cpad <<= 1;         // columns handled per pass (power of two <= 256)
    int64_t rows = (M + 255) / 256;
    if (rows < 64) rows = 64;
    const int wgs = (int)((M + rows - 1) / rows);
    hipLaunchKernelGGL(col_sum_scalar_kernel, dim3(wgs), dim3(256), 0, st, x, M, C, cpad, (int)rows, out);
    return dsf_launch_status();
}
