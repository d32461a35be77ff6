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

// ---- pass 1: per-channel reductions of a (M, C) matrix + finalisation by the last workgroup ---------------
// MODE 0: stats of x:          s0 = sum x,            s1 = sum x^2          -> mean, invstd, running stats
// MODE 1: backward reductions: s0 = sum g,            s1 = sum g * xhat     -> sums (for pass 2), dbeta, dgamma
//                              (g = gy * (y > 0) when the ReLU is fused)
// Each lane owns 4 consecutive channels (16-byte loads, 4 rows in flight), the row-lanes of a workgroup fold through
// LDS and the workgroup writes ONE float partial per channel (no atomics on the 2C sums: hundreds of workgroups on
// 2C addresses serialise at the memory side).  bn_finalize_kernel adds the partials in double, in a fixed order.
// (An in-launch finalisation by the last-arriving workgroup was measured: its serial walk over up to 256 KB of
// partials costs 60+ us, a second tiny launch ~4 us.)
struct BnFinal {
    // MODE 0
    float eps, momentum;
    float* mean; float* invstd; float* running_mean; float* running_var;
    // MODE 1
    double* sums; float* dgamma; float* dbeta;
    int accum = 0;          // dgamma / dbeta are ADDED to what the buffers hold (a second application of the layer in one backward pass)
};

// G2: the incoming gradient is the SUM of two tensors, gy + gy2 (the fan-in of a block output that feeds the next block's first
//     convolution and its identity path, model/resnet.py:39-55: autograd would add them with a pass of its own, 2R + 1W).
// WG: the (summed, ReLU-masked) gradient g is also WRITTEN to gout -- it is the residual path's gradient, and the apply pass
//     that follows then reads x and gout only (no y, no second addend, no mask), writing dx alone.
template <int MODE, bool YMASK = false, bool G2 = false, bool WG = false>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                        const float* __restrict__ y, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int64_t M, int C, int relu,
                                                        int rows_per_wg, float* __restrict__ part, int acc_rows,
                                                        const float* __restrict__ gy2 = nullptr, float* __restrict__ gout = nullptr) {
    // (with acc_rows > 0 `part` is really a double array: the accumulation rows are DOUBLES, see bn_fold_rows)
    // acc_rows == 0: workgroup w stores partial row w (bn_finalize_kernel folds them in a fixed order: deterministic).
    // acc_rows  > 0: the workgroup ADDS its partial row into row (w mod acc_rows) of a zeroed [acc_rows][2][C] block of DOUBLES
    //                with global_atomic_add_f64 (64 adders per address at 512 workgroups); the apply kernels fold those few
    //                rows in their own prologue, so that no finalise launch sits between the passes.  Doubles: the float
    //                partials are then summed exactly as bn_finalize_kernel does (a double sum of <= 1024 floats is order-
    //                independent to ~1e-16), so the statistics match the ordered path to the last float bit almost always.
    // relu: 0 none, 1 mask from the saved output y (y > 0), 2 mask recomputed from x with the forward's own expression
    // (x - mean) * (invstd * gamma) + beta > 0 -- bit-identical to the forward, and y is not read (nor kept) at all
    __shared__ float s_part[2][256 * 4];
    const int t = threadIdx.x;
    // blockIdx.z: independent (M, C) matrices stacked in memory, each with its own accumulation block (instance
    // normalisation: one per sample; 1 for BatchNorm)
    x += (int64_t)blockIdx.z * M * C;
    if (acc_rows) part += (int64_t)blockIdx.z * acc_rows * 2 * C * 2;         // (doubles: two floats each)
    // float4 columns: a workgroup covers a block of c4b <= 256 of them (blockIdx.y: C > 1024 takes several blocks)
    const int c4b = min(C >> 2, 256);             // 256 % c4b == 0
    const int lcol = t % c4b, rl = t / c4b;
    const int col = blockIdx.y * c4b + lcol;
    const int rlanes = 256 / c4b;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = (r0 + rows_per_wg < M) ? r0 + rows_per_wg : M;
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {1.f, 1.f, 1.f, 1.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mu[k] = mean[col * 4 + k]; is[k] = invstd[col * 4 + k];
            sc[k] = is[k] * (gamma ? gamma[col * 4 + k] : 1.f);
            sh[k] = beta ? beta[col * 4 + k] : 0.f;
        }
    }
    auto accumulate = [&](const float4& xv, float4& gv, const float4& yv, const float4& hv) {
        const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0[k] += xe[k]; a1[k] = fmaf(xe[k], xe[k], a1[k]); }
        } else {
            float ge[4] = {gv.x, gv.y, gv.z, gv.w};
            if (G2) { ge[0] += hv.x; ge[1] += hv.y; ge[2] += hv.z; ge[3] += hv.w; }     // (the add autograd would have done: same fp32 sum)
            if (YMASK) {                     // (YMASK <=> relu == 1: the launchers pick the instantiation)
                const float ye[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) ge[k] = (ye[k] > 0.f) ? ge[k] : 0.f;
            } else if (relu == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) ge[k] = ((xe[k] - mu[k]) * sc[k] + sh[k] > 0.f) ? ge[k] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0[k] += ge[k]; a1[k] = fmaf(ge[k], (xe[k] - mu[k]) * is[k], a1[k]); }
            if (WG) gv = make_float4(ge[0], ge[1], ge[2], ge[3]);
        }
    };
    // U rows per lane in flight (forward 8 x 16 bytes of x; backward 4 of x, of gy, and of y with a saved-output mask): with one workgroup per CU
    // the pass is latency-bound on what a wave keeps outstanding (4 rows: 2.8 TB/s on a 34 MB tensor).  Rows past r1 load
    // nothing and add zeros.
    constexpr int U = (MODE == 0) ? 8 : 4;      // backward: 2-3 tensors per row, and the wave has to fit beside the backward-weights kernels
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r = r0 + rl; r < r1; r += (int64_t)U * rlanes) {
        float4 xv[U], gv[U], yv[U], hv[G2 ? U : 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t rr = r + (int64_t)u * rlanes;
            const bool ok = rr < r1;
            const int64_t o = (ok ? rr : r) * C + col * 4;
            xv[u] = ok ? *reinterpret_cast<const float4*>(x + o) : z4;
            gv[u] = (MODE == 1 && ok) ? *reinterpret_cast<const float4*>(gy + o) : z4;
            yv[u] = (YMASK && ok) ? *reinterpret_cast<const float4*>(y + o) : z4;
            if (G2) hv[u] = ok ? *reinterpret_cast<const float4*>(gy2 + o) : z4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            accumulate(xv[u], gv[u], yv[u], hv[G2 ? u : 0]);
            if (WG) {
                const int64_t rr = r + (int64_t)u * rlanes;
                if (rr < r1) *reinterpret_cast<float4*>(gout + rr * C + col * 4) = gv[u];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s_part[0][t * 4 + k] = a0[k]; s_part[1][t * 4 + k] = a1[k]; }
    __syncthreads();
    const int nv = 2 * C;
    float* out = part + (int64_t)blockIdx.x * nv + blockIdx.y * c4b * 4;
    double* acc = reinterpret_cast<double*>(part) + (int64_t)(acc_rows ? (int)(blockIdx.x % acc_rows) : 0) * nv + blockIdx.y * c4b * 4;
    for (int ch = t; ch < c4b * 4; ch += 256) {
        const int cc = ch >> 2, kk = ch & 3;
        float d0 = 0.f, d1 = 0.f;
        for (int q = 0; q < rlanes; ++q) { d0 += s_part[0][(q * c4b + cc) * 4 + kk]; d1 += s_part[1][(q * c4b + cc) * 4 + kk]; }
        if (acc_rows) {
            __hip_atomic_fetch_add(acc + ch, (double)d0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(acc + C + ch, (double)d1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else { out[ch] = d0; out[C + ch] = d1; }
    }
}

// ---- pass 1b: combine the per-workgroup partials (double, fixed order: deterministic) and finalise ----------------
// One workgroup per 16 channels (32 values): 8 partial-lanes per value walk the `wgs` partial rows, fold through LDS.
template <int MODE>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int wgs, int64_t M, int C,
                                                          BnFinal fin) {
    __shared__ double s_fold[256];
    const int t = threadIdx.x, vl = t & 31, pl = t >> 5;           // value-lane (16 channels x {s0, s1}), partial-lane
    const int cl = vl & 15, which = vl >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int nv = 2 * C;
    double d = 0.0;
    if (c < C) {
        const float* p = part + which * C + c;
        int w = pl;
        for (; w + 24 < wgs; w += 32) {
            const float p0 = p[(int64_t)w * nv], p1 = p[(int64_t)(w + 8) * nv], p2 = p[(int64_t)(w + 16) * nv], p3 = p[(int64_t)(w + 24) * nv];
            d += ((double)p0 + (double)p1) + ((double)p2 + (double)p3);
        }
        for (; w < wgs; w += 8) d += (double)p[(int64_t)w * nv];
    }
    s_fold[t] = d;
    __syncthreads();
    if (t < 16 && c < C) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { s0 += s_fold[q * 32 + cl]; s1 += s_fold[q * 32 + 16 + cl]; }
        if (MODE == 0) {
            const double m = s0 / (double)M;
            double var = s1 / (double)M - m * m;
            if (var < 0.0) var = 0.0;
            fin.mean[c] = (float)m;
            fin.invstd[c] = (float)(1.0 / sqrt(var + (double)fin.eps));
            if (fin.running_mean) {
                const double unbiased = (M > 1) ? var * (double)M / (double)(M - 1) : var;
                fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * (float)m;
                fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unbiased;
            }
        } else {
            fin.sums[c] = s0;
            fin.sums[C + c] = s1;
            if (fin.dbeta) fin.dbeta[c] = (fin.accum ? fin.dbeta[c] : 0.f) + (float)s0;
            if (fin.dgamma) fin.dgamma[c] = (fin.accum ? fin.dgamma[c] : 0.f) + (float)s1;
        }
    }
}

// ---- per-workgroup channel parameters of the apply passes ---------------------------------------------------------------------
// A workgroup's 256 threads hold only nq = min(C / 4, 256) distinct channel quads (thread t: quad (blockIdx.x * 256 + t) mod C/4).
// Rounds 1-3 let EVERY thread fold the accumulation rows and load the per-channel vectors of its quad: 512 bytes of loads per
// thread, 1 MB per CU through the vector L1 at 8 workgroups -- a third of the pass on the 34 MB layers.  Now the 256 / nq threads
// that share a quad split its rows, the partial sums meet in LDS, the first nq threads ("owners") finish the arithmetic and
// publish NP float4 per quad, and everybody reads its own.  C >= 1024: one quad per thread, nothing to share.
constexpr int BN_FOLD_LDS = 256 * 8;               // doubles: one 8-double partial per thread; reused for the published float4s

// the two folded sums of quad `c` for the owner threads (t < nq); rows [row][2][C] doubles; all threads must call
__device__ __forceinline__ void bn_fold_rows_wg(const double* __restrict__ rows, int n_rows, int C, int nq, int c, double* s_buf,
                                                double (&s0)[4], double (&s1)[4]) {
    const int t = threadIdx.x, dup = 256 / nq, j = t / nq;
    double p[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int r = j; r < n_rows; r += dup) {
        const double2* pa = reinterpret_cast<const double2*>(rows + (int64_t)r * 2 * C + c);
        const double2* pb = reinterpret_cast<const double2*>(rows + (int64_t)r * 2 * C + C + c);
        const double2 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
        p[0] += a0.x; p[1] += a0.y; p[2] += a1.x; p[3] += a1.y;
        p[4] += b0.x; p[5] += b0.y; p[6] += b1.x; p[7] += b1.y;
    }
    const int helpers = min(dup, n_rows);
    if (j < helpers) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s_buf[k * 256 + t] = p[k];          // [value][thread]: conflict-free
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) { s0[k] = 0.0; s1[k] = 0.0; }
    if (t < nq) {
        for (int h = 0; h < helpers; ++h) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { s0[k] += s_buf[k * 256 + h * nq + t]; s1[k] += s_buf[(4 + k) * 256 + h * nq + t]; }
        }
    }
    __syncthreads();                                                     // s_buf is free again
}

// owners publish par[NP][4] for their quad; every thread of the quad reads it back
template <int NP>
__device__ __forceinline__ void bn_share_params(int nq, float (&par)[NP][4], double* s_buf) {
    float4* s_par = reinterpret_cast<float4*>(s_buf);                    // NP * nq float4 <= 6 * 128 * 16 bytes < 16 KB
    const int t = threadIdx.x;
    if (t < nq) {
#pragma unroll
        for (int q = 0; q < NP; ++q) s_par[q * nq + t] = make_float4(par[q][0], par[q][1], par[q][2], par[q][3]);
    }
    __syncthreads();
    const int o = t % nq;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const float4 v = s_par[q * nq + o];
        par[q][0] = v.x; par[q][1] = v.y; par[q][2] = v.z; par[q][3] = v.w;
    }
}

// ---- streaming accessors of the apply passes -----------------------------------------------------------------------------------
// NT: non-temporal 16-byte accesses (global_load_dwordx4 ... nt): the apply passes touch every byte once.
typedef float bn_v4f __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 bn_ld4(const float* __restrict__ p, int64_t j) {
    if (NT) {
        const bn_v4f v = __builtin_nontemporal_load(reinterpret_cast<const bn_v4f*>(p) + j);
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return reinterpret_cast<const float4*>(p)[j];
}
template <bool NT> __device__ __forceinline__ void bn_st4(float* __restrict__ p, int64_t j, const float4& v) {
    if (NT) {
        bn_v4f w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
        __builtin_nontemporal_store(w, reinterpret_cast<bn_v4f*>(p) + j);
    } else {
        reinterpret_cast<float4*>(p)[j] = v;
    }
}
// VAR (tuning aid DSF_BN_VAR, read per call): bit 0 = non-temporal loads, bit 1 = non-temporal stores, bit 2 = the next batch of
// loads is issued BEFORE the current one is processed and stored (two register sets)
constexpr int BN_VAR_DEFAULT = 1;     // non-temporal LOADS, ordinary stores: what an apply pass writes is read by the next kernel (in-step A/B on the final round-6 code, one box: config 2 15.95 -> 15.77 ms, config 5 72.4 -> 71.6, config 3 15.31 -> 15.06, config 4 153.2 = 153.3; mid-round, before the other changes, 3 = loads and stores had measured better on configs 4 / 5: profiles/r06_bn_nontemporal.txt); the prefetching variants 4 / 7 bring nothing more
// DSF_BN_WRITE_G=0 (tuning aid, read per call): the sums pass does not write the masked gradient, the apply pass re-reads gy (+ gy2) and y
static inline bool bn_write_g() { const char* e = getenv("DSF_BN_WRITE_G"); return !e || atoi(e) != 0; }
static inline int bn_var() { const char* e = getenv("DSF_BN_VAR"); const int v = e ? atoi(e) : BN_VAR_DEFAULT; return (v < 0 || v > 7) ? BN_VAR_DEFAULT : v; }

// ---- pass 2 forward: y = (x - mean) * invstd * gamma + beta (+ residual) (relu) ---------------------
// The grid stride is a multiple of the float4 column count, so a thread's channel quad is loop-invariant.
// FOLD: mean / invstd are not read but COMPUTED in the prologue from the accumulation rows (folded once per workgroup, see
// above), with bn_finalize_kernel<0>'s arithmetic; the owners of the first workgroup(s) (i0 < C / 4) also store mean / invstd
// for the backward pass and update the running statistics.
// Four float4 of x (and of the residual) are in flight per thread, and the first batch is issued BEFORE the prologue.
template <bool FOLD, int VAR>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int64_t n4, int C, int relu, float* __restrict__ y,
                                                       const double* __restrict__ rows, int n_rows, int64_t M, BnFinal fin) {
    __shared__ double s_buf[BN_FOLD_LDS];
    constexpr int U = 4;
    constexpr bool NTL = (VAR & 1) != 0, NTS = (VAR & 2) != 0, PF = (VAR & 4) != 0;
    const int c4n = C >> 2, nq = min(c4n, 256);
    const int64_t i0 = blockIdx.x * (int64_t)256 + threadIdx.x;
    const int64_t S = (int64_t)gridDim.x * 256;
    const int c = (int)(i0 % c4n) * 4;
    const bool owner = (int)threadIdx.x < nq;
    if (FOLD && gridDim.y > 1) {                               // blockIdx.y: stacked matrices of n4 float4 each (instance normalisation)
        x += (int64_t)blockIdx.y * n4 * 4; y += (int64_t)blockIdx.y * n4 * 4;
        if (res) res += (int64_t)blockIdx.y * n4 * 4;
        rows += (int64_t)blockIdx.y * n_rows * 2 * C;
    }
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 xa[U], ra[U], xb[PF ? U : 1], rb[PF ? U : 1];
    auto fetch = [&](int64_t i, float4* xv, float4* rv) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = i + u * S;
            const bool ok = j < n4;
            xv[u] = ok ? bn_ld4<NTL>(x, j) : z4;
            rv[u] = (res && ok) ? bn_ld4<NTL>(res, j) : z4;
        }
    };
    fetch(i0, xa, ra);
    float par[3][4];                                           // mean, scale = invstd * gamma, shift = beta
    if (FOLD) {
        double s0[4], s1[4];
        bn_fold_rows_wg(rows, n_rows, C, nq, c, s_buf, s0, s1);
        if (owner) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double m = s0[k] / (double)M;
                double var = s1[k] / (double)M - m * m;
                if (var < 0.0) var = 0.0;
                const float is = (float)(1.0 / sqrt(var + (double)fin.eps));
                par[0][k] = (float)m;
                par[1][k] = is * (gamma ? gamma[c + k] : 1.f);
                par[2][k] = beta ? beta[c + k] : 0.f;
                if (i0 < c4n && fin.mean) {
                    fin.mean[c + k] = (float)m;
                    fin.invstd[c + k] = is;
                    if (fin.running_mean) {
                        const double unbiased = (M > 1) ? var * (double)M / (double)(M - 1) : var;
                        fin.running_mean[c + k] = (1.f - fin.momentum) * fin.running_mean[c + k] + fin.momentum * (float)m;
                        fin.running_var[c + k] = (1.f - fin.momentum) * fin.running_var[c + k] + fin.momentum * (float)unbiased;
                    }
                }
            }
        }
    } else if (owner) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            par[0][k] = mean[c + k];
            par[1][k] = invstd[c + k] * (gamma ? gamma[c + k] : 1.f);
            par[2][k] = beta ? beta[c + k] : 0.f;
        }
    }
    if (nq < 256) bn_share_params<3>(nq, par, s_buf);
    auto process = [&](int64_t i, const float4* xv, const float4* rv) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = i + u * S;
            if (j >= n4) break;
            const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (xe[k] - par[0][k]) * par[1][k] + par[2][k];
            if (res) { o[0] += rv[u].x; o[1] += rv[u].y; o[2] += rv[u].z; o[3] += rv[u].w; }
            if (relu) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = fmaxf(o[k], 0.f);
            }
            bn_st4<NTS>(y, j, make_float4(o[0], o[1], o[2], o[3]));
        }
    };
    const int64_t US = (int64_t)U * S;
    if (PF) {
        for (int64_t i = i0; i < n4;) {
            if (i + US < n4) fetch(i + US, xb, rb);
            process(i, xa, ra);
            i += US;
            if (i >= n4) break;
            if (i + US < n4) fetch(i + US, xa, ra);
            process(i, xb, rb);
            i += US;
        }
    } else {
        for (int64_t i = i0; i < n4;) {
            process(i, xa, ra);
            i += US;
            if (i < n4) fetch(i, xa, ra);
        }
    }
}

// ---- pass 2 backward: dx = gamma*invstd*(g - sum_g/M - xhat*sum_gx/M); dres = g -----------------------
// FOLD: the two channel sums come from the accumulation rows (folded once per workgroup, as in bn_apply_kernel<true>) instead of
// the finalise launch's doubles; the owners of the first workgroup(s) also store dgamma / dbeta.
// G2: g = gy + gy2 (see bn_reduce_kernel).  A launch behind a reduce pass that wrote the masked gradient (WG) gets that tensor as
// gy with relu = 0 and dres = nullptr: two reads, one write.
// (launch bound 5, not 6: the same 80 VGPRs, but bound 6 made the allocator spill three dwords that the main loop reloaded per
//  iteration; config 4 172.0 -> 170.9 ms, config 2 unchanged)
template <bool FOLD, bool G2, int VAR>
__global__ __launch_bounds__(256, G2 ? 3 : ((VAR & 4) ? 4 : 5)) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                           const float* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const double* __restrict__ acc,
                                                           int64_t M, int64_t n4, int C, int relu, float* __restrict__ dx,
                                                           float* __restrict__ dres, const double* __restrict__ rows, int n_rows,
                                                           BnFinal fin, const double* __restrict__ count,
                                                           const float* __restrict__ gy2) {
    // count != nullptr: the divisor is the element count of the WHOLE (cross-replica) batch, read from device memory
    __shared__ double s_buf[BN_FOLD_LDS];
    constexpr int U = 2;                                      // (the wave has to fit beside the backward-weights kernels: ~150 VGPRs per SIMD)
    constexpr bool NTL = (VAR & 1) != 0, NTS = (VAR & 2) != 0, PF = (VAR & 4) != 0;
    const int c4n = C >> 2, nq = min(c4n, 256);
    const int64_t i0 = blockIdx.x * (int64_t)256 + threadIdx.x;
    const int64_t S = (int64_t)gridDim.x * 256;
    const int c = (int)(i0 % c4n) * 4;
    const bool owner = (int)threadIdx.x < nq;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    struct Set { float4 xv[U], gv[U], yv[U], hv[G2 ? U : 1]; };
    Set A, B;                                                 // (B: the second register set of the prefetching variant)
    auto fetch = [&](int64_t i, Set& s) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = i + u * S;
            const bool ok = j < n4;
            s.xv[u] = ok ? bn_ld4<NTL>(x, j) : z4;
            s.gv[u] = ok ? bn_ld4<NTL>(gy, j) : z4;
            s.yv[u] = (relu == 1 && ok) ? bn_ld4<NTL>(y, j) : z4;
            if (G2) s.hv[u] = ok ? bn_ld4<NTL>(gy2, j) : z4;
        }
    };
    fetch(i0, A);
    double f0[4] = {0.0, 0.0, 0.0, 0.0}, f1[4] = {0.0, 0.0, 0.0, 0.0};
    if (FOLD) {
        bn_fold_rows_wg(rows, n_rows, C, nq, c, s_buf, f0, f1);
        if (owner && i0 < c4n) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (fin.dbeta) fin.dbeta[c + k] = (fin.accum ? fin.dbeta[c + k] : 0.f) + (float)f0[k];
                if (fin.dgamma) fin.dgamma[c + k] = (fin.accum ? fin.dgamma[c + k] : 0.f) + (float)f1[k];
            }
        }
    }
    float par[6][4];                                           // mean, invstd, sum_g / M, sum_gx / M, invstd * gamma, beta
    if (owner) {
        const float invM = count ? (float)(1.0 / count[0]) : 1.0f / (float)M;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            par[0][k] = mean[c + k]; par[1][k] = invstd[c + k];
            par[2][k] = (float)(FOLD ? f0[k] : acc[c + k]) * invM; par[3][k] = (float)(FOLD ? f1[k] : acc[C + c + k]) * invM;
            par[4][k] = (gamma ? gamma[c + k] : 1.f) * par[1][k];
            par[5][k] = beta ? beta[c + k] : 0.f;
        }
    }
    if (nq < 256) bn_share_params<6>(nq, par, s_buf);
    auto process = [&](int64_t i, const Set& s) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = i + u * S;
            if (j >= n4) break;
            const float xe[4] = {s.xv[u].x, s.xv[u].y, s.xv[u].z, s.xv[u].w};
            float ge[4] = {s.gv[u].x, s.gv[u].y, s.gv[u].z, s.gv[u].w};
            if (G2) { ge[0] += s.hv[u].x; ge[1] += s.hv[u].y; ge[2] += s.hv[u].z; ge[3] += s.hv[u].w; }
            if (relu == 1) {
                const float ye[4] = {s.yv[u].x, s.yv[u].y, s.yv[u].z, s.yv[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) ge[k] = (ye[k] > 0.f) ? ge[k] : 0.f;
            } else if (relu == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) ge[k] = ((xe[k] - par[0][k]) * par[4][k] + par[5][k] > 0.f) ? ge[k] : 0.f;
            }
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (xe[k] - par[0][k]) * par[1][k];
                o[k] = par[4][k] * (ge[k] - par[2][k] - xh * par[3][k]);
            }
            bn_st4<NTS>(dx, j, make_float4(o[0], o[1], o[2], o[3]));
            if (dres) bn_st4<NTS>(dres, j, make_float4(ge[0], ge[1], ge[2], ge[3]));
        }
    };
    const int64_t US = (int64_t)U * S;
    if (PF) {
        for (int64_t i = i0; i < n4;) {
            if (i + US < n4) fetch(i + US, B);
            process(i, A);
            i += US;
            if (i >= n4) break;
            if (i + US < n4) fetch(i + US, A);
            process(i, B);
            i += US;
        }
    } else {
        for (int64_t i = i0; i < n4;) {
            process(i, A);
            i += US;
            if (i < n4) fetch(i, A);
        }
    }
}

// ---- small maps: statistics AND apply in one launch -------------------------------------------------------------------------
// On the 4 x 4 and 2 x 2 levels of an hourglass (M = B H W <= 1024 rows) a launch costs more than the pass: the reduce + apply
// pair is 12 us of which a few are work, and it sits on the step's critical chain 36 times per direction (config 3).  Here ONE
// workgroup owns a channel quad (one float4 column): its 256 threads hold the column's rows in registers (R = 1 or 4 float4 each,
// M <= 256 R), fold the two sums in double through shuffles and LDS -- a fixed order: the kernels serve the deterministic mode too
// -- and apply from the registers.  The arithmetic per element is the two-launch path's, operation for operation; the sums differ
// from it in summation order only.
__device__ __forceinline__ void bn_small_fold(double (&v)[8], double* s_red) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s_red[wave * 8 + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (s_red[k] + s_red[8 + k]) + (s_red[16 + k] + s_red[24 + k]);
}

template <int R>
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int M,
                                                           int C, int relu, float* __restrict__ y, BnFinal fin) {
    __shared__ double s_red[32];
    const int t = threadIdx.x, C4 = C >> 2, c4 = blockIdx.x, c = c4 * 4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 xv[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int r = t + u * 256;
        xv[u] = z4;                                                       // (not a select: it would pick between POINTERS, with z4 in scratch)
        if (r < M) xv[u] = reinterpret_cast<const float4*>(x)[(int64_t)r * C4 + c4];
    }
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { a0[k] += xe[k]; a1[k] = fmaf(xe[k], xe[k], a1[k]); }
    }
    double v[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (double)a0[k]; v[4 + k] = (double)a1[k]; }
    bn_small_fold(v, s_red);
    float par[3][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double m = v[k] / (double)M;
        double var = v[4 + k] / (double)M - m * m;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)fin.eps));
        par[0][k] = (float)m;
        par[1][k] = is * (gamma ? gamma[c + k] : 1.f);
        par[2][k] = beta ? beta[c + k] : 0.f;
        if (t == 0 && fin.mean) {
            fin.mean[c + k] = (float)m;
            fin.invstd[c + k] = is;
            if (fin.running_mean) {
                const double unbiased = (M > 1) ? var * (double)M / (double)(M - 1) : var;
                fin.running_mean[c + k] = (1.f - fin.momentum) * fin.running_mean[c + k] + fin.momentum * (float)m;
                fin.running_var[c + k] = (1.f - fin.momentum) * fin.running_var[c + k] + fin.momentum * (float)unbiased;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int r = t + u * 256;
        if (r < M) {
            const int64_t j = (int64_t)r * C4 + c4;
            const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (xe[k] - par[0][k]) * par[1][k] + par[2][k];
            if (res) { const float4 rv = reinterpret_cast<const float4*>(res)[j]; o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w; }
            if (relu) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = fmaxf(o[k], 0.f);
            }
            reinterpret_cast<float4*>(y)[j] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

template <int R>
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                           const float* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int M, int C, int relu,
                                                           float* __restrict__ dx, float* __restrict__ dres, BnFinal fin,
                                                           const float* __restrict__ gy2) {
    __shared__ double s_red[32];
    const int t = threadIdx.x, C4 = C >> 2, c4 = blockIdx.x, c = c4 * 4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float mu[4], is[4], sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mu[k] = mean[c + k]; is[k] = invstd[c + k];
        sc[k] = is[k] * (gamma ? gamma[c + k] : 1.f);
        sh[k] = beta ? beta[c + k] : 0.f;
    }
    float4 xv[R], gv[R];
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int r = t + u * 256;
        const bool ok = r < M;
        const int64_t j = (int64_t)(ok ? r : 0) * C4 + c4;
        xv[u] = ok ? reinterpret_cast<const float4*>(x)[j] : z4;
        gv[u] = ok ? reinterpret_cast<const float4*>(gy)[j] : z4;
        const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
        float ge[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
        if (gy2 && ok) {                                                  // second addend of the incoming gradient (see bn_reduce_kernel)
            const float4 hv = reinterpret_cast<const float4*>(gy2)[j];
            ge[0] += hv.x; ge[1] += hv.y; ge[2] += hv.z; ge[3] += hv.w;
        }
        if (relu == 1) {
            const float4 yv = ok ? reinterpret_cast<const float4*>(y)[j] : z4;
            const float ye[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) ge[k] = (ye[k] > 0.f) ? ge[k] : 0.f;
        } else if (relu == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ge[k] = ((xe[k] - mu[k]) * sc[k] + sh[k] > 0.f) ? ge[k] : 0.f;
        }
        gv[u] = make_float4(ge[0], ge[1], ge[2], ge[3]);                  // (the masked gradient: what both passes use)
#pragma unroll
        for (int k = 0; k < 4; ++k) { a0[k] += ge[k]; a1[k] = fmaf(ge[k], (xe[k] - mu[k]) * is[k], a1[k]); }
    }
    double v[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (double)a0[k]; v[4 + k] = (double)a1[k]; }
    bn_small_fold(v, s_red);
    float m0[4], m1[4];
    const float invM = 1.0f / (float)M;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (t == 0) {
            if (fin.dbeta) fin.dbeta[c + k] = (fin.accum ? fin.dbeta[c + k] : 0.f) + (float)v[k];
            if (fin.dgamma) fin.dgamma[c + k] = (fin.accum ? fin.dgamma[c + k] : 0.f) + (float)v[4 + k];
        }
        m0[k] = (float)v[k] * invM; m1[k] = (float)v[4 + k] * invM;
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int r = t + u * 256;
        if (r < M) {
            const int64_t j = (int64_t)r * C4 + c4;
            const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
            const float ge[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (xe[k] - mu[k]) * is[k];
                o[k] = sc[k] * (ge[k] - m0[k] - xh * m1[k]);
            }
            reinterpret_cast<float4*>(dx)[j] = make_float4(o[0], o[1], o[2], o[3]);
            if (dres) reinterpret_cast<float4*>(dres)[j] = gv[u];
        }
    }
}

constexpr int BN_SMALL_MAX_ROWS = 1024;      // (16 float4 per thread, M <= 4096, was built and measured: 19-30 us forward, 37-49 backward against ~12 for the pair --
                                             //  a thread per ROW of a 1-4 MB column block is an uncoalesced walk; 4 per thread breaks even, 1 per thread halves the time)
static const int bn_small_on = [] { const char* e = getenv("DSF_BN_SMALL"); return e ? atoi(e) : 1; }();     // tuning aid
inline bool bn_small_ok(int64_t M, int C) { return bn_small_on && M <= BN_SMALL_MAX_ROWS && C >= 4 && (C & 3) == 0; }

// Per-channel sums of an (M rows, C channels) row-major matrix (bias gradients of NHWC activations).
// Each lane owns one float4 column group, the 256 / (C/4) row-lanes of a workgroup walk rows_per_wg rows with 8
// independent 16-byte loads in flight per lane; row-lanes fold through LDS and the workgroup writes one partial row
// (float atomics from ~2000 workgroups onto C addresses serialise at the memory side: measured 200 us for a 134 MB
// tensor); col_sum_combine_kernel adds the <= 256 partial rows.
__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ x, int64_t M, int C, int c4n,
                                                      int rows_per_wg, float* __restrict__ part) {
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
            reinterpret_cast<float4*>(part + (int64_t)blockIdx.x * C)[cb + col] = tot;
        }
        __syncthreads();
    }
}

// out[c] = sum over the partial rows, fixed order (deterministic); 64 channels x 4 partial-lanes per workgroup
__global__ __launch_bounds__(256) void col_sum_combine_kernel(const float* __restrict__ part, int wgs, int C,
                                                              float* __restrict__ out) {
    __shared__ float s[256];
    const int t = threadIdx.x, cl = t & 63, pl = t >> 6;
    const int c = blockIdx.x * 64 + cl;
    float d = 0.f;
    if (c < C) {
        int w = pl;
        for (; w + 12 < wgs; w += 16) {
            const float p0 = part[(int64_t)w * C + c], p1 = part[(int64_t)(w + 4) * C + c];
            const float p2 = part[(int64_t)(w + 8) * C + c], p3 = part[(int64_t)(w + 12) * C + c];
            d += (p0 + p1) + (p2 + p3);
        }
        for (; w < wgs; w += 4) d += part[(int64_t)w * C + c];
    }
    s[t] = d;
    __syncthreads();
    if (pl == 0 && c < C) out[c] = (s[cl] + s[64 + cl]) + (s[128 + cl] + s[192 + cl]);
}

// Small inputs (<= 1 M elements: bias gradients on the 8 x 8 ... 2 x 2 maps of an hourglass, where a launch costs more than the
// sum): ONE launch, a workgroup per 16 columns (4 float4 lanes x 64 row lanes), fixed-order fold.
__global__ __launch_bounds__(256) void col_sum_small_kernel(const float* __restrict__ x, int M, int C, float* __restrict__ out) {
    __shared__ float4 s[256];
    const int t = threadIdx.x, cl = t & 3, rl = t >> 2;
    const int C4 = C >> 2, c4 = blockIdx.x * 4 + cl;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < C4) {
        const float4* base = reinterpret_cast<const float4*>(x) + c4;
        int r = rl;
        for (; r + 7 * 64 < M; r += 8 * 64) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[(int64_t)(r + u * 64) * C4];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; r < M; r += 64) {
            const float4 v = base[(int64_t)r * C4];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    s[t] = acc;
    __syncthreads();
    for (int st = 32; st > 0; st >>= 1) {
        if (rl < st) {
            const float4 v = s[(rl + st) * 4 + cl];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            s[t] = acc;
        }
        __syncthreads();
    }
    if (rl == 0 && c4 < C4) reinterpret_cast<float4*>(out)[c4] = acc;
}

// scalar-column variant for channel counts that are not a multiple of 4
__global__ __launch_bounds__(256) void col_sum_scalar_kernel(const float* __restrict__ x, int64_t M, int C, int cpad,
                                                             int rows_per_wg, float* __restrict__ out, float* __restrict__ part) {
    // part != nullptr: one partial row per workgroup for col_sum_combine_kernel (fixed order); else float atomics into a zeroed out
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
            if (part) part[(int64_t)blockIdx.x * C + cb + c] = tot;
            else atomicAdd(out + cb + c, tot);
        }
        __syncthreads();
    }
}

// channel counts: a power of two from 4 to 1024 (one column block), or a multiple of 1024 (ResNet-50's 2048)
inline bool bn_shape_ok(int C) {
    const int c4n = C >> 2;
    return C >= 4 && (C & 3) == 0 && (c4n <= 256 ? 256 % c4n == 0 : c4n % 256 == 0);
}
inline int bn_col_blocks(int C) { return ((C >> 2) + 255) / 256; }

constexpr int BN_MAX_WGS = 256;
constexpr int BN_ACC_WGS = 512;                                 // reduction workgroups when they meet by atomics (no partial rows to size; 1024 / 2048 measured equal)
inline int bn_rows_per_wg(int64_t M, int C, int max_wgs = BN_MAX_WGS) {
    const int rlanes = 256 / min(C >> 2, 256);
    int64_t r = (M + max_wgs - 1) / max_wgs;
    if (r < (int64_t)rlanes * 8) r = (int64_t)rlanes * 8;      // at least one full batch of loads per row-lane
    return (int)r;
}
inline int bn_apply_grid(int64_t n4, int C) {
    int64_t g = (n4 + 1023) / 1024;                             // >= 4 float4 per thread
    // at most 1024 workgroups (DSF_BN_APPLY_WGS: tuning aid, read per call): 512 .. 1536 measured alike, 2048 costs 0.3-0.5 % of the
    // B = 32 and B = 192 steps, 4096 and 8192 1.5 % of the latter
    const char* cap_e = getenv("DSF_BN_APPLY_WGS");
    const int64_t cap = cap_e ? atoi(cap_e) : 1024;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    const int m = bn_col_blocks(C);                             // the grid stride must be a multiple of the float4 column count
    g = ((g + m - 1) / m) * m;
    return (int)g;
}
// workspace layout: [0, 2C) doubles = channel sums of the backward pass; then BN_MAX_WGS * 2C floats of per-workgroup
// partials.  No state survives a call.
inline float* bn_ws_part(double* ws, int C) { return reinterpret_cast<float*>(ws + 2 * C); }

inline void bn_small_forward(const float* x, const float* residual, const float* gamma, const float* beta, int64_t M, int C, int relu,
                             float* y, const BnFinal& fin, hipStream_t st) {
    const dim3 grid(C >> 2);
    if (M <= 256) hipLaunchKernelGGL(bn_small_fwd_kernel<1>, grid, dim3(256), 0, st, x, residual, gamma, beta, (int)M, C, relu, y, fin);
    else hipLaunchKernelGGL(bn_small_fwd_kernel<4>, grid, dim3(256), 0, st, x, residual, gamma, beta, (int)M, C, relu, y, fin);
}
inline void bn_small_backward(const float* x, const float* grad_y, const float* y, const float* save_mean, const float* save_invstd,
                              const float* gamma, const float* beta, int64_t M, int C, int relu, float* grad_x, float* grad_residual,
                              const BnFinal& fin, hipStream_t st, const float* grad_y2 = nullptr) {
    const dim3 grid(C >> 2);
    if (M <= 256) hipLaunchKernelGGL(bn_small_bwd_kernel<1>, grid, dim3(256), 0, st, x, grad_y, y, save_mean, save_invstd, gamma, beta, (int)M, C, relu, grad_x, grad_residual, fin, grad_y2);
    else hipLaunchKernelGGL(bn_small_bwd_kernel<4>, grid, dim3(256), 0, st, x, grad_y, y, save_mean, save_invstd, gamma, beta, (int)M, C, relu, grad_x, grad_residual, fin, grad_y2);
}

// ---- stem: BatchNorm + ReLU + MaxPool2d as ONE apply pass, and the pooling's backward inside the BatchNorm backward ------------
// The backbone stem is conv5x5 -> BatchNorm -> ReLU -> MaxPool2d(3, 2, 1) (reference model/backbone.py:200-204) on the largest map of
// the step (B x 128 x 128 x 64: 134 MB).  Unfused, the normalised map is written (134 MB), read by the pooling (134 MB), and in the
// backward the pooling's gather writes a 134 MB gradient that both BatchNorm backward passes read.  Here
//  * forward: the apply pass computes the normalised, rectified values of a K x K window from x and writes only the pooled output and
//    the 1-byte argmax (pool.hip's rule: first maximum in (kh, kw) order, NaN wins) -- the full-resolution output never exists;
//  * backward: both passes take the gradient of a full-resolution element from the pooled gradient (bn_pool_gather: the terms of
//    pool.hip's maxpool_bwd_kernel in its order, so the value is bit-for-bit the unfused one), and the ReLU mask is recomputed from
//    x as in the unpooled layer (relu mode 2).
// Traffic: forward 134 R + 42 W instead of 402; backward 2 x (134 + 42) R + 134 W instead of 845.
struct BnPoolP { int Hi, Wi, Ho, Wo, k, stride, pad; };

// the <= 2 x 2 output windows (k <= 2 stride) that contain input pixel `pix`: pooled gradient, argmax bytes, and the window position
// this pixel has in each (255: no such window -- never an argmax byte, k <= 15); addresses of absent windows are clamped onto present ones
struct BnPoolTap { float4 g[4]; uchar4 a[4]; int me[4]; };
__device__ __forceinline__ void bn_pool_fetch(const float* __restrict__ g, const uint8_t* __restrict__ arg, const BnPoolP& p, int C, uint32_t pix,
                                              int c, BnPoolTap& t) {
    const uint32_t row = pix / (uint32_t)p.Wi, b = row / (uint32_t)p.Hi;
    const int ix = (int)(pix - row * (uint32_t)p.Wi), iy = (int)(row - b * (uint32_t)p.Hi);
    const int oy_hi = min((iy + p.pad) / p.stride, p.Ho - 1), ox_hi = min((ix + p.pad) / p.stride, p.Wo - 1);
    const int oy_lo = max(0, (iy + p.pad - p.k + p.stride) / p.stride), ox_lo = max(0, (ix + p.pad - p.k + p.stride) / p.stride);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int oy = oy_lo + (w >> 1), ox = ox_lo + (w & 1);
        const bool v = oy <= oy_hi && ox <= ox_hi;
        const int64_t o = (((int64_t)b * p.Ho + min(oy, p.Ho - 1)) * p.Wo + min(ox, p.Wo - 1)) * C + c;
        t.me[w] = v ? (iy - (oy * p.stride - p.pad)) * p.k + (ix - (ox * p.stride - p.pad)) : 255;
        t.a[w] = *reinterpret_cast<const uchar4*>(arg + o);
        t.g[w] = *reinterpret_cast<const float4*>(g + o);
    }
}
__device__ __forceinline__ float4 bn_pool_gather(const BnPoolTap& t) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        a[0] += (t.a[w].x == t.me[w]) ? t.g[w].x : 0.f; a[1] += (t.a[w].y == t.me[w]) ? t.g[w].y : 0.f;
        a[2] += (t.a[w].z == t.me[w]) ? t.g[w].z : 0.f; a[3] += (t.a[w].w == t.me[w]) ? t.g[w].w : 0.f;
    }
    return make_float4(a[0], a[1], a[2], a[3]);
}

// forward apply: FOLD prologue of bn_apply_kernel (statistics from the accumulation rows; the first workgroup stores mean / invstd and
// updates the running statistics), then a thread per (output pixel, channel quad)
template <int K>
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int64_t n4o, int C, float* __restrict__ y,
                                                               uint8_t* __restrict__ arg, const double* __restrict__ rows, int n_rows,
                                                               int64_t M, BnFinal fin, BnPoolP p) {
    __shared__ double s_buf[BN_FOLD_LDS];
    const int c4n = C >> 2, nq = min(c4n, 256);
    const int64_t i0 = blockIdx.x * (int64_t)256 + threadIdx.x;
    const int64_t S = (int64_t)gridDim.x * 256;
    const int c = (int)(i0 % c4n) * 4;
    const bool owner = (int)threadIdx.x < nq;
    float par[3][4];                                           // mean, scale = invstd * gamma, shift = beta
    double s0[4], s1[4];
    bn_fold_rows_wg(rows, n_rows, C, nq, c, s_buf, s0, s1);
    if (owner) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double m = s0[k] / (double)M;
            double var = s1[k] / (double)M - m * m;
            if (var < 0.0) var = 0.0;
            const float is = (float)(1.0 / sqrt(var + (double)fin.eps));
            par[0][k] = (float)m;
            par[1][k] = is * (gamma ? gamma[c + k] : 1.f);
            par[2][k] = beta ? beta[c + k] : 0.f;
            if (i0 < c4n && fin.mean) {
                fin.mean[c + k] = (float)m;
                fin.invstd[c + k] = is;
                if (fin.running_mean) {
                    const double unbiased = (M > 1) ? var * (double)M / (double)(M - 1) : var;
                    fin.running_mean[c + k] = (1.f - fin.momentum) * fin.running_mean[c + k] + fin.momentum * (float)m;
                    fin.running_var[c + k] = (1.f - fin.momentum) * fin.running_var[c + k] + fin.momentum * (float)unbiased;
                }
            }
        }
    }
    if (nq < 256) bn_share_params<3>(nq, par, s_buf);
    for (int64_t j = i0; j < n4o; j += S) {
        const uint32_t q = (uint32_t)j / (uint32_t)c4n;                   // (the launcher checked n4o < 2^31)
        const uint32_t row = q / (uint32_t)p.Wo, b = row / (uint32_t)p.Ho;
        const int ox = (int)(q - row * (uint32_t)p.Wo), oy = (int)(row - b * (uint32_t)p.Ho);
        float4 xv[K * K];
#pragma unroll
        for (int kh = 0; kh < K; ++kh)
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {                    // (absent taps load a present pixel: clamped address, value unused)
                const int iy = min(max(oy * p.stride - p.pad + kh, 0), p.Hi - 1), ix = min(max(ox * p.stride - p.pad + kw, 0), p.Wi - 1);
                xv[kh * K + kw] = *reinterpret_cast<const float4*>(x + (((int64_t)b * p.Hi + iy) * p.Wi + ix) * C + c);
            }
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int where[4] = {-1, -1, -1, -1};
#pragma unroll
        for (int kh = 0; kh < K; ++kh)
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int iy = oy * p.stride - p.pad + kh, ix = ox * p.stride - p.pad + kw;
                if ((unsigned)iy >= (unsigned)p.Hi || (unsigned)ix >= (unsigned)p.Wi) continue;
                const float4 v = xv[kh * K + kw];
                const float xe[4] = {v.x, v.y, v.z, v.w};
                const int pos = kh * K + kw;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o = fmaxf((xe[e] - par[0][e]) * par[1][e] + par[2][e], 0.f);      // bn_apply_kernel's expression
                    if (where[e] < 0) where[e] = pos;
                    if (o > best[e] || o != o) { best[e] = o; where[e] = pos; }
                }
            }
        *reinterpret_cast<float4*>(y + j * 4) = make_float4(best[0], best[1], best[2], best[3]);
        if (arg) *reinterpret_cast<uchar4*>(arg + j * 4) = make_uchar4((uint8_t)where[0], (uint8_t)where[1], (uint8_t)where[2], (uint8_t)where[3]);
    }
}

// backward sums: bn_reduce_kernel<1> with relu mode 2, the gradient gathered from the pooled gradient; adds into the accumulation rows
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                 const uint8_t* __restrict__ arg, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int64_t M, int C, int rows_per_wg,
                                                                 double* __restrict__ acc, int acc_rows, BnPoolP p) {
    __shared__ float s_part[2][256 * 4];
    const int t = threadIdx.x;
    const int c4b = min(C >> 2, 256);
    const int lcol = t % c4b, rl = t / c4b;
    const int col = blockIdx.y * c4b + lcol;
    const int rlanes = 256 / c4b;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = (r0 + rows_per_wg < M) ? r0 + rows_per_wg : M;
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f}, mu[4], is[4], sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mu[k] = mean[col * 4 + k]; is[k] = invstd[col * 4 + k];
        sc[k] = is[k] * (gamma ? gamma[col * 4 + k] : 1.f);
        sh[k] = beta ? beta[col * 4 + k] : 0.f;
    }
    constexpr int U = 2;
    for (int64_t r = r0 + rl; r < r1; r += (int64_t)U * rlanes) {
        float4 xv[U]; BnPoolTap tap[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t rr = r + (int64_t)u * rlanes;
            const int64_t rc = (rr < r1) ? rr : r;
            xv[u] = *reinterpret_cast<const float4*>(x + rc * C + col * 4);
            bn_pool_fetch(g, arg, p, C, (uint32_t)rc, col * 4, tap[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + (int64_t)u * rlanes >= r1) break;
            const float4 gv = bn_pool_gather(tap[u]);
            const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
            float ge[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) ge[k] = ((xe[k] - mu[k]) * sc[k] + sh[k] > 0.f) ? ge[k] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0[k] += ge[k]; a1[k] = fmaf(ge[k], (xe[k] - mu[k]) * is[k], a1[k]); }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s_part[0][t * 4 + k] = a0[k]; s_part[1][t * 4 + k] = a1[k]; }
    __syncthreads();
    double* out = acc + (int64_t)(blockIdx.x % acc_rows) * 2 * C + blockIdx.y * c4b * 4;
    for (int ch = t; ch < c4b * 4; ch += 256) {
        const int cc = ch >> 2, kk = ch & 3;
        float d0 = 0.f, d1 = 0.f;
        for (int q = 0; q < rlanes; ++q) { d0 += s_part[0][(q * c4b + cc) * 4 + kk]; d1 += s_part[1][(q * c4b + cc) * 4 + kk]; }
        __hip_atomic_fetch_add(out + ch, (double)d0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(out + C + ch, (double)d1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// backward apply: bn_bwd_apply_kernel<FOLD> with relu mode 2 and the gathered gradient; x is read once (non-temporal), dx written
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                const uint8_t* __restrict__ arg, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int64_t M, int64_t n4, int C,
                                                                float* __restrict__ dx, const double* __restrict__ rows, int n_rows,
                                                                BnFinal fin, BnPoolP p) {
    __shared__ double s_buf[BN_FOLD_LDS];
    constexpr int U = 2;
    const int c4n = C >> 2, nq = min(c4n, 256);
    const int64_t i0 = blockIdx.x * (int64_t)256 + threadIdx.x;
    const int64_t S = (int64_t)gridDim.x * 256;
    const int c = (int)(i0 % c4n) * 4;
    const bool owner = (int)threadIdx.x < nq;
    float4 xv[U]; BnPoolTap tap[U];
    auto fetch = [&](int64_t i) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = (i + u * S < n4) ? i + u * S : i0;
            xv[u] = bn_ld4<true>(x, j);
            bn_pool_fetch(g, arg, p, C, (uint32_t)j / (uint32_t)c4n, c, tap[u]);      // (n4 < 2^31: checked by the launcher)
        }
    };
    fetch(i0);
    double f0[4], f1[4];
    bn_fold_rows_wg(rows, n_rows, C, nq, c, s_buf, f0, f1);
    if (owner && i0 < c4n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (fin.dbeta) fin.dbeta[c + k] = (fin.accum ? fin.dbeta[c + k] : 0.f) + (float)f0[k];
            if (fin.dgamma) fin.dgamma[c + k] = (fin.accum ? fin.dgamma[c + k] : 0.f) + (float)f1[k];
        }
    }
    float par[6][4];                                           // mean, invstd, sum_g / M, sum_gx / M, invstd * gamma, beta
    if (owner) {
        const float invM = 1.0f / (float)M;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            par[0][k] = mean[c + k]; par[1][k] = invstd[c + k];
            par[2][k] = (float)f0[k] * invM; par[3][k] = (float)f1[k] * invM;
            par[4][k] = (gamma ? gamma[c + k] : 1.f) * par[1][k];
            par[5][k] = beta ? beta[c + k] : 0.f;
        }
    }
    if (nq < 256) bn_share_params<6>(nq, par, s_buf);
    const int64_t US = (int64_t)U * S;
    for (int64_t i = i0; i < n4;) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = i + u * S;
            if (j >= n4) break;
            const float4 gv = bn_pool_gather(tap[u]);
            const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
            float ge[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) ge[k] = ((xe[k] - par[0][k]) * par[4][k] + par[5][k] > 0.f) ? ge[k] : 0.f;
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (xe[k] - par[0][k]) * par[1][k];
                o[k] = par[4][k] * (ge[k] - par[2][k] - xh * par[3][k]);
            }
            bn_st4<false>(dx, j, make_float4(o[0], o[1], o[2], o[3]));
        }
        i += US;
        if (i < n4) fetch(i);
    }
}

// ---- launch helpers: the VAR / G2 / mask instantiations ---------------------------------------------------------------------
#define BN_VAR_SWITCH(var, CALL) \
    switch (var) { case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; \
                   case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; default: CALL(0); break; }

template <bool FOLD>
inline void bn_launch_apply(int grid_x, int grid_y, hipStream_t st, const float* x, const float* res, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, int64_t n4, int C, int relu, float* y, const double* rows, int n_rows,
                            int64_t M, const BnFinal& fin) {
#define BN_CALL(V) hipLaunchKernelGGL((bn_apply_kernel<FOLD, V>), dim3(grid_x, grid_y), dim3(256), 0, st, x, res, mean, invstd, gamma, beta, n4, C, relu, y, rows, n_rows, M, fin)
    BN_VAR_SWITCH(bn_var(), BN_CALL)
#undef BN_CALL
}

template <bool FOLD>
inline void bn_launch_bwd_apply(int grid_x, hipStream_t st, const float* x, const float* gy, const float* gy2, const float* y, const float* mean,
                                const float* invstd, const float* gamma, const float* beta, const double* acc, int64_t M, int64_t n4, int C,
                                int relu, float* dx, float* dres, const double* rows, int n_rows, const BnFinal& fin, const double* count) {
    if (gy2) {
#define BN_CALL(V) hipLaunchKernelGGL((bn_bwd_apply_kernel<FOLD, true, V>), dim3(grid_x), dim3(256), 0, st, x, gy, y, mean, invstd, gamma, beta, acc, M, n4, C, relu, dx, dres, rows, n_rows, fin, count, gy2)
        BN_VAR_SWITCH(bn_var(), BN_CALL)
#undef BN_CALL
    } else {
#define BN_CALL(V) hipLaunchKernelGGL((bn_bwd_apply_kernel<FOLD, false, V>), dim3(grid_x), dim3(256), 0, st, x, gy, y, mean, invstd, gamma, beta, acc, M, n4, C, relu, dx, dres, rows, n_rows, fin, count, gy2)
        BN_VAR_SWITCH(bn_var(), BN_CALL)
#undef BN_CALL
    }
}

// backward sums pass: relu 1 -> mask from y; gy2 -> two addends; gout -> the masked gradient is written (and the apply pass reads it)
inline void bn_launch_bwd_reduce(dim3 grid, hipStream_t st, const float* x, const float* gy, const float* gy2, const float* y, const float* mean,
                                 const float* invstd, const float* gamma, const float* beta, int64_t M, int C, int relu, int rows, float* part,
                                 int acc_rows, float* gout) {
#define BN_RED(YM, G2, WG) hipLaunchKernelGGL((bn_reduce_kernel<1, YM, G2, WG>), grid, dim3(256), 0, st, x, gy, y, mean, invstd, gamma, beta, M, C, relu, rows, part, acc_rows, gy2, gout)
    const int sel = (relu == 1 ? 4 : 0) | (gy2 ? 2 : 0) | (gout ? 1 : 0);
    switch (sel) {
        case 0: BN_RED(false, false, false); break;
        case 1: BN_RED(false, false, true); break;
        case 2: BN_RED(false, true, false); break;
        case 3: BN_RED(false, true, true); break;
        case 4: BN_RED(true, false, false); break;
        case 5: BN_RED(true, false, true); break;
        case 6: BN_RED(true, true, false); break;
        default: BN_RED(true, true, true); break;
    }
#undef BN_RED
}

}  // namespace

extern "C" int64_t dsf_bn_workspace_bytes(int C) {
    return (int64_t)(2 * C) * 8 + (int64_t)BN_MAX_WGS * 2 * C * 4 + 16;
}

extern "C" int dsf_bn_forward(const float* x, const float* residual, const float* gamma, const float* beta, int64_t M,
                              int C, float eps, float momentum, int relu, float* running_mean, float* running_var,
                              float* y, float* save_mean, float* save_invstd, double* workspace, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && y && save_mean && save_invstd && workspace && M > 0);
    if (!bn_shape_ok(C)) return DSF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (bn_small_ok(M, C)) {                                         // small maps: one launch, fixed-order sums
        BnFinal fin = {eps, momentum, save_mean, save_invstd, running_mean, running_var, nullptr, nullptr, nullptr};
        bn_small_forward(x, residual, gamma, beta, M, C, relu, y, fin, st);
        return dsf_launch_status();
    }
    const int rows = bn_rows_per_wg(M, C);
    const int wgs = (int)((M + rows - 1) / rows);
    BnFinal fin = {eps, momentum, save_mean, save_invstd, running_mean, running_var, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL((bn_reduce_kernel<0, false>), dim3(wgs, bn_col_blocks(C)), dim3(256), 0, st, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                       M, C, 0, rows, bn_ws_part(workspace, C), 0);
    hipLaunchKernelGGL(bn_finalize_kernel<0>, dim3((C + 15) / 16), dim3(256), 0, st, bn_ws_part(workspace, C), wgs, M, C, fin);
    const int64_t n4 = M * (C >> 2);
    bn_launch_apply<false>(bn_apply_grid(n4, C), 1, st, x, residual, save_mean, save_invstd, gamma,
                       beta, n4, C, relu, y, nullptr, 0, M, fin);
    return dsf_launch_status();
}

// Forward with the statistics pass already done: `part` holds `rows` partial rows [row][2][C] (sum, sum of squares) written
// by the producing convolution's epilogue (dsf_conv_x6_forward_bn) -- two launches instead of three, one read of x less.
extern "C" int dsf_bn_forward_from_stats(const float* x, const float* residual, const float* gamma, const float* beta, int64_t M,
                                         int C, float eps, float momentum, int relu, float* running_mean, float* running_var,
                                         float* y, float* save_mean, float* save_invstd, const float* part, int rows,
                                         dsf_stream_t stream) {
    DSF_CHECK_ARG(x && y && save_mean && save_invstd && part && rows > 0 && M > 0);
    if (!bn_shape_ok(C)) return DSF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    BnFinal fin = {eps, momentum, save_mean, save_invstd, running_mean, running_var, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(bn_finalize_kernel<0>, dim3((C + 15) / 16), dim3(256), 0, st, part, rows, M, C, fin);
    const int64_t n4 = M * (C >> 2);
    bn_launch_apply<false>(bn_apply_grid(n4, C), 1, st, x, residual, save_mean, save_invstd, gamma,
                       beta, n4, C, relu, y, nullptr, 0, M, fin);
    return dsf_launch_status();
}

// inference / frozen statistics: y = (x - mean) * invstd * gamma + beta with given mean / invstd
extern "C" int dsf_bn_apply(const float* x, const float* residual, const float* gamma, const float* beta,
                            const float* mean, const float* invstd, int64_t M, int C, int relu, float* y,
                            dsf_stream_t stream) {
    DSF_CHECK_ARG(x && y && mean && invstd && M > 0);
    if (!bn_shape_ok(C)) return DSF_ERR_UNSUPPORTED;
    const int64_t n4 = M * (C >> 2);
    BnFinal fin = {0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bn_launch_apply<false>(bn_apply_grid(n4, C), 1, (hipStream_t)stream, x, residual, mean, invstd,
                       gamma, beta, n4, C, relu, y, nullptr, 0, M, fin);
    return dsf_launch_status();
}

// grad_y2 (may be NULL): second addend of the incoming gradient, g = grad_y + grad_y2.  With a residual gradient wanted the sums
// pass writes the masked g into grad_residual and the apply pass reads it back (x + g in, dx out) instead of re-reading
// grad_y (+ grad_y2) and y: add + sums + apply were 8R + 3W of the activation, now 6R + 2W (5R + 2W with one addend, was 6R + 2W).
static int bn_backward_ordered(const float* x, const float* grad_y, const float* grad_y2, const float* y, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, int64_t M, int C, int relu,
                               float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta,
                               double* workspace, hipStream_t st, int accumulate_affine = 0) {
    const int rows = bn_rows_per_wg(M, C);
    const int wgs = (int)((M + rows - 1) / rows);
    BnFinal fin = {0.f, 0.f, nullptr, nullptr, nullptr, nullptr, workspace, grad_gamma, grad_beta, accumulate_affine};
    if (bn_small_ok(M, C)) {
        bn_small_backward(x, grad_y, y, save_mean, save_invstd, gamma, beta, M, C, relu, grad_x, grad_residual, fin, st, grad_y2);
        return dsf_launch_status();
    }
    float* gout = bn_write_g() ? grad_residual : nullptr;
    bn_launch_bwd_reduce(dim3(wgs, bn_col_blocks(C)), st, x, grad_y, grad_y2, y, save_mean, save_invstd, gamma, beta, M, C, relu, rows,
                         bn_ws_part(workspace, C), 0, gout);
    hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3((C + 15) / 16), dim3(256), 0, st, bn_ws_part(workspace, C), wgs, M, C, fin);
    const int64_t n4 = M * (C >> 2);
    if (gout)
        bn_launch_bwd_apply<false>(bn_apply_grid(n4, C), st, x, gout, nullptr, nullptr, save_mean, save_invstd, gamma, beta, workspace, M, n4, C, 0,
                                   grad_x, nullptr, nullptr, 0, fin, nullptr);
    else
        bn_launch_bwd_apply<false>(bn_apply_grid(n4, C), st, x, grad_y, grad_y2, y, save_mean, save_invstd, gamma, beta, workspace, M, n4, C, relu,
                                   grad_x, grad_residual, nullptr, 0, fin, nullptr);
    return dsf_launch_status();
}

extern "C" int dsf_bn_backward(const float* x, const float* grad_y, const float* y, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, int64_t M, int C, int relu,
                               float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta,
                               double* workspace, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && grad_y && save_mean && save_invstd && grad_x && workspace && M > 0 && relu >= 0 && relu <= 2 &&
                  (relu != 1 || y));
    if (!bn_shape_ok(C)) return DSF_ERR_UNSUPPORTED;
    return bn_backward_ordered(x, grad_y, nullptr, y, gamma, beta, save_mean, save_invstd, M, C, relu, grad_x, grad_residual, grad_gamma,
                               grad_beta, workspace, (hipStream_t)stream);
}

extern "C" int dsf_bn_backward_pair(const float* x, const float* grad_y, const float* grad_y2, const float* y, const float* gamma,
                                    const float* beta, const float* save_mean, const float* save_invstd, int64_t M, int C, int relu,
                                    float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta, int accumulate_affine,
                                    double* workspace, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && grad_y && save_mean && save_invstd && grad_x && workspace && M > 0 && relu >= 0 && relu <= 2 &&
                  (relu != 1 || y));
    if (!bn_shape_ok(C)) return DSF_ERR_UNSUPPORTED;
    return bn_backward_ordered(x, grad_y, grad_y2, y, gamma, beta, save_mean, save_invstd, M, C, relu, grad_x, grad_residual, grad_gamma,
                               grad_beta, workspace, (hipStream_t)stream, accumulate_affine);
}

// ---- the same passes WITHOUT the finalise launches (default, float-atomic mode) --------------------------------------
// `acc` is a caller-zeroed block of dsf_bn_acc_rows() rows [row][2][C] of DOUBLES: the statistics pass (here, or the producing
// convolution's epilogue: dsf_conv_x6_forward_bn_acc) adds its per-workgroup float sums into row (workgroup mod rows) with
// double atomics, and the apply kernel folds the rows in its own prologue (in double, ascending) -- forward = 1 launch after a
// convolution that filled the rows (2 otherwise), backward = 2, where the ordered-partials path above needs 2-3 and 3.
// Results differ from that path by the order of DOUBLE additions only (~1e-16 relative on the sums); in
// deterministic mode these entry points return DSF_ERR_UNSUPPORTED and the caller uses the ordered path.
// 8 rows of doubles: 512 reduce workgroups = 64 adders per address, as a 512-workgroup convolution epilogue -- a few hundred KB of atomics per launch, spread over its duration (measured with 32 float rows: the
// 64-load fold cost more than the finalise launch it replaced)
constexpr int BN_ACC_ROWS = 8;
extern "C" int dsf_bn_acc_rows(void) { return BN_ACC_ROWS; }

extern "C" int dsf_bn_forward_acc(const float* x, const float* residual, const float* gamma, const float* beta, int64_t M, int C,
                                  float eps, float momentum, int relu, float* running_mean, float* running_var, float* y,
                                  float* save_mean, float* save_invstd, double* acc, int acc_filled, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && y && save_mean && save_invstd && acc && M > 0);
    if (!bn_shape_ok(C) || dsf_deterministic()) return DSF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    BnFinal fin = {eps, momentum, save_mean, save_invstd, running_mean, running_var, nullptr, nullptr, nullptr};
    if (!acc_filled && bn_small_ok(M, C)) {                          // statistics + apply in one launch (`acc` stays untouched)
        bn_small_forward(x, residual, gamma, beta, M, C, relu, y, fin, st);
        return dsf_launch_status();
    }
    if (!acc_filled) {
        const int rows = bn_rows_per_wg(M, C, BN_ACC_WGS);
        const int wgs = (int)((M + rows - 1) / rows);
        hipLaunchKernelGGL((bn_reduce_kernel<0, false>), dim3(wgs, bn_col_blocks(C)), dim3(256), 0, st, x, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, M, C, 0, rows, reinterpret_cast<float*>(acc), BN_ACC_ROWS);
    }
    const int64_t n4 = M * (C >> 2);
    // TIMING PROBE (DSF_BN_PROBE=skip_single_apply, read per call; results are WRONG): the apply pass of a BatchNorm whose statistics
    // came from the convolution's epilogue and whose output has no residual -- the passes that a BatchNorm(+ReLU) in the consuming
    // convolution's loader would remove -- is not launched: the whole-step time without them bounds that fusion from above
    // (profiles/r06_bn_loader_bound.txt).
    if (acc_filled && !residual) {
        const char* pr = getenv("DSF_BN_PROBE");
        if (pr && pr[0] == 's') return DSF_OK;
    }
    bn_launch_apply<true>(bn_apply_grid(n4, C), 1, st, x, residual, nullptr, nullptr, gamma, beta,
                       n4, C, relu, y, acc, BN_ACC_ROWS, M, fin);
    return dsf_launch_status();
}

static int bn_backward_acc_impl(const float* x, const float* grad_y, const float* grad_y2, const float* y, const float* gamma, const float* beta,
                                const float* save_mean, const float* save_invstd, int64_t M, int C, int relu, float* grad_x,
                                float* grad_residual, float* grad_gamma, float* grad_beta, double* acc, hipStream_t st,
                                int accumulate_affine = 0) {
    // 1024 reduction workgroups (DSF_BN_BWD_WGS: tuning aid, read per call): on the B = 192 tensors (800 MB) 512 left the pass at
    // 2.7-3.4 TB/s -- config 4 165.4 -> 163.5 ms per step with 1024, 2048 and 4096 alike; the B = 32 step does not care.  Eight rows
    // in flight per lane instead of four: no gain (162.5 vs 163.6 ms)
    const char* wg_e = getenv("DSF_BN_BWD_WGS");
    int max_wgs = wg_e ? atoi(wg_e) : 1024;
    if (max_wgs < 1) max_wgs = 1024;                                 // (0 / garbage would divide by zero in bn_rows_per_wg)
    const int rows = bn_rows_per_wg(M, C, max_wgs);
    const int wgs = (int)((M + rows - 1) / rows);
    BnFinal fin = {0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, grad_gamma, grad_beta, accumulate_affine};
    if (bn_small_ok(M, C)) {                                         // both sums + apply in one launch (`acc` stays untouched)
        bn_small_backward(x, grad_y, y, save_mean, save_invstd, gamma, beta, M, C, relu, grad_x, grad_residual, fin, st, grad_y2);
        return dsf_launch_status();
    }
    float* gout = bn_write_g() ? grad_residual : nullptr;
    bn_launch_bwd_reduce(dim3(wgs, bn_col_blocks(C)), st, x, grad_y, grad_y2, y, save_mean, save_invstd, gamma, beta, M, C, relu, rows,
                         reinterpret_cast<float*>(acc), BN_ACC_ROWS, gout);
    const int64_t n4 = M * (C >> 2);
    if (gout)
        bn_launch_bwd_apply<true>(bn_apply_grid(n4, C), st, x, gout, nullptr, nullptr, save_mean, save_invstd, gamma, beta, nullptr, M, n4, C, 0,
                                  grad_x, nullptr, acc, BN_ACC_ROWS, fin, nullptr);
    else
        bn_launch_bwd_apply<true>(bn_apply_grid(n4, C), st, x, grad_y, grad_y2, y, save_mean, save_invstd, gamma, beta, nullptr, M, n4, C, relu,
                                  grad_x, grad_residual, acc, BN_ACC_ROWS, fin, nullptr);
    return dsf_launch_status();
}

extern "C" int dsf_bn_backward_acc(const float* x, const float* grad_y, const float* y, const float* gamma, const float* beta,
                                   const float* save_mean, const float* save_invstd, int64_t M, int C, int relu, float* grad_x,
                                   float* grad_residual, float* grad_gamma, float* grad_beta, double* acc, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && grad_y && save_mean && save_invstd && acc && M > 0 && relu >= 0 && relu <= 2 && (relu != 1 || y));
    if (!bn_shape_ok(C) || dsf_deterministic()) return DSF_ERR_UNSUPPORTED;
    if (!grad_x) {                                                    // sums only (dsf_conv_c1_wrw_bn applies them); not on the one-launch small maps
        if (bn_small_ok(M, C) || grad_residual) return DSF_ERR_UNSUPPORTED;
        const char* wg_e = getenv("DSF_BN_BWD_WGS");
        int max_wgs = wg_e ? atoi(wg_e) : 1024;
        if (max_wgs < 1) max_wgs = 1024;
        const int rows = bn_rows_per_wg(M, C, max_wgs);
        bn_launch_bwd_reduce(dim3((unsigned)((M + rows - 1) / rows), bn_col_blocks(C)), (hipStream_t)stream, x, grad_y, nullptr, y, save_mean, save_invstd,
                             gamma, beta, M, C, relu, rows, reinterpret_cast<float*>(acc), BN_ACC_ROWS, nullptr);
        return dsf_launch_status();
    }
    return bn_backward_acc_impl(x, grad_y, nullptr, y, gamma, beta, save_mean, save_invstd, M, C, relu, grad_x, grad_residual, grad_gamma,
                                grad_beta, acc, (hipStream_t)stream);
}

extern "C" int dsf_bn_backward_acc_pair(const float* x, const float* grad_y, const float* grad_y2, const float* y, const float* gamma,
                                        const float* beta, const float* save_mean, const float* save_invstd, int64_t M, int C, int relu,
                                        float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta, int accumulate_affine,
                                        double* acc, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && grad_y && save_mean && save_invstd && grad_x && acc && M > 0 && relu >= 0 && relu <= 2 && (relu != 1 || y));
    if (!bn_shape_ok(C) || dsf_deterministic()) return DSF_ERR_UNSUPPORTED;
    return bn_backward_acc_impl(x, grad_y, grad_y2, y, gamma, beta, save_mean, save_invstd, M, C, relu, grad_x, grad_residual, grad_gamma,
                                grad_beta, acc, (hipStream_t)stream, accumulate_affine);
}

// ---- stem: BatchNorm + ReLU + MaxPool2d (see bn_relu_pool_fwd_kernel) ----------------------------------------------------------
static bool bn_pool_ok(int B, int Hi, int Wi, int C, int k, int stride, int pad, int* Ho, int* Wo) {
    if (!(B > 0 && Hi > 0 && Wi > 0 && bn_shape_ok(C) && (k == 2 || k == 3) && stride >= 1 && pad >= 0 && 2 * pad <= k && k <= 2 * stride)) return false;
    *Ho = (Hi + 2 * pad - k) / stride + 1; *Wo = (Wi + 2 * pad - k) / stride + 1;
    return *Ho > 0 && *Wo > 0 && (int64_t)B * Hi * Wi * (C >> 2) < ((int64_t)1 << 31);
}

extern "C" int dsf_bn_relu_pool_forward(const float* x, const float* gamma, const float* beta, int B, int Hi, int Wi, int C, int k, int stride,
                                        int pad, float eps, float momentum, float* running_mean, float* running_var, float* y,
                                        uint8_t* argmax, float* save_mean, float* save_invstd, double* acc, int acc_filled,
                                        dsf_stream_t stream) {
    DSF_CHECK_ARG(x && y && save_mean && save_invstd && acc);
    int Ho, Wo;
    if (!bn_pool_ok(B, Hi, Wi, C, k, stride, pad, &Ho, &Wo) || dsf_deterministic()) return DSF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int64_t M = (int64_t)B * Hi * Wi;
    BnFinal fin = {eps, momentum, save_mean, save_invstd, running_mean, running_var, nullptr, nullptr, nullptr};
    if (!acc_filled) {                                                // (else: the producing convolution's epilogue left the sums in `acc`)
        const int rows = bn_rows_per_wg(M, C, BN_ACC_WGS);
        const int wgs = (int)((M + rows - 1) / rows);
        hipLaunchKernelGGL((bn_reduce_kernel<0, false>), dim3(wgs, bn_col_blocks(C)), dim3(256), 0, st, x, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, M, C, 0, rows, reinterpret_cast<float*>(acc), BN_ACC_ROWS);
    }
    const BnPoolP p = {Hi, Wi, Ho, Wo, k, stride, pad};
    const int64_t n4o = (int64_t)B * Ho * Wo * (C >> 2);
    const int grid = bn_apply_grid(n4o * 4, C);                       // (a thread's unit of work is a K x K window, not one float4)
    if (k == 3) hipLaunchKernelGGL(bn_relu_pool_fwd_kernel<3>, dim3(grid), dim3(256), 0, st, x, gamma, beta, n4o, C, y, argmax, acc, BN_ACC_ROWS, M, fin, p);
    else hipLaunchKernelGGL(bn_relu_pool_fwd_kernel<2>, dim3(grid), dim3(256), 0, st, x, gamma, beta, n4o, C, y, argmax, acc, BN_ACC_ROWS, M, fin, p);
    return dsf_launch_status();
}

extern "C" int dsf_bn_relu_pool_backward(const float* x, const float* grad_y, const uint8_t* argmax, const float* gamma, const float* beta,
                                         const float* save_mean, const float* save_invstd, int B, int Hi, int Wi, int C, int k, int stride,
                                         int pad, float* grad_x, float* grad_gamma, float* grad_beta, int accumulate_affine, double* acc,
                                         dsf_stream_t stream) {
    DSF_CHECK_ARG(x && grad_y && argmax && save_mean && save_invstd && acc);
    int Ho, Wo;
    if (!bn_pool_ok(B, Hi, Wi, C, k, stride, pad, &Ho, &Wo) || dsf_deterministic()) return DSF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int64_t M = (int64_t)B * Hi * Wi, n4 = M * (C >> 2);
    const BnPoolP p = {Hi, Wi, Ho, Wo, k, stride, pad};
    const char* wg_e = getenv("DSF_BN_BWD_WGS");
    int max_wgs = wg_e ? atoi(wg_e) : 1024;
    if (max_wgs < 1) max_wgs = 1024;
    const int rows = bn_rows_per_wg(M, C, max_wgs);
    const int wgs = (int)((M + rows - 1) / rows);
    BnFinal fin = {0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, grad_gamma, grad_beta, accumulate_affine};
    hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3(wgs, bn_col_blocks(C)), dim3(256), 0, st, x, grad_y, argmax, save_mean, save_invstd, gamma,
                       beta, M, C, rows, acc, BN_ACC_ROWS, p);
    if (!grad_x) return dsf_launch_status();                          // sums only: dsf_conv_c1_wrw_bn takes it from here
    hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3(bn_apply_grid(n4, C)), dim3(256), 0, st, x, grad_y, argmax, save_mean, save_invstd, gamma,
                       beta, M, n4, C, grad_x, acc, BN_ACC_ROWS, fin, p);
    return dsf_launch_status();
}

// ---- cross-replica BatchNorm (SyncBatchNorm): ONE exchange of 2C + 1 doubles per layer and pass -------------
// forward: dsf_bn_local_sums -> the caller all-reduces [sum x | sum x^2 | count] -> dsf_bn_forward_from_sums;
// backward: dsf_bn_backward_sums -> all-reduce of [sum g | sum g xhat] -> dsf_bn_backward_apply with the global count.
// mean / invstd / running statistics from GLOBAL sums (sums[0..C), sums[C..2C), count = sums[2C]); one thread per channel
__global__ __launch_bounds__(256) void bn_stats_from_sums_kernel(const double* __restrict__ sums, int C, BnFinal fin) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double n = sums[2 * C];
    const double m = sums[c] / n;
    double var = sums[C + c] / n - m * m;
    if (var < 0.0) var = 0.0;
    fin.mean[c] = (float)m;
    fin.invstd[c] = (float)(1.0 / sqrt(var + (double)fin.eps));
    if (fin.running_mean) {
        const double unbiased = (n > 1.0) ? var * n / (n - 1.0) : var;
        fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * (float)m;
        fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unbiased;
    }
}

extern "C" int dsf_bn_local_sums(const float* x, int64_t M, int C, const float* part, int rows, double* sums, double* workspace,
                                 dsf_stream_t stream) {
    DSF_CHECK_ARG(x && sums && M > 0 && (part ? rows > 0 : workspace != nullptr));
    if (!bn_shape_ok(C)) return DSF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    BnFinal fin = {0.f, 0.f, nullptr, nullptr, nullptr, nullptr, sums, nullptr, nullptr};
    if (!part) {                                                    // no epilogue rows: this call's own statistics pass
        const int rpw = bn_rows_per_wg(M, C);
        rows = (int)((M + rpw - 1) / rpw);
        hipLaunchKernelGGL((bn_reduce_kernel<0, false>), dim3(rows, bn_col_blocks(C)), dim3(256), 0, st, x, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, M, C, 0, rpw, bn_ws_part(workspace, C), 0);
        part = bn_ws_part(workspace, C);
    }
    hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3((C + 15) / 16), dim3(256), 0, st, part, rows, M, C, fin);   // MODE 1: the two sums as doubles
    return dsf_launch_status();
}

extern "C" int dsf_bn_forward_from_sums(const float* x, const float* residual, const float* gamma, const float* beta, int64_t M, int C,
                                        float eps, float momentum, int relu, float* running_mean, float* running_var, float* y,
                                        float* save_mean, float* save_invstd, const double* sums, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && y && save_mean && save_invstd && sums && M > 0);
    if (!bn_shape_ok(C)) return DSF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    BnFinal fin = {eps, momentum, save_mean, save_invstd, running_mean, running_var, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(bn_stats_from_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, C, fin);
    const int64_t n4 = M * (C >> 2);
    bn_launch_apply<false>(bn_apply_grid(n4, C), 1, st, x, residual, save_mean, save_invstd, gamma,
                       beta, n4, C, relu, y, nullptr, 0, M, fin);
    return dsf_launch_status();
}

extern "C" int dsf_bn_backward_sums(const float* x, const float* grad_y, const float* y, const float* gamma, const float* beta,
                                    const float* save_mean, const float* save_invstd, int64_t M, int C, int relu, double* sums,
                                    float* grad_gamma, float* grad_beta, double* workspace, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && grad_y && save_mean && save_invstd && sums && workspace && M > 0 && relu >= 0 && relu <= 2 && (relu != 1 || y));
    if (!bn_shape_ok(C)) return DSF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int rows = bn_rows_per_wg(M, C);
    const int wgs = (int)((M + rows - 1) / rows);
    BnFinal fin = {0.f, 0.f, nullptr, nullptr, nullptr, nullptr, sums, grad_gamma, grad_beta};      // dgamma / dbeta: this replica's share
    bn_launch_bwd_reduce(dim3(wgs, bn_col_blocks(C)), st, x, grad_y, nullptr, y, save_mean, save_invstd, gamma, beta, M, C, relu, rows,
                         bn_ws_part(workspace, C), 0, nullptr);
    hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3((C + 15) / 16), dim3(256), 0, st, bn_ws_part(workspace, C), wgs, M, C, fin);
    return dsf_launch_status();
}

extern "C" int dsf_bn_backward_apply(const float* x, const float* grad_y, const float* y, const float* gamma, const float* beta,
                                     const float* save_mean, const float* save_invstd, const double* sums, const double* count,
                                     int64_t M, int C, int relu, float* grad_x, float* grad_residual, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && grad_y && save_mean && save_invstd && sums && count && grad_x && M > 0 && relu >= 0 && relu <= 2 && (relu != 1 || y));
    if (!bn_shape_ok(C)) return DSF_ERR_UNSUPPORTED;
    BnFinal fin = {0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const int64_t n4 = M * (C >> 2);
    bn_launch_bwd_apply<false>(bn_apply_grid(n4, C), (hipStream_t)stream, x, grad_y, nullptr, y, save_mean, save_invstd, gamma, beta, sums, M, n4, C,
                               relu, grad_x, grad_residual, nullptr, 0, fin, count);
    return dsf_launch_status();
}

// ---- InstanceNorm2d (affine = False, no running statistics) (+ residual) (+ ReLU) on NHWC, inference ------------------------
// (the frozen Consis-CycleGAN generator, reference render_model/transfer.py:393-448: 23 of them per 128x128 image).  x (B, HW, C):
// one statistics pass with the sample as a grid dimension (per-sample sums added into acc [B][1][2][C] doubles, caller-zeroed)
// and one apply pass that turns them into mean / invstd in its prologue.  torch runs it as batch_norm over a (1, B C, H, W) view:
// two kernels plus two layout copies around them, ~10x the time.
extern "C" int dsf_instnorm_forward(const float* x, const float* residual, int B, int64_t HW, int C, float eps, int relu, float* y,
                                    double* acc, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && y && acc && B >= 0 && HW > 0);
    if (!bn_shape_ok(C) || B > 65535) return DSF_ERR_UNSUPPORTED;
    if (B == 0) return DSF_OK;
    hipStream_t st = (hipStream_t)stream;
    const int rows = bn_rows_per_wg(HW, C);
    const int wgs = (int)((HW + rows - 1) / rows);
    hipLaunchKernelGGL((bn_reduce_kernel<0, false>), dim3(wgs, bn_col_blocks(C), B), dim3(256), 0, st, x, nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr, HW, C, 0, rows, reinterpret_cast<float*>(acc), 1);
    BnFinal fin = {eps, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const int64_t n4 = HW * (C >> 2);
    bn_launch_apply<true>(bn_apply_grid(n4, C), B, st, x, residual, nullptr, nullptr, nullptr,
                       nullptr, n4, C, relu, y, acc, 1, HW, fin);
    return dsf_launch_status();
}

// ---- nn.ReflectionPad2d on NHWC (the generator's padding, transfer.py:409,428-444): y (B, H + 2p, W + 2p, C) ---------------------
__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C4,
                                                          int pad, int64_t n4) {
    const int Ho = H + 2 * pad, Wo = W + 2 * pad;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        int64_t q = i / C4;
        const int ox = (int)(q % Wo); q /= Wo;
        const int oy = (int)(q % Ho); const int64_t b = q / Ho;
        int iy = oy - pad, ix = ox - pad;
        iy = iy < 0 ? -iy : (iy >= H ? 2 * H - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= W ? 2 * W - 2 - ix : ix);
        reinterpret_cast<float4*>(y)[i] = reinterpret_cast<const float4*>(x)[((b * H + iy) * W + ix) * C4 + c];
    }
}
__global__ __launch_bounds__(256) void reflect_pad_scalar_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C,
                                                                 int pad, int64_t n) {
    const int Ho = H + 2 * pad, Wo = W + 2 * pad;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t q = i / C;
        const int ox = (int)(q % Wo); q /= Wo;
        const int oy = (int)(q % Ho); const int64_t b = q / Ho;
        int iy = oy - pad, ix = ox - pad;
        iy = iy < 0 ? -iy : (iy >= H ? 2 * H - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= W ? 2 * W - 2 - ix : ix);
        y[i] = x[((b * H + iy) * W + ix) * C + c];
    }
}
extern "C" int dsf_reflect_pad_nhwc(const float* x, float* y, int B, int H, int W, int C, int pad, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && y && B >= 0 && H > 0 && W > 0 && C > 0 && pad >= 0 && pad < H && pad < W);
    if (B == 0) return DSF_OK;
    const int64_t n = (int64_t)B * (H + 2 * pad) * (W + 2 * pad) * C;
    if ((C & 3) == 0) {
        const int64_t n4 = n / 4;
        hipLaunchKernelGGL(reflect_pad_kernel, dim3((unsigned)((n4 + 1023) / 1024 > 8192 ? 8192 : (n4 + 1023) / 1024)), dim3(256), 0,
                           (hipStream_t)stream, x, y, H, W, C >> 2, pad, n4);
    } else {
        hipLaunchKernelGGL(reflect_pad_scalar_kernel, dim3((unsigned)((n + 1023) / 1024 > 8192 ? 8192 : (n + 1023) / 1024)), dim3(256), 0,
                           (hipStream_t)stream, x, y, H, W, C, pad, n);
    }
    return dsf_launch_status();
}

constexpr int COLSUM_MAX_WGS = 256;
extern "C" int64_t dsf_col_sum_workspace_bytes(int C) { return (int64_t)COLSUM_MAX_WGS * C * 4; }

extern "C" int dsf_col_sum(const float* x, int64_t M, int C, float* out, float* workspace, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && out && M > 0 && C > 0);
    hipStream_t st = (hipStream_t)stream;
    static const int small_on = [] { const char* e = getenv("DSF_COLSUM_SMALL"); return e ? atoi(e) : 1; }();     // tuning aid
    if (small_on && (C & 3) == 0 && M * C <= (1 << 20) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        hipLaunchKernelGGL(col_sum_small_kernel, dim3((C / 4 + 3) / 4), dim3(256), 0, st, x, (int)M, C, out);
        return dsf_launch_status();
    }
    if (workspace && (C & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0) {
        int c4n = 1;
        while (c4n < (C >> 2) && c4n < 256) c4n <<= 1;        // float4 column groups per pass (power of two <= 256)
        const int rlanes = 256 / c4n;
        int64_t rows = (M + COLSUM_MAX_WGS - 1) / COLSUM_MAX_WGS;
        const int64_t min_rows = (int64_t)rlanes * 8;          // at least one unrolled trip per lane
        if (rows < min_rows) rows = min_rows;
        const int wgs = (int)((M + rows - 1) / rows);
        hipLaunchKernelGGL(col_sum_kernel, dim3(wgs), dim3(256), 0, st, x, M, C, c4n, (int)rows, workspace);
        hipLaunchKernelGGL(col_sum_combine_kernel, dim3((C + 63) / 64), dim3(256), 0, st, workspace, wgs, C, out);
        return dsf_launch_status();
    }
    int cpad = 1;
    while (cpad < C && cpad < 256) cpad <<= 1;         // columns handled per pass (power of two <= 256)
    int64_t rows = (M + 255) / 256;
    if (rows < 64) rows = 64;
    int wgs = (int)((M + rows - 1) / rows);            // <= 256 = COLSUM_MAX_WGS partial rows
    if (workspace) {
        // partial rows + the fixed-order fold, as above (float atomics from the workgroups gave sums that depended on their
        // order -- the heads' 63- and 21-channel bias gradients differed between a one-stream and a forked-stream run of config 3)
        hipLaunchKernelGGL(col_sum_scalar_kernel, dim3(wgs), dim3(256), 0, st, x, M, C, cpad, (int)rows, out, workspace);
        hipLaunchKernelGGL(col_sum_combine_kernel, dim3((C + 63) / 64), dim3(256), 0, st, workspace, wgs, C, out);
        return dsf_launch_status();
    }
    if (dsf_deterministic()) { rows = M; wgs = 1; }    // no workspace: one workgroup walks every row in deterministic mode
    if (dsf_zero_async(out, sizeof(float) * C, st) != hipSuccess) return DSF_ERR_LAUNCH;
    hipLaunchKernelGGL(col_sum_scalar_kernel, dim3(wgs), dim3(256), 0, st, x, M, C, cpad, (int)rows, out, (float*)nullptr);
    return dsf_launch_status();
}
