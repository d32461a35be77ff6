// Direct convolution for 1-channel inputs (the 5x5 ResNet stem, the 7x7 stride-2 hourglass stem, the generator's first
// layer; reference model/backbone.py:196-199, model/hourglass.py:178, render_model/transfer.py:409).  With Ci = 1 the
// implicit GEMM has K = KH*KW <= 49: its tiles are mostly padding (the generic kernels ran at 10-20 TFLOP/s) and the layer
// is bound by the 64-channel tensor on the other side (134 MB at B = 32, 128 x 128).  Here lane = output channel: a wave
// walks row segments of 8 output pixels, reads each needed input row segment once (one coalesced load, then v_readlane
// broadcasts into scalar registers) and does KH*KW*8 FMAs per lane against weights it keeps in registers; the 64-channel
// tensor is read / written exactly once, 256 bytes per pixel per instruction.
#include "common.h"

namespace {

constexpr int PX = 8;                                   // output pixels of one row per step

struct C1P { int B, Hi, Wi, Ho, Wo, Co, pad; };

// input columns ox0*S - pad + j, j < (PX-1)*S + K, of input row iy (zeros outside the image) -> lane j
template <int K, int S>
__device__ __forceinline__ float c1_row(const float* __restrict__ X, const C1P& p, int b, int iy, int ix0, int lane) {
    constexpr int NIN = (PX - 1) * S + K;
    const int ix = ix0 + lane;
    const bool ok = lane < NIN && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
    return ok ? X[((int64_t)b * p.Hi + iy) * p.Wi + ix] : 0.f;
}

// STATS: the per-channel sum and sum of squares of the output (the batch statistics of the BatchNorm behind the stem) are taken
// from the accumulators -- lane = channel: two more registers -- and ADDED as doubles into row (workgroup mod acc_rows) of a zeroed
// [acc_rows][2][Co] block, the layout norm.hip's apply kernels fold in their prologue (dsf_bn_forward_acc with acc_filled): the
// BatchNorm's own statistics pass over the 134 MB output is not run.
template <int K, int S, bool STATS>
__device__ __forceinline__ void c1_fwd_body(const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ bias,
                                            float* __restrict__ Y, const C1P& p, int segs_per_row, int64_t n_segs, double* __restrict__ stat,
                                            int acc_rows) {
    constexpr int NIN = (PX - 1) * S + K;
    static_assert(NIN <= 64, "one wave-wide load per input row");
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const bool c_ok = lane < p.Co;
    float w[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) w[i] = c_ok ? W[i * p.Co + lane] : 0.f;
    const float bv = (bias && c_ok) ? bias[lane] : 0.f;
    float s0 = 0.f, s1 = 0.f;
    for (int64_t seg = wave; seg < n_segs; seg += n_waves) {
        const int sx = (int)(seg % segs_per_row);
        const int64_t row = seg / segs_per_row;
        const int oy = (int)(row % p.Ho), b = (int)(row / p.Ho);
        const int ox0 = sx * PX, ix0 = ox0 * S - p.pad;
        float acc[PX];
#pragma unroll
        for (int q = 0; q < PX; ++q) acc[q] = bv;
        float v[K];
#pragma unroll
        for (int kh = 0; kh < K; ++kh) v[kh] = c1_row<K, S>(X, p, b, oy * S + kh - p.pad, ix0, lane);
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {                 // one input row at a time: its NIN values live in scalar registers
            float xs[NIN];
#pragma unroll
            for (int j = 0; j < NIN; ++j) xs[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[kh]), j));     // (the builtin is typed int)
#pragma unroll
            for (int kw = 0; kw < K; ++kw)
#pragma unroll
                for (int q = 0; q < PX; ++q) acc[q] = fmaf(xs[q * S + kw], w[kh * K + kw], acc[q]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c_ok) {
            float* dst = Y + (((int64_t)b * p.Ho + oy) * p.Wo + ox0) * p.Co + lane;
#pragma unroll
            for (int q = 0; q < PX; ++q, dst += p.Co)
                if (ox0 + q < p.Wo) {
                    *dst = acc[q];
                    if (STATS) { s0 += acc[q]; s1 = fmaf(acc[q], acc[q], s1); }
                }
        }
    }
    if (STATS) {
        __shared__ float red[2][4][64];
        const int wv = threadIdx.x >> 6;
        red[0][wv][lane] = s0; red[1][wv][lane] = s1;
        __syncthreads();
        if (wv == 0 && c_ok) {
            double* row = stat + (int64_t)(blockIdx.x % acc_rows) * 2 * p.Co;
            __hip_atomic_fetch_add(row + lane, (double)((red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane])),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(row + p.Co + lane, (double)((red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane])),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int K, int S>
__global__ __launch_bounds__(256) void conv_c1_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ Y, C1P p,
                                                          int segs_per_row, int64_t n_segs) {
    c1_fwd_body<K, S, false>(X, W, bias, Y, p, segs_per_row, n_segs, nullptr, 1);
}

template <int K, int S>
__global__ __launch_bounds__(256) void conv_c1_fwd_stats_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                                const float* __restrict__ bias, float* __restrict__ Y, C1P p,
                                                                int segs_per_row, int64_t n_segs, double* __restrict__ stat, int acc_rows) {
    c1_fwd_body<K, S, true>(X, W, bias, Y, p, segs_per_row, n_segs, stat, acc_rows);
}

// dW[kh][kw][co] = sum over pixels of x[iy][ix] * gy[pixel][co]: per-lane accumulators, workgroup partials, then a combine
template <int K, int S>
__global__ __launch_bounds__(256) void conv_c1_wrw_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                          float* __restrict__ part, C1P p, int segs_per_row, int64_t n_segs) {
    constexpr int NIN = (PX - 1) * S + K;
    __shared__ float red[4][K * K][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, n_waves = (int64_t)gridDim.x * 4;
    const bool c_ok = lane < p.Co;
    float acc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
    for (int64_t seg = wave; seg < n_segs; seg += n_waves) {
        const int sx = (int)(seg % segs_per_row);
        const int64_t row = seg / segs_per_row;
        const int oy = (int)(row % p.Ho), b = (int)(row / p.Ho);
        const int ox0 = sx * PX, ix0 = ox0 * S - p.pad;
        float g[PX];
        const float* src = dY + (((int64_t)b * p.Ho + oy) * p.Wo + ox0) * p.Co + lane;
#pragma unroll
        for (int q = 0; q < PX; ++q, src += p.Co) g[q] = (c_ok && ox0 + q < p.Wo) ? *src : 0.f;
        float v[K];
#pragma unroll
        for (int kh = 0; kh < K; ++kh) v[kh] = c1_row<K, S>(X, p, b, oy * S + kh - p.pad, ix0, lane);
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            float xs[NIN];
#pragma unroll
            for (int j = 0; j < NIN; ++j) xs[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[kh]), j));     // (the builtin is typed int)
#pragma unroll
            for (int kw = 0; kw < K; ++kw)
#pragma unroll
                for (int q = 0; q < PX; ++q) acc[kh * K + kw] = fmaf(xs[q * S + kw], g[q], acc[kh * K + kw]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < K * K; ++i) red[wv][i][lane] = acc[i];
    __syncthreads();
    for (int i = threadIdx.x; i < K * K * 64; i += 256) {
        const int t = i >> 6, c = i & 63;
        part[(int64_t)blockIdx.x * (K * K * 64) + i] = (red[0][t][c] + red[1][t][c]) + (red[2][t][c] + red[3][t][c]);
    }
}

// ---- backward of the stem: BatchNorm(+ReLU)(+MaxPool2d) backward apply AND the convolution's dW in one launch ----------------------
// Behind a 1-channel convolution nobody needs the gradient of the convolution's OUTPUT except the dW sum above: the BatchNorm
// backward's apply pass writes 134 MB (B = 32) that conv_c1_wrw_kernel reads back once.  Here conv_c1_wrw_kernel's wave -- lane =
// channel, row segments of 8 pixels -- computes that gradient in registers instead of loading it: per pixel the saved convolution
// output y, the incoming gradient (POOL 0: the BatchNorm output's gradient, one coalesced load; POOL 1 / 2: gathered from the POOLED
// gradient through the argmax bytes of MaxPool2d(3, 2, 1) / (2, 2, 0), norm.hip's bn_pool_gather terms in their order), the ReLU
// mask recomputed from y, and norm.hip's bn_bwd_apply_kernel expression, operation for operation: dW is bit for bit what the two
// launches produce.  The channel sums come from the accumulation rows a sums-only pass left (folded per wave: 16 doubles per lane);
// workgroup 0 stores dgamma / dbeta.  A segment's pooled neighbourhood is 2 rows x 5 columns: 10 (byte, float) loads per lane, all
// issued before the first use; the window position a pixel has in a pooled column is a compile-time function of (pixel, column).
struct C1Bn { const float* gamma; const float* beta; const float* mean; const float* invstd; const double* rows; int n_rows; int relu;
              float* dgamma; float* dbeta; int accum; int PHo, PWo; };

template <int K, int S, int POOL>
__global__ __launch_bounds__(256) void conv_c1_wrw_bn_kernel(const float* __restrict__ X, const float* __restrict__ Yc, const float* __restrict__ G,
                                                             const uint8_t* __restrict__ arg, float* __restrict__ part, C1P p, C1Bn n,
                                                             int segs_per_row, int64_t n_segs) {
    constexpr int NIN = (PX - 1) * S + K;
    constexpr int PK = (POOL == 1) ? 3 : 2, PP = (POOL == 1) ? 1 : 0;                         // pooling window, padding
    constexpr int NPR = (POOL == 1) ? 2 : 1, NPC = (POOL == 1) ? PX / 2 + 1 : PX / 2;         // pooled rows / columns a segment's pixels sit in
    __shared__ float red[4][K * K + 1][64];                     // (slot K*K: the bias gradient, the per-channel sum of the gradient)
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, n_waves = (int64_t)gridDim.x * 4;
    const bool c_ok = lane < p.Co;
    const int ch = c_ok ? lane : 0;
    float sb = 0.f;
    double f0 = 0.0, f1 = 0.0;
    for (int r = 0; r < n.n_rows; ++r) { f0 += n.rows[(int64_t)r * 2 * p.Co + ch]; f1 += n.rows[(int64_t)r * 2 * p.Co + p.Co + ch]; }
    if (blockIdx.x == 0 && wv == 0 && c_ok) {
        if (n.dbeta) n.dbeta[lane] = (n.accum ? n.dbeta[lane] : 0.f) + (float)f0;
        if (n.dgamma) n.dgamma[lane] = (n.accum ? n.dgamma[lane] : 0.f) + (float)f1;
    }
    const float invM = 1.0f / (float)((int64_t)p.B * p.Ho * p.Wo);
    const float mu = n.mean[ch], is = n.invstd[ch], m0 = (float)f0 * invM, m1 = (float)f1 * invM;
    const float sc = (n.gamma ? n.gamma[ch] : 1.f) * is, sh = n.beta ? n.beta[ch] : 0.f;
    float acc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
    for (int64_t seg = wave; seg < n_segs; seg += n_waves) {
        const int sx = (int)(seg % segs_per_row);
        const int64_t row = seg / segs_per_row;
        const int oy = (int)(row % p.Ho), b = (int)(row / p.Ho);
        const int ox0 = sx * PX, ix0 = ox0 * S - p.pad;
        float yv[PX], g[PX];
        const int64_t o0 = (((int64_t)b * p.Ho + oy) * p.Wo + ox0) * p.Co + lane;
#pragma unroll
        for (int q = 0; q < PX; ++q) yv[q] = (c_ok && ox0 + q < p.Wo) ? Yc[o0 + (int64_t)q * p.Co] : 0.f;
        if (POOL == 0) {
#pragma unroll
            for (int q = 0; q < PX; ++q) g[q] = (c_ok && ox0 + q < p.Wo) ? G[o0 + (int64_t)q * p.Co] : 0.f;
        } else {
            // pooled rows pr0 (, pr0 + 1) and columns pc0 .. pc0 + NPC - 1 (ox0 is a multiple of 8): pixel (oy, ox0 + q) sits at window
            // position (oy - (2 pr - PP), q - 2 j + PP) of pooled pixel (pr, pc0 + j) when that position lies inside the window
            const int pr0 = (oy + PP - PK + 2) >> 1, pc0 = ox0 >> 1;
            float gp[NPR][NPC]; int ap[NPR][NPC]; int kh[NPR];
#pragma unroll
            for (int i = 0; i < NPR; ++i) {
                const int pr = pr0 + i;
                const int k = oy - (2 * pr - PP);
                kh[i] = (k >= 0 && k < PK && pr >= 0 && pr < n.PHo) ? k : -1;
                const int prc = min(max(pr, 0), n.PHo - 1);
#pragma unroll
                for (int j = 0; j < NPC; ++j) {
                    const int64_t o = (((int64_t)b * n.PHo + prc) * n.PWo + min(pc0 + j, n.PWo - 1)) * p.Co + ch;
                    gp[i][j] = G[o]; ap[i][j] = arg[o];
                }
            }
#pragma unroll
            for (int q = 0; q < PX; ++q) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < NPR; ++i)                   // (rows, then columns: maxpool_bwd_kernel's order)
#pragma unroll
                    for (int j = 0; j < NPC; ++j) {
                        const int kw = q - 2 * j + PP;          // compile-time
                        if (kw < 0 || kw >= PK) continue;
                        const int me = (kh[i] >= 0 && pc0 + j < n.PWo) ? kh[i] * PK + kw : 255;
                        s += (ap[i][j] == me) ? gp[i][j] : 0.f;
                    }
                g[q] = (c_ok && ox0 + q < p.Wo) ? s : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < PX; ++q) {                          // bn_bwd_apply_kernel's arithmetic (relu mode 2)
            float ge = g[q];
            if (n.relu) ge = ((yv[q] - mu) * sc + sh > 0.f) ? ge : 0.f;
            const float xh = (yv[q] - mu) * is;
            const float d = sc * (ge - m0 - xh * m1);
            g[q] = (c_ok && ox0 + q < p.Wo) ? d : 0.f;
            sb += g[q];
        }
        float v[K];
#pragma unroll
        for (int kh2 = 0; kh2 < K; ++kh2) v[kh2] = c1_row<K, S>(X, p, b, oy * S + kh2 - p.pad, ix0, lane);
#pragma unroll
        for (int kh2 = 0; kh2 < K; ++kh2) {
            float xs[NIN];
#pragma unroll
            for (int j = 0; j < NIN; ++j) xs[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[kh2]), j));
#pragma unroll
            for (int kw = 0; kw < K; ++kw)
#pragma unroll
                for (int q = 0; q < PX; ++q) acc[kh2 * K + kw] = fmaf(xs[q * S + kw], g[q], acc[kh2 * K + kw]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < K * K; ++i) red[wv][i][lane] = acc[i];
    red[wv][K * K][lane] = sb;
    __syncthreads();
    for (int i = threadIdx.x; i < (K * K + 1) * 64; i += 256) {
        const int t = i >> 6, c = i & 63;
        part[(int64_t)blockIdx.x * ((K * K + 1) * 64) + i] = (red[0][t][c] + red[1][t][c]) + (red[2][t][c] + red[3][t][c]);
    }
}

// one workgroup per tap: lane = channel, the SIXTEEN waves take every sixteenth partial (round 6: four waves walked 512 partials each,
// four loads in flight -- 44 us at the very end of the backward pass); fixed order, double accumulation
__global__ __launch_bounds__(1024) void conv_c1_wrw_combine_kernel(const float* __restrict__ part, int n_part, int taps, int Co,
                                                                   int accumulate, float* __restrict__ dW) {
    __shared__ double red[16][64];
    const int t = blockIdx.x, c = threadIdx.x & 63, g = threadIdx.x >> 6;
    double s = 0.0;
#pragma unroll 8
    for (int b = g; b < n_part; b += 16) s += (double)part[((int64_t)b * taps + t) * 64 + c];
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && c < Co) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][c];
        float* dst = dW + t * Co + c;
        *dst = accumulate ? *dst + (float)v : (float)v;
    }
}

constexpr int C1_STATS_WGS = 4096;                    // the statistics variant: 128 double atomics per workgroup onto acc_rows x 2 Co addresses; B = 32 stem alone: 68.7 / 56.0 / 52.1 / 50.1 us at 512 / 1024 / 2048 / 4096 workgroups, 50.1 without the statistics
constexpr int C1_WRW_WGS = 2048;                      // 8 waves per SIMD: the kernel lives on loads in flight

inline bool c1_ok(int Co, int K, int stride) { return Co >= 1 && Co <= 64 && (K == 5 || K == 7) && (stride == 1 || stride == 2); }


// ------------------------------------------------------------------------------------------------
// conv_co1_fwd_kernel: ONE output channel (the generator's last layer: ReflectionPad2d(3) + Conv2d(64, 1, 7), reference
// render_model/transfer.py; 1 M output pixels x 3136 products at B = 64).  As an implicit GEMM it pads N from 1 to 64 columns
// (igemm_x6b_kernel<64, false, 256>: 1.79 ms, 3.7 TFLOP/s of useful work).  Here a lane owns output pixels: a workgroup takes a
// 64 x 8 tile (wave w: rows 2 w and 2 w + 1, lane = column), stages the (8 + K - 1) x (64 + K - 1) input patch of an 8-channel chunk
// in LDS ([pixel][8 + 4 floats]: the 16-lane groups of a ds_read_b128 hit 64 distinct banks), and every lane accumulates its two
// pixels' dot products with fp32 FMAs -- an input row is read once for the two output rows it feeds; the weights are wave-uniform
// (scalar loads).  HBM: the input once (+ halo), 4 bytes out per pixel.
// ------------------------------------------------------------------------------------------------
struct Co1P { int B, Hi, Wi, Ci, Ho, Wo, pad; };

template <int K>
__global__ __launch_bounds__(256) void conv_co1_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ Y, Co1P p,
                                                           int tiles_x, int tiles_y) {
    constexpr int TX = 64, TY = 8, CH = 8, PITCH = CH + 4;               // floats per staged pixel
    constexpr int PW = TX + K - 1, PH = TY + K - 1, NPIX = PW * PH;
    __shared__ __attribute__((aligned(16))) float s_in[NPIX * PITCH];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int tile = blockIdx.x;
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y; const int b = tile / tiles_y;
    const int x0 = tx * TX, y0 = ty * TY;                                // first output pixel of the tile
    float acc0 = 0.f, acc1 = 0.f;
    for (int c0 = 0; c0 < p.Ci; c0 += CH) {
        __syncthreads();                                                 // the previous chunk's readers are done
        for (int e = t; e < NPIX * 2; e += 256) {                        // (patch pixel, channel quad)
            const int pix = e >> 1, q = e & 1;
            const int py = pix / PW, px = pix - py * PW;
            const int iy = y0 + py - p.pad, ix = x0 + px - p.pad;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
                v = *reinterpret_cast<const float4*>(X + ((int64_t)(b * p.Hi + iy) * p.Wi + ix) * p.Ci + c0 + q * 4);
            *reinterpret_cast<float4*>(&s_in[pix * PITCH + q * 4]) = v;
        }
        __syncthreads();
        const float* wc = W + c0;                                        // W [K][K][Ci]: tap (ky, kx) of this chunk at wc[(ky K + kx) Ci ..]
#pragma unroll
        for (int ry = 0; ry <= K; ++ry) {                                // input row 2 wave + ry feeds output row 0 (tap ry) and 1 (tap ry - 1)
            const float* row = &s_in[((2 * wave + ry) * PW + lane) * PITCH];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float4 a = *reinterpret_cast<const float4*>(row + kx * PITCH);
                const float4 c = *reinterpret_cast<const float4*>(row + kx * PITCH + 4);
                const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
                if (ry < K) {
                    const float* w0 = wc + (ry * K + kx) * p.Ci;
#pragma unroll
                    for (int j = 0; j < CH; ++j) acc0 = fmaf(v[j], w0[j], acc0);
                }
                if (ry >= 1) {
                    const float* w1 = wc + ((ry - 1) * K + kx) * p.Ci;
#pragma unroll
                    for (int j = 0; j < CH; ++j) acc1 = fmaf(v[j], w1[j], acc1);
                }
            }
        }
    }
    const float bv = bias ? bias[0] : 0.f;
    const int ox = x0 + lane, oy = y0 + 2 * wave;
    if (ox < p.Wo) {
        if (oy < p.Ho) Y[(int64_t)(b * p.Ho + oy) * p.Wo + ox] = acc0 + bv;
        if (oy + 1 < p.Ho) Y[(int64_t)(b * p.Ho + oy + 1) * p.Wo + ox] = acc1 + bv;
    }
}

}  // namespace

extern "C" {

int dsf_conv_c1_supported(int Co, int KH, int KW, int stride) { return (KH == KW && c1_ok(Co, KH, stride)) ? 1 : 0; }

int64_t dsf_conv_c1_workspace_bytes(int KH, int KW) { return (int64_t)C1_WRW_WGS * (KH * KW + 1) * 64 * 4; }     // (+ 1: dsf_conv_c1_wrw_bn's bias slot)

int dsf_conv_c1_forward(const float* X, const float* W, const float* bias, float* Y, int B, int Hi, int Wi, int Ho, int Wo, int Co,
                        int K, int stride, int pad, dsf_stream_t stream) {
    DSF_CHECK_ARG(X && W && Y && B >= 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && pad >= 0);
    if (!c1_ok(Co, K, stride)) return DSF_ERR_UNSUPPORTED;
    if (B == 0) return DSF_OK;
    C1P p = {B, Hi, Wi, Ho, Wo, Co, pad};
    const int spr = (Wo + PX - 1) / PX;
    const int64_t n_segs = (int64_t)B * Ho * spr;
    int64_t wgs = (n_segs + 3) / 4;
    if (wgs > 4096) wgs = 4096;
#define DSF_LAUNCH_C1(Kv, Sv) hipLaunchKernelGGL((conv_c1_fwd_kernel<Kv, Sv>), dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, \
                                                 X, W, bias, Y, p, spr, n_segs)
    if (K == 5) { if (stride == 1) DSF_LAUNCH_C1(5, 1); else DSF_LAUNCH_C1(5, 2); }
    else { if (stride == 1) DSF_LAUNCH_C1(7, 1); else DSF_LAUNCH_C1(7, 2); }
#undef DSF_LAUNCH_C1
    return dsf_launch_status();
}

int dsf_conv_c1_forward_bn_acc(const float* X, const float* W, const float* bias, float* Y, int B, int Hi, int Wi, int Ho, int Wo, int Co, int K, int stride,
                               int pad, double* acc, int acc_rows, dsf_stream_t stream) {
    DSF_CHECK_ARG(X && W && Y && acc && acc_rows >= 1 && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && pad >= 0);
    if (!c1_ok(Co, K, stride) || dsf_deterministic()) return DSF_ERR_UNSUPPORTED;
    C1P p = {B, Hi, Wi, Ho, Wo, Co, pad};
    const int spr = (Wo + PX - 1) / PX;
    const int64_t n_segs = (int64_t)B * Ho * spr;
    int64_t wgs = (n_segs + 3) / 4;
    const char* cap_e = getenv("DSF_C1_STATS_WGS");                    // tuning aid, read per call
    const int64_t cap = (cap_e && atoi(cap_e) > 0) ? atoi(cap_e) : C1_STATS_WGS;
    if (wgs > cap) wgs = cap;
#define DSF_LAUNCH_C1(Kv, Sv) hipLaunchKernelGGL((conv_c1_fwd_stats_kernel<Kv, Sv>), dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, \
                                                 X, W, bias, Y, p, spr, n_segs, acc, acc_rows)
    if (K == 5) { if (stride == 1) DSF_LAUNCH_C1(5, 1); else DSF_LAUNCH_C1(5, 2); }
    else { if (stride == 1) DSF_LAUNCH_C1(7, 1); else DSF_LAUNCH_C1(7, 2); }
#undef DSF_LAUNCH_C1
    return dsf_launch_status();
}

int dsf_conv_c1_wrw_bn(const float* X, const float* Y, const float* grad, const uint8_t* argmax, const float* gamma, const float* beta,
                       const float* save_mean, const float* save_invstd, const double* acc, int acc_rows, int relu, int pool_k, int pool_stride,
                       int pool_pad, float* dW, float* grad_gamma, float* grad_beta, int accumulate_affine, float* workspace, int B, int Hi,
                       int Wi, int Ho, int Wo, int Co, int K, int stride, int pad, dsf_stream_t stream) {
    DSF_CHECK_ARG(X && Y && grad && save_mean && save_invstd && acc && acc_rows >= 1 && dW && workspace && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 &&
                  Wo > 0 && pad >= 0 && (pool_k == 0 || argmax));
    if (!c1_ok(Co, K, stride)) return DSF_ERR_UNSUPPORTED;
    const int pool = pool_k == 0 ? 0 : ((pool_k == 3 && pool_stride == 2 && pool_pad == 1) ? 1 : ((pool_k == 2 && pool_stride == 2 && pool_pad == 0) ? 2 : -1));
    if (pool < 0) return DSF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    C1P p = {B, Hi, Wi, Ho, Wo, Co, pad};
    C1Bn n = {gamma, beta, save_mean, save_invstd, acc, acc_rows, relu, grad_gamma, grad_beta, accumulate_affine,
              pool ? (Ho + 2 * pool_pad - pool_k) / pool_stride + 1 : 0, pool ? (Wo + 2 * pool_pad - pool_k) / pool_stride + 1 : 0};
    if (pool && (n.PHo <= 0 || n.PWo <= 0)) return DSF_ERR_UNSUPPORTED;
    const int spr = (Wo + PX - 1) / PX;
    const int64_t n_segs = (int64_t)B * Ho * spr;
    int wgs = (int)((n_segs + 3) / 4 < C1_WRW_WGS ? (n_segs + 3) / 4 : C1_WRW_WGS);
#define DSF_LAUNCH_C1B(Kv, Sv, Pv) hipLaunchKernelGGL((conv_c1_wrw_bn_kernel<Kv, Sv, Pv>), dim3(wgs), dim3(256), 0, st, X, Y, grad, argmax, workspace, p, n, spr, n_segs)
#define DSF_LAUNCH_C1B_P(Kv, Sv) do { if (pool == 0) DSF_LAUNCH_C1B(Kv, Sv, 0); else if (pool == 1) DSF_LAUNCH_C1B(Kv, Sv, 1); else DSF_LAUNCH_C1B(Kv, Sv, 2); } while (0)
    if (K == 5) { if (stride == 1) DSF_LAUNCH_C1B_P(5, 1); else DSF_LAUNCH_C1B_P(5, 2); }
    else { if (stride == 1) DSF_LAUNCH_C1B_P(7, 1); else DSF_LAUNCH_C1B_P(7, 2); }
#undef DSF_LAUNCH_C1B_P
#undef DSF_LAUNCH_C1B
    hipLaunchKernelGGL(conv_c1_wrw_combine_kernel, dim3(K * K + 1), dim3(1024), 0, st, workspace, wgs, K * K + 1, Co, 0, dW);
    return dsf_launch_status();
}

int dsf_conv_c1_wrw(const float* X, const float* dY, float* dW, float* workspace, int B, int Hi, int Wi, int Ho, int Wo, int Co,
                    int K, int stride, int pad, int accumulate, dsf_stream_t stream) {
    DSF_CHECK_ARG(X && dY && dW && workspace && B >= 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && pad >= 0);
    if (!c1_ok(Co, K, stride)) return DSF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        if (!accumulate && dsf_zero_async(dW, sizeof(float) * K * K * Co, st) != hipSuccess) return DSF_ERR_LAUNCH;
        return DSF_OK;
    }
    C1P p = {B, Hi, Wi, Ho, Wo, Co, pad};
    const int spr = (Wo + PX - 1) / PX;
    const int64_t n_segs = (int64_t)B * Ho * spr;
    int wgs = (int)((n_segs + 3) / 4 < C1_WRW_WGS ? (n_segs + 3) / 4 : C1_WRW_WGS);
#define DSF_LAUNCH_C1W(Kv, Sv) hipLaunchKernelGGL((conv_c1_wrw_kernel<Kv, Sv>), dim3(wgs), dim3(256), 0, st, X, dY, workspace, p, spr, n_segs)
    if (K == 5) { if (stride == 1) DSF_LAUNCH_C1W(5, 1); else DSF_LAUNCH_C1W(5, 2); }
    else { if (stride == 1) DSF_LAUNCH_C1W(7, 1); else DSF_LAUNCH_C1W(7, 2); }
#undef DSF_LAUNCH_C1W
    hipLaunchKernelGGL(conv_c1_wrw_combine_kernel, dim3(K * K), dim3(1024), 0, st, workspace, wgs, K * K, Co, accumulate, dW);
    return dsf_launch_status();
}

// One output channel, square K in {3, 5, 7}, stride 1, Ci % 8 == 0: X (B,Hi,Wi,Ci) NHWC, W [K][K][Ci] (= the kernel layout
// [K][K][Ci][1]), Y (B,Ho,Wo); other shapes DSF_ERR_UNSUPPORTED.
int dsf_conv_co1_forward(const float* X, const float* W, const float* bias, float* Y, int B, int Hi, int Wi, int Ci, int Ho, int Wo,
                         int K, int pad, dsf_stream_t stream) {
    DSF_CHECK_ARG(X && W && Y && B >= 0 && Hi > 0 && Wi > 0 && Ci > 0 && Ho > 0 && Wo > 0 && pad >= 0);
    if (!(K == 3 || K == 5 || K == 7) || (Ci & 7) || Ho != Hi + 2 * pad - K + 1 || Wo != Wi + 2 * pad - K + 1) return DSF_ERR_UNSUPPORTED;
    if (B == 0) return DSF_OK;
    Co1P p = {B, Hi, Wi, Ci, Ho, Wo, pad};
    const int tiles_x = (Wo + 63) / 64, tiles_y = (Ho + 7) / 8;
    const dim3 grid((unsigned)(B * tiles_x * tiles_y));
    if (K == 3) hipLaunchKernelGGL(conv_co1_fwd_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, X, W, bias, Y, p, tiles_x, tiles_y);
    else if (K == 5) hipLaunchKernelGGL(conv_co1_fwd_kernel<5>, grid, dim3(256), 0, (hipStream_t)stream, X, W, bias, Y, p, tiles_x, tiles_y);
    else hipLaunchKernelGGL(conv_co1_fwd_kernel<7>, grid, dim3(256), 0, (hipStream_t)stream, X, W, bias, Y, p, tiles_x, tiles_y);
    return dsf_launch_status();
}

}  // extern "C"
