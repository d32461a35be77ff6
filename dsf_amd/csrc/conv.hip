// K11: fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32
// in / f32 accumulate, 157 TFLOP/s peak) for the backbone's dense layers.
//
// Why hand-written: ROCm 7.2's MIOpen ships no gfx950 find-db / perf-db / kernel-db.  In find mode
// (the reference's cudnn.benchmark=True) every solver is JIT-compiled for every layer (hours on a
// fresh box); in immediate mode it falls back to `naive_conv_*` kernels (45 ms per launch, 3.3 s per
// step measured: profiles/r01_step_miopen_immediate_stats.csv).  These two kernels replace all
// Conv2d / ConvTranspose2d forward, backward-data and backward-weight passes of ResNet-18/50,
// the hourglass and the CycleGAN generator.
//
// Layout: NHWC activations (torch channels_last), weights as a [KH*KW*Ci][Co] row-major GEMM
// operand.  GEMM view: M = B*Ho*Wo output pixels, N = Co, K = KH*KW*Ci; the A operand is gathered
// on the fly (no im2col buffer in HBM).  Block tile 128(M) x BN(N: 128 or 64) x 32(K), 256 threads =
// 4 waves, each wave a 64x64 (or 32x64) sub-tile of 32x32 MFMA tiles.  Operands are staged k-major in
// LDS so that the MFMA fragment read (lane l -> row l&31, k = l>>5) is a conflict-free ds_read_b32;
// the next K-chunk's global loads are issued before the MFMA block and written to LDS after it.
// The fp32 MFMA takes 64 cycles per 32x32x2, so per 32-deep chunk a wave spends 4096 MFMA cycles
// against 16 dwords of global traffic per lane: the kernel is MFMA-bound by construction.
//
// `dil` generalises the gather to transposed convolutions (virtual input = X upsampled by `dil`
// with zeros): forward conv (stride s, dil 1), backward-data of a stride-1 conv (flipped weights),
// backward-data of a stride-2 conv and ConvTranspose2d forward (dil 2) all run the same kernel.
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvP {
    int B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, dil, pad_h, pad_w;
};

constexpr int BM = 128, BK = 32;

// ------------------------------------------------------------------------------------------------
// forward-type kernel: Y[m][n] = sum_k A[m][k] W[k][n] (+ bias[n])
// ------------------------------------------------------------------------------------------------
// FLAT: K is walked as one flat (tap, channel) index in 32-wide chunks (small Ci, e.g. the 1-channel
// stem) instead of per-tap channel chunks.  k_splits > 1: the K range is split across workgroups and
// the partial tiles are combined with float atomics (small-M layers: 512ch @ 8x8 has only 64 tiles).
// perm: with dil > 1 output pixels are ordered parity-class-major, so all rows of a tile share
// (oy % dil, ox % dil) and the taps that only ever hit inserted zeros are skipped tile-wide.
template <int BN, bool FLAT>
__global__ __launch_bounds__(256) void igemm_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ Y, ConvP p,
                                                        int m_tiles, int n_tiles, int k_splits, int perm) {
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int WM = (BN == 128) ? 64 : 32;       // wave tile rows
    constexpr int TM = WM / 32, TN = 2;
    constexpr int B4 = BN / 4;                       // float4 per weight row of the tile
    constexpr int BROWS = 256 / B4;                  // weight rows loaded per pass
    constexpr int BPASS = BK / BROWS;
    __shared__ float As[BK * LDA];
    __shared__ float Bs[BK * LDB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (BN == 128) ? (wave >> 1) : wave, wn = (BN == 128) ? (wave & 1) : 0;
    int bid = blockIdx.x;
    const int m_tile = bid % m_tiles; bid /= m_tiles;
    const int n_tile = bid % n_tiles; const int ks = bid / n_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int M = p.B * p.Ho * p.Wo;
    const bool vec_a = (p.Ci & 3) == 0, vec_b = (p.Co & 3) == 0;
    const int Hq = p.Ho / p.dil, Wq = p.Wo / p.dil, Mc = p.B * Hq * Wq;      // perm only
    auto decode = [&](int m, int& b, int& oy, int& ox) {
        if (perm) {
            const int cls = m / Mc, r = m % Mc;
            const int qx = r % Wq, q = r / Wq;
            ox = qx * p.dil + cls % p.dil; oy = (q % Hq) * p.dil + cls / p.dil; b = q / Hq;
        } else {
            ox = m % p.Wo; const int q = m / p.Wo; oy = q % p.Ho; b = q / p.Ho;
        }
    };
    // tile-uniform parity class (-1: mixed / not applicable)
    int tile_py = -1, tile_px = -1;
    if (perm) {
        const int c0 = m0 / Mc, c1 = min(m0 + BM - 1, M - 1) / Mc;
        if (c0 == c1) { tile_py = c0 / p.dil; tile_px = c0 % p.dil; }
    }

    // A-load coordinates: 4 rows per thread, one float4 (4 consecutive k) each
    const int a_k4 = (t & 7) * 4;
    int a_b[4], a_oy[4], a_ox[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (t >> 3) + 32 * i;
        a_ok[i] = m < M;
        decode(a_ok[i] ? m : 0, a_b[i], a_oy[i], a_ox[i]);
    }
    const int b_n4 = (t % B4) * 4, b_row = t / B4;
    const int vH = (p.Hi - 1) * p.dil + 1, vW = (p.Wi - 1) * p.dil + 1;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[4], rb[BPASS];
    const int Ktot = p.KH * p.KW * p.Ci;
    const int chunks_per_tap = (p.Ci + BK - 1) / BK;
    const int n_chunks = FLAT ? (Ktot + BK - 1) / BK : p.KH * p.KW * chunks_per_tap;
    const int per_split = (n_chunks + k_splits - 1) / k_splits;
    const int chunk_lo = ks * per_split, chunk_hi = min(n_chunks, chunk_lo + per_split);

    auto tap_live = [&](int chunk) -> bool {          // false: every row of this tile reads inserted zeros
        if (FLAT || tile_py < 0) return true;
        const int tap = chunk / chunks_per_tap;
        const int kh = tap / p.KW, kw = tap % p.KW;
        return ((tile_py + kh - p.pad_h) % p.dil == 0) && ((tile_px + kw - p.pad_w) % p.dil == 0);
    };
    auto next_live = [&](int chunk) { while (chunk < chunk_hi && !tap_live(chunk)) ++chunk; return chunk; };

    auto load_chunk = [&](int chunk) {
        if (FLAT) {
            const int kbase = chunk * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = kbase + a_k4 + j;
                    if (k < Ktot && a_ok[i]) {
                        const int tap = k / p.Ci, c = k % p.Ci;
                        const int vy = a_oy[i] * p.stride + tap / p.KW - p.pad_h, vx = a_ox[i] * p.stride + tap % p.KW - p.pad_w;
                        if (vy >= 0 && vy < p.Hi && vx >= 0 && vx < p.Wi)
                            e[j] = X[(((int64_t)a_b[i] * p.Hi + vy) * p.Wi + vx) * p.Ci + c];
                    }
                }
                ra[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int k = kbase + b_row + BROWS * i, n = n0 + b_n4;
                if (k < Ktot && n < p.Co) {
                    const float* src = W + (int64_t)k * p.Co + n;
                    if (vec_b) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        v.x = src[0];
                        if (n + 1 < p.Co) v.y = src[1];
                        if (n + 2 < p.Co) v.z = src[2];
                        if (n + 3 < p.Co) v.w = src[3];
                    }
                }
                rb[i] = v;
            }
            return;
        }
        const int tap = chunk / chunks_per_tap, c0 = (chunk % chunks_per_tap) * BK;
        const int kh = tap / p.KW, kw = tap % p.KW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int vy = a_oy[i] * p.stride + kh - p.pad_h, vx = a_ox[i] * p.stride + kw - p.pad_w;
            bool ok = a_ok[i] && vy >= 0 && vy < vH && vx >= 0 && vx < vW;
            int iy = vy, ix = vx;
            if (p.dil > 1) {
                ok = ok && (vy % p.dil == 0) && (vx % p.dil == 0);
                iy = vy / p.dil; ix = vx / p.dil;
            }
            const int c = c0 + a_k4;
            if (ok && c < p.Ci) {
                const float* src = X + (((int64_t)a_b[i] * p.Hi + iy) * p.Wi + ix) * p.Ci + c;
                if (vec_a) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (c + 1 < p.Ci) v.y = src[1];
                    if (c + 2 < p.Ci) v.z = src[2];
                    if (c + 3 < p.Ci) v.w = src[3];
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int kk = b_row + BROWS * i;
            const int c = c0 + kk, n = n0 + b_n4;
            if (c < p.Ci && n < p.Co) {
                const float* src = W + ((int64_t)tap * p.Ci + c) * p.Co + n;
                if (vec_b) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (n + 1 < p.Co) v.y = src[1];
                    if (n + 2 < p.Co) v.z = src[2];
                    if (n + 3 < p.Co) v.w = src[3];
                }
            }
            rb[i] = v;
        }
    };

    int chunk = next_live(chunk_lo);
    if (chunk < chunk_hi) load_chunk(chunk);
    while (chunk < chunk_hi) {
        const int nxt = next_live(chunk + 1);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (t >> 3) + 32 * i;
            As[(a_k4 + 0) * LDA + r] = ra[i].x; As[(a_k4 + 1) * LDA + r] = ra[i].y;
            As[(a_k4 + 2) * LDA + r] = ra[i].z; As[(a_k4 + 3) * LDA + r] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i)
            *reinterpret_cast<float4*>(&Bs[(b_row + BROWS * i) * LDB + b_n4]) = rb[i];
        __syncthreads();
        if (nxt < chunk_hi) load_chunk(nxt);                   // in flight during the MFMA block
#pragma unroll 4
        for (int kk = 0; kk < BK; kk += 2) {
            const int k = kk + (lane >> 5);
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[k * LDA + wm * WM + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[k * LDB + wn * 64 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        chunk = nxt;
    }

    // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= M) continue;
            int64_t row = m;
            if (perm) { int b, oy, ox; decode(m, b, oy, ox); row = ((int64_t)b * p.Ho + oy) * p.Wo + ox; }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * 64 + j * 32 + (lane & 31);
                if (n >= p.Co) continue;
                const float bv = (bias && ks == 0) ? bias[n] : 0.f;
                if (k_splits > 1) atomicAdd(Y + row * p.Co + n, acc[i][j][r] + bv);
                else Y[row * p.Co + n] = acc[i][j][r] + bv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fast path of the forward-type kernel: dil == 1, Ci % 4 == 0, Co % 4 == 0, operands < 4 GiB.
// Same tiling and arithmetic as igemm_fwd_kernel, software-pipelined for the 64-cycle fp32 MFMA:
//  * gather through buffer_load_dwordx4 with a raw buffer descriptor: padding taps / ragged rows get the
//    offset 0xFFFFFFFF, which the hardware range check turns into zeros -- no select on the loaded
//    data, so nothing consumes the loads until the LDS store after the MFMA block (the compiler had put
//    `s_waitcnt vmcnt(0)` + 32 v_cndmask in front of the MFMAs of every chunk);
//  * two LDS stages, one barrier per 32-deep chunk: loads of chunk c+1 are in flight during the 64
//    MFMAs of chunk c and are stored to the other stage afterwards;
//  * MFMA fragments are double-buffered in registers (reads of k-step s+1 issued before the MFMAs of s);
//  * LDS tiles are k-major [32][128] with the column XOR-swizzled by (k & 28): the transposing
//    ds_write_b32 of the A tile (lanes = 8 k-quads x 4 rows) and the 32-wide fragment ds_read_b32 are both
//    bank-conflict-free without padding, so two stages are exactly 64 KiB and two workgroups share a CU;
//  * workgroup -> tile map is XCD-contiguous (the 8 XCDs take blockIdx round-robin; each XCD walks its
//    own range of tiles with the n tiles of one pixel tile adjacent, so they share the A rows in its L2).
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t dsf_buffer(const void* ptr, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 dsf_buffer_load4(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
    const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
    const f32x4 f = __builtin_bit_cast(f32x4, raw);
    return make_float4(f[0], f[1], f[2], f[3]);
}
constexpr uint32_t OOB = 0xFFFFFFFFu;

// blockIdx -> tile number such that every XCD (blockIdx % 8) owns one contiguous range of tiles
__device__ __forceinline__ int xcd_contiguous(int bid, int total) {
    const int q = total >> 3, rem = total & 7, xcd = bid & 7, local = bid >> 3;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
}

// LDS column swizzle of k-row k: spreads the k-quads a 32-lane store group touches over the 32 banks
template <int BKT>
__device__ __forceinline__ int lds_swz(int k) { return BKT == 32 ? (k & 28) : ((k & 12) << 1); }

struct NoPiece { __device__ __forceinline__ void operator()(int) const {} };

// One BKT-deep chunk of the block GEMM from one LDS stage: BKT/2 k-steps x (TM x TN) MFMAs per wave.
// `piece(s)` is issued right after the MFMAs of k-step s: the loader hands its global loads (address arithmetic +
// buffer_load) over one per k-step, so that VALU work runs in the shadow of the 64-cycle MFMAs instead of in front of
// them (one wave's loader was ~1000 cycles per chunk against 2048-4096 cycles of MFMA).
template <int BN, int TM, int TN, int BKT, bool SWZ = true, int BMT = BM, typename Piece = NoPiece>
__device__ __forceinline__ void mma_chunk(const float* __restrict__ Asb, const float* __restrict__ Bsb, int arow, int bcol,
                                          int lane, f32x16 (&acc)[TM][TN], Piece piece = Piece()) {
    const int h = lane >> 5, l31 = lane & 31;
    float af[2][TM], bf[2][TN];
    auto frag = [&](int kk, float* a, float* b) {
        const int k = kk + h, sw = SWZ ? (l31 ^ lds_swz<BKT>(kk)) : l31;       // kk is even: swz(kk + h) == swz(kk)
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = Asb[k * BMT + arow + i * 32 + sw];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bsb[k * BN + bcol + j * 32 + sw];
    };
    frag(0, af[0], bf[0]);
#pragma unroll
    for (int s = 0; s < BKT / 2; ++s) {
        if (s + 1 < BKT / 2) frag(2 * (s + 1), af[(s + 1) & 1], bf[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);           // keep the next step's LDS reads ahead of this step's MFMAs
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][i], bf[s & 1][j], acc[i][j], 0, 0, 0);
        piece(s);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// BMT: pixel rows per tile.  128, or 64 (BN = 128 only; each wave a 32x64 sub-tile) for small maps: twice the tiles, so
// 8x8..32x32 layers need half the split-K partials (or none) -- their float-atomic epilogues cost more than the MFMAs saved.
template <int BN, bool WT, int BKT, int BMT>
__global__ __launch_bounds__(256) void igemm_fwd_fast_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ Y,
                                                             ConvP p, int m_tiles, int n_tiles, int k_splits,
                                                             uint32_t x_bytes, uint32_t w_bytes) {
    constexpr int WM = (BN == 128) ? BMT / 2 : BMT / 4;
    constexpr int TM = WM / 32, TN = 2;
    constexpr int KQ = BKT / 4;                      // k-quads per tile row
    constexpr int AROWS = 256 / KQ, APASS = BMT / AROWS;         // transposing loader: thread -> (row, k-quad)
    constexpr int B4 = BN / 4, BROWS = 256 / B4, BPASS = BKT / BROWS;
    constexpr int WTPASS = BN / AROWS;               // WT: thread -> (column, k-quad)
    static_assert(TM >= 1 && APASS >= 1 && BPASS >= 1 && WTPASS >= 1, "tile / thread mapping");
    __shared__ float As[2][BKT * BMT];
    __shared__ float Bs[2][BKT * BN];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (BN == 128) ? (wave >> 1) : wave, wn = (BN == 128) ? (wave & 1) : 0;
    int tile = xcd_contiguous(blockIdx.x, m_tiles * n_tiles * k_splits);
    const int n_tile = tile % n_tiles; tile /= n_tiles;
    const int m_tile = tile % m_tiles; const int ks = tile / m_tiles;
    const int m0 = m_tile * BMT, n0 = n_tile * BN;
    const int M = p.B * p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t xbuf = dsf_buffer(X, x_bytes), wbuf = dsf_buffer(W, w_bytes);

    const int a_k4 = (t % KQ) * 4, a_r = t / KQ;
    int a_base[APASS], a_iy[APASS], a_ix[APASS];
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int m = m0 + a_r + AROWS * i;
        const bool ok = m < M;
        const int mm = ok ? m : 0;
        const int ox = mm % p.Wo, q = mm / p.Wo, oy = q % p.Ho, b = q / p.Ho;
        a_iy[i] = ok ? oy * p.stride - p.pad_h : -0x40000000;      // invalid rows fail every bounds test
        a_ix[i] = ox * p.stride - p.pad_w;
        a_base[i] = ((b * p.Hi + oy * p.stride - p.pad_h) * p.Wi + a_ix[i]) * p.Ci + a_k4;
    }
    const int b_n4 = (t % B4) * 4, b_row = t / B4;
    const bool b_nok = n0 + b_n4 < p.Co;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int chunks_per_tap = (p.Ci + BKT - 1) / BKT;
    const int n_chunks = p.KH * p.KW * chunks_per_tap;
    const int per_split = (n_chunks + k_splits - 1) / k_splits;
    const int chunk_lo = ks * per_split, chunk_hi = min(n_chunks, chunk_lo + per_split);

    // two register sets for the global loads: chunk c + 2 is requested while chunk c + 1 still waits in the other set
    // for its LDS stage, so a load has two chunk times (1.7-3.4 us of MFMA work) to come back, not one
    float4 ra[2][APASS], rb[2][WT ? WTPASS : BPASS];
    // wave-uniform walk state of the chunk being loaded
    int l_tap = chunk_lo / chunks_per_tap, l_c0 = (chunk_lo % chunks_per_tap) * BKT;
    int l_kh = l_tap / p.KW, l_kw = l_tap % p.KW;
    constexpr int NB = WT ? WTPASS : BPASS, NPIECE = APASS + NB;
    static_assert(NPIECE <= BKT / 2, "one loader piece per k-step");
    // piece i of a chunk's loads into register set SET: i < APASS an A row group, then the B groups; the last piece
    // advances the walk.  `live` = the chunk exists (loads past the end are still issued, with the out-of-range offset,
    // so the number of loads in flight -- what s_waitcnt vmcnt counts -- does not depend on the trip count)
    auto load_piece = [&](auto SET, int i, bool live) {
        constexpr int S = decltype(SET)::value;
        const bool c_ok = live && l_c0 + a_k4 < p.Ci;
        if (i < APASS) {
            const int tap_off = (l_kh * p.Wi + l_kw) * p.Ci + l_c0;
            const bool ok = c_ok && (unsigned)(a_iy[i] + l_kh) < (unsigned)p.Hi && (unsigned)(a_ix[i] + l_kw) < (unsigned)p.Wi;
            ra[S][i] = dsf_buffer_load4(xbuf, ok ? (uint32_t)(a_base[i] + tap_off) * 4u : OOB);
        } else if (WT) {
            const int j = i - APASS;
            const int tapf = (p.KH - 1 - l_kh) * p.KW + (p.KW - 1 - l_kw);
            const int n = n0 + a_r + AROWS * j;
            const bool ok = c_ok && n < p.Co;
            rb[S][j] = dsf_buffer_load4(wbuf, ok ? (uint32_t)((tapf * p.Co + n) * p.Ci + l_c0 + a_k4) * 4u : OOB);
        } else {
            const int j = i - APASS;
            const int wrow = l_tap * p.Ci + l_c0;
            const int kk = b_row + BROWS * j;
            const bool ok = live && b_nok && (l_c0 + kk < p.Ci);
            rb[S][j] = dsf_buffer_load4(wbuf, ok ? (uint32_t)((wrow + kk) * p.Co + n0 + b_n4) * 4u : OOB);
        }
        if (i == NPIECE - 1) {                                      // advance (scalar)
            l_c0 += BKT;
            if (l_c0 >= p.Ci) {
                l_c0 = 0; ++l_tap; ++l_kw;
                if (l_kw == p.KW) { l_kw = 0; ++l_kh; }
            }
        }
    };
    auto load_all = [&](auto SET, bool live) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) load_piece(SET, i, live);
    };
    // group i of register set SET -> LDS stage `buf` (k-major, swizzled): i < APASS an A row group, then the B groups
    auto stage_piece = [&](auto SET, int buf, int i) {
        constexpr int S = decltype(SET)::value;
        float* Asb = As[buf];
        float* Bsb = Bs[buf];
        const int swz = lds_swz<BKT>(a_k4);                         // same for the 4 k-rows of the quad
        if (i < APASS) {
            const int r = (a_r + AROWS * i) ^ swz;
            Asb[(a_k4 + 0) * BMT + r] = ra[S][i].x; Asb[(a_k4 + 1) * BMT + r] = ra[S][i].y;
            Asb[(a_k4 + 2) * BMT + r] = ra[S][i].z; Asb[(a_k4 + 3) * BMT + r] = ra[S][i].w;
        } else if (WT) {
            const int j = i - APASS;
            const int c = (a_r + AROWS * j) ^ swz;
            Bsb[(a_k4 + 0) * BN + c] = rb[S][j].x; Bsb[(a_k4 + 1) * BN + c] = rb[S][j].y;
            Bsb[(a_k4 + 2) * BN + c] = rb[S][j].z; Bsb[(a_k4 + 3) * BN + c] = rb[S][j].w;
        } else {
            const int j = i - APASS;
            const int kk = b_row + BROWS * j;
            *reinterpret_cast<float4*>(&Bsb[kk * BN + (b_n4 ^ lds_swz<BKT>(kk))]) = rb[S][j];
        }
    };
    auto stage = [&](auto SET, int buf) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) stage_piece(SET, buf, i);
    };

    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    if (chunk_lo < chunk_hi) {
        load_all(Set0{}, true);
        load_all(Set1{}, chunk_lo + 1 < chunk_hi);
        stage(Set0{}, 0);
    }
    __syncthreads();
    // chunk c computes from LDS stage (c - lo) & 1; its register set (same parity) is free and receives chunk c + 2;
    // chunk c + 1 moves from the other set to the other stage after the MFMAs
    // The k-steps of a chunk carry, one per step, first the NPIECE loads of chunk c + 2 (into this chunk's own register
    // set, free since its stage) and then the NPIECE LDS stores of chunk c + 1 (other set -> other stage, whose last
    // reader finished before the previous barrier): by the end of the MFMA block only the barrier is left.
    static_assert(2 * NPIECE <= BKT / 2, "loads and stage stores are handed out one per k-step");
    auto body = [&](auto SET, auto OTHER, int chunk) {
        constexpr int buf = decltype(SET)::value;
        const bool live1 = chunk + 1 < chunk_hi, live2 = chunk + 2 < chunk_hi;
        mma_chunk<BN, TM, TN, BKT, true, BMT>(As[buf], Bs[buf], wm * WM, wn * 64, lane, acc, [&](int s) {
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) {
                if (i == s) load_piece(SET, i, live2);              // constant index after unrolling
                if (i + NPIECE == s && live1) stage_piece(OTHER, buf ^ 1, i);
            }
        });
        __syncthreads();
    };
    for (int chunk = chunk_lo; chunk < chunk_hi; chunk += 2) {
        body(Set0{}, Set1{}, chunk);
        if (chunk + 1 < chunk_hi) body(Set1{}, Set0{}, chunk + 1);
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            if (n >= p.Co) continue;
            const float bv = (bias && ks == 0) ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= M) continue;
                if (k_splits > 1) atomicAdd(Y + (int64_t)m * p.Co + n, acc[i][j][r] + bv);
                else Y[(int64_t)m * p.Co + n] = acc[i][j][r] + bv;
            }
        }
}

// dil == 2 twin of the fast path (ConvTranspose2d forward, backward-data of stride-2 convolutions); same pipeline.
// Transposed-convolution gather (virtual input = X upsampled by 2 with zeros, stride 1): output pixels are ordered
// parity-class-major, so a tile shares (oy & 1, ox & 1), only KH*KW/4 taps can meet data and the others are skipped
// tile-wide without touching memory.
template <int BN, int BKT, bool WT>
__global__ __launch_bounds__(256) void igemm_fwd_dil2_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ Y,
                                                             ConvP p, int m_tiles, int n_tiles, int k_splits,
                                                             uint32_t x_bytes, uint32_t w_bytes) {
    constexpr int WM = (BN == 128) ? 64 : 32;
    constexpr int TM = WM / 32, TN = 2;
    constexpr int KQ = BKT / 4, AROWS = 256 / KQ, APASS = BM / AROWS;
    constexpr int B4 = BN / 4, BROWS = 256 / B4, BPASS = BKT / BROWS;
    constexpr int WTPASS = BN / AROWS;               // WT: thread -> (column, k-quad), as in igemm_fwd_fast_kernel
    static_assert(APASS >= 1 && BPASS >= 1 && WTPASS >= 1, "tile / thread mapping");
    __shared__ float As[2][BKT * BM];
    __shared__ float Bs[2][BKT * BN];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (BN == 128) ? (wave >> 1) : wave, wn = (BN == 128) ? (wave & 1) : 0;
    int tile = xcd_contiguous(blockIdx.x, m_tiles * n_tiles * k_splits);
    const int n_tile = tile % n_tiles; tile /= n_tiles;
    const int m_tile = tile % m_tiles; const int ks = tile / m_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int M = p.B * p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t xbuf = dsf_buffer(X, x_bytes), wbuf = dsf_buffer(W, w_bytes);

    const int Hq = p.Ho >> 1, Wq = p.Wo >> 1, Mc = p.B * Hq * Wq;
    auto decode = [&](int m, int& b, int& oy, int& ox) {
        const int cls = m / Mc, r = m % Mc;
        const int qx = r % Wq, q = r / Wq;
        ox = qx * 2 + (cls & 1); oy = (q % Hq) * 2 + (cls >> 1); b = q / Hq;
    };
    int tile_py = -1, tile_px = -1;
    {
        const int c0 = m0 / Mc, c1 = min(m0 + BM - 1, M - 1) / Mc;
        if (c0 == c1) { tile_py = c0 >> 1; tile_px = c0 & 1; }
    }
    const int a_k4 = (t % KQ) * 4, a_r = t / KQ;
    int a_base[APASS], a_iy[APASS], a_ix[APASS];   // a_base = b * Hi (row base), a_iy / a_ix = virtual coords of tap (0,0)
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int m = m0 + a_r + AROWS * i;
        const bool ok = m < M;
        int b, oy, ox;
        decode(ok ? m : 0, b, oy, ox);
        a_iy[i] = ok ? oy - p.pad_h : -0x40000000;
        a_ix[i] = ox - p.pad_w;
        a_base[i] = b * p.Hi;
    }
    const int b_n4 = (t % B4) * 4, b_row = t / B4;
    const bool b_nok = n0 + b_n4 < p.Co;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Live taps of this tile: with a uniform parity class only taps kh == (pad_h - py) mod 2 (and likewise kw) can meet
    // data, the others read inserted zeros for every row and are never visited.  The K range of a split is cut in this
    // LIVE chunk space, so split-K partitions carry equal work (cutting the full tap list left some splits empty).
    const int chunks_per_tap = (p.Ci + BKT - 1) / BKT;
    const bool uniform = tile_py >= 0;
    const int kh0 = uniform ? ((p.pad_h - tile_py) & 1) : 0, kw0 = uniform ? ((p.pad_w - tile_px) & 1) : 0;
    const int kstep = uniform ? 2 : 1;
    const int cnt_h = (p.KH - kh0 + kstep - 1) / kstep, cnt_w = (p.KW - kw0 + kstep - 1) / kstep;
    const int n_chunks = cnt_h * cnt_w * chunks_per_tap;
    const int per_split = (n_chunks + k_splits - 1) / k_splits;
    const int chunk_lo = ks * per_split, chunk_hi = min(n_chunks, chunk_lo + per_split);

    float4 ra[2][APASS], rb[2][WT ? WTPASS : BPASS];    // two register sets: see igemm_fwd_fast_kernel
    // wave-uniform walk state of the chunk being loaded (live-chunk index -> live tap -> (kh, kw), channel offset)
    int l_chunk = chunk_lo;
    int l_lt = chunk_lo / chunks_per_tap, l_c0 = (chunk_lo % chunks_per_tap) * BKT;
    int l_kh = kh0 + kstep * (l_lt / cnt_w), l_kw = kw0 + kstep * (l_lt % cnt_w);
    int l_tap = l_kh * p.KW + l_kw;
    const int vH = (p.Hi - 1) * 2 + 1, vW = (p.Wi - 1) * 2 + 1;
    constexpr int NB = WT ? WTPASS : BPASS, NPIECE = APASS + NB;
    static_assert(NPIECE <= BKT / 2, "one loader piece per k-step");
    auto load_piece = [&](auto SET, int i, bool live) {
        constexpr int S = decltype(SET)::value;
        const bool c_ok = live && l_c0 + a_k4 < p.Ci;
        if (i < APASS) {
            const int vy = a_iy[i] + l_kh, vx = a_ix[i] + l_kw;
            const bool ok = c_ok && (unsigned)vy < (unsigned)vH && (unsigned)vx < (unsigned)vW && ((vy | vx) & 1) == 0;
            ra[S][i] = dsf_buffer_load4(xbuf, ok ? (uint32_t)(((a_base[i] + (vy >> 1)) * p.Wi + (vx >> 1)) * p.Ci + l_c0 + a_k4) * 4u : OOB);
        } else if (WT) {
            const int j = i - APASS;
            const int tapf = (p.KH - 1 - l_kh) * p.KW + (p.KW - 1 - l_kw);
            const int n = n0 + a_r + AROWS * j;
            const bool ok = c_ok && n < p.Co;
            rb[S][j] = dsf_buffer_load4(wbuf, ok ? (uint32_t)((tapf * p.Co + n) * p.Ci + l_c0 + a_k4) * 4u : OOB);
        } else {
            const int j = i - APASS;
            const int wrow = l_tap * p.Ci + l_c0;
            const int kk = b_row + BROWS * j;
            const bool ok = live && b_nok && (l_c0 + kk < p.Ci);
            rb[S][j] = dsf_buffer_load4(wbuf, ok ? (uint32_t)((wrow + kk) * p.Co + n0 + b_n4) * 4u : OOB);
        }
        if (i == NPIECE - 1 && live) {                              // advance (scalar) to the next live chunk
            l_c0 += BKT; ++l_chunk;
            if (l_c0 >= p.Ci) {
                l_c0 = 0; l_kw += kstep;
                if (l_kw >= p.KW) { l_kw = kw0; l_kh += kstep; }
                l_tap = l_kh * p.KW + l_kw;
            }
        }
    };
    auto load_all = [&](auto SET, bool live) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) load_piece(SET, i, live);
    };
    auto stage_piece = [&](auto SET, int buf, int i) {              // group i of register set SET -> LDS stage `buf`
        constexpr int S = decltype(SET)::value;
        float* Asb = As[buf];
        float* Bsb = Bs[buf];
        const int swz = lds_swz<BKT>(a_k4);
        if (i < APASS) {
            const int r = (a_r + AROWS * i) ^ swz;
            Asb[(a_k4 + 0) * BM + r] = ra[S][i].x; Asb[(a_k4 + 1) * BM + r] = ra[S][i].y;
            Asb[(a_k4 + 2) * BM + r] = ra[S][i].z; Asb[(a_k4 + 3) * BM + r] = ra[S][i].w;
        } else if (WT) {
            const int j = i - APASS;
            const int c = (a_r + AROWS * j) ^ swz;
            Bsb[(a_k4 + 0) * BN + c] = rb[S][j].x; Bsb[(a_k4 + 1) * BN + c] = rb[S][j].y;
            Bsb[(a_k4 + 2) * BN + c] = rb[S][j].z; Bsb[(a_k4 + 3) * BN + c] = rb[S][j].w;
        } else {
            const int j = i - APASS;
            const int kk = b_row + BROWS * j;
            *reinterpret_cast<float4*>(&Bsb[kk * BN + (b_n4 ^ lds_swz<BKT>(kk))]) = rb[S][j];
        }
    };
    auto stage = [&](auto SET, int buf) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) stage_piece(SET, buf, i);
    };

    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    bool cur = l_chunk < chunk_hi;                    // the chunk about to be computed exists
    load_all(Set0{}, cur);
    bool nxt = l_chunk < chunk_hi;                    // ... and the one after it
    load_all(Set1{}, nxt);
    if (cur) stage(Set0{}, 0);
    __syncthreads();
    static_assert(2 * NPIECE <= BKT / 2, "loads and stage stores are handed out one per k-step");
    auto body = [&](auto SET, auto OTHER) {
        constexpr int buf = decltype(SET)::value;
        const bool live1 = nxt, live2 = l_chunk < chunk_hi;        // the walk stands at the chunk two ahead
        mma_chunk<BN, TM, TN, BKT>(As[buf], Bs[buf], wm * WM, wn * 64, lane, acc, [&](int s) {
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) {
                if (i == s) load_piece(SET, i, live2);
                if (i + NPIECE == s && live1) stage_piece(OTHER, buf ^ 1, i);
            }
        });
        __syncthreads();
        cur = nxt; nxt = live2;
    };
    while (cur) {
        body(Set0{}, Set1{});
        if (cur) body(Set1{}, Set0{});
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            if (n >= p.Co) continue;
            const float bv = (bias && ks == 0) ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= M) continue;
                int b, oy, ox;
                decode(m, b, oy, ox);
                const int64_t row = ((int64_t)b * p.Ho + oy) * p.Wo + ox;
                if (k_splits > 1) atomicAdd(Y + row * p.Co + n, acc[i][j][r] + bv);
                else Y[row * p.Co + n] = acc[i][j][r] + bv;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// backward-weights: dW[k][n] += sum_m A[m][k] dY[m][n], reduction over pixels split across blocks
// ------------------------------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(256) void igemm_wrw_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                        float* __restrict__ dW, ConvP p, int k_tiles, int n_tiles,
                                                        int m_per_split) {
    constexpr int BKT = 128;                         // k rows of the output tile
    constexpr int LDA = BKT + 4, LDB = BN + 4;
    constexpr int WM = (BN == 128) ? 64 : 32;
    constexpr int TM = WM / 32, TN = 2;
    constexpr int B4 = BN / 4, BROWS = 256 / B4, BPASS = BK / BROWS;
    __shared__ float As[BK * LDA];                   // [pixel][k]
    __shared__ float Bs[BK * LDB];                   // [pixel][n]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (BN == 128) ? (wave >> 1) : wave, wn = (BN == 128) ? (wave & 1) : 0;
    int bid = blockIdx.x;
    const int k_tile = bid % k_tiles; bid /= k_tiles;
    const int n_tile = bid % n_tiles; const int split = bid / n_tiles;
    const int k0 = k_tile * BKT, n0 = n_tile * BN;
    const int M = p.B * p.Ho * p.Wo, K = p.KH * p.KW * p.Ci;
    const int m_begin = split * m_per_split, m_end = min(M, m_begin + m_per_split);
    const bool vec_a = (p.Ci & 3) == 0, vec_b = (p.Co & 3) == 0;

    // A loader: thread -> 4 consecutive k (same tap when Ci%4==0), 4 pixel rows per chunk
    const int a_k = k0 + (t & 31) * 4;
    int a_kh[4], a_kw[4], a_c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = min(a_k + j, K - 1);
        const int tap = k / p.Ci;
        a_c[j] = k % p.Ci; a_kh[j] = tap / p.KW; a_kw[j] = tap % p.KW;
    }
    const int b_n4 = (t % B4) * 4, b_row = t / B4;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[4], rb[BPASS];
    auto load_chunk = [&](int mc) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int m = mc + (t >> 5) + 8 * i;
            if (m < m_end && a_k < K) {
                const int ox = m % p.Wo, q = m / p.Wo, oy = q % p.Ho, b = q / p.Ho;
                if (vec_a) {
                    const int iy = oy * p.stride + a_kh[0] - p.pad_h, ix = ox * p.stride + a_kw[0] - p.pad_w;
                    if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi)
                        v = *reinterpret_cast<const float4*>(X + (((int64_t)b * p.Hi + iy) * p.Wi + ix) * p.Ci + a_c[0]);
                } else {
                    float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int iy = oy * p.stride + a_kh[j] - p.pad_h, ix = ox * p.stride + a_kw[j] - p.pad_w;
                        if (a_k + j < K && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi)
                            e[j] = X[(((int64_t)b * p.Hi + iy) * p.Wi + ix) * p.Ci + a_c[j]];
                    }
                    v = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int m = mc + b_row + BROWS * i, n = n0 + b_n4;
            if (m < m_end && n < p.Co) {
                const float* src = dY + (int64_t)m * p.Co + n;
                if (vec_b) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (n + 1 < p.Co) v.y = src[1];
                    if (n + 2 < p.Co) v.z = src[2];
                    if (n + 3 < p.Co) v.w = src[3];
                }
            }
            rb[i] = v;
        }
    };

    if (m_begin < m_end) load_chunk(m_begin);
    for (int mc = m_begin; mc < m_end; mc += BK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&As[((t >> 5) + 8 * i) * LDA + (t & 31) * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BPASS; ++i) *reinterpret_cast<float4*>(&Bs[(b_row + BROWS * i) * LDB + b_n4]) = rb[i];
        __syncthreads();
        if (mc + BK < m_end) load_chunk(mc + BK);
#pragma unroll 4
        for (int kk = 0; kk < BK; kk += 2) {
            const int r = kk + (lane >> 5);
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[r * LDA + wm * WM + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[r * LDB + wn * 64 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            if (n >= p.Co) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (k < K) atomicAdd(dW + (int64_t)k * p.Co + n, acc[i][j][r]);
            }
        }
}

// fast path of backward-weights: Ci % 4 == 0, Co % 4 == 0, < 2^31 elements.  Pixel decode by
// multiply-shift (magic = ceil(2^40 / d), exact for n*d < 2^40), clamped-address loads + select instead of
// branches; otherwise identical to igemm_wrw_kernel.
__device__ __forceinline__ uint32_t fast_div(uint32_t n, uint64_t magic) { return (uint32_t)(((uint64_t)n * magic) >> 40); }

// Pipelined like igemm_fwd_fast_kernel (buffer loads with hardware zero fill, two LDS stages and one barrier per
// BKT-pixel chunk, register-double-buffered MFMA fragments); both tiles are stored as loaded (pixel-major rows of 128
// k / BN n values), so the float4 LDS stores and the 32-wide fragment reads are conflict-free without a swizzle.
template <int BN, int BKT>
__global__ __launch_bounds__(256) void igemm_wrw_fast_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                             float* __restrict__ dW, ConvP p, int k_tiles, int n_tiles,
                                                             int n_splits, int m_per_split, uint64_t magic_wo,
                                                             uint64_t magic_ho, uint32_t x_bytes, uint32_t dy_bytes) {
    constexpr int KT = 128;                          // k values (tap, channel) per tile = BM of the block GEMM
    constexpr int WM = (BN == 128) ? 64 : 32;
    constexpr int TM = WM / 32, TN = 2;
    constexpr int APASS = BKT / 8;                   // thread -> (pixel row t >> 5 (+ 8 i), k-quad t & 31)
    constexpr int B4 = BN / 4, BROWS = 256 / B4, BPASS = BKT / BROWS;
    static_assert(KT == BM && APASS >= 1 && BPASS >= 1, "tile / thread mapping");
    __shared__ float As[2][BKT * KT];
    __shared__ float Bs[2][BKT * BN];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (BN == 128) ? (wave >> 1) : wave, wn = (BN == 128) ? (wave & 1) : 0;
    int tile = xcd_contiguous(blockIdx.x, k_tiles * n_tiles * n_splits);
    const int k_tile = tile % k_tiles; tile /= k_tiles;
    const int n_tile = tile % n_tiles; const int split = tile / n_tiles;
    const int k0 = k_tile * KT, n0 = n_tile * BN;
    const int M = p.B * p.Ho * p.Wo, K = p.KH * p.KW * p.Ci;
    const int m_begin = split * m_per_split, m_end = min(M, m_begin + m_per_split);
    const __amdgpu_buffer_rsrc_t xbuf = dsf_buffer(X, x_bytes), ybuf = dsf_buffer(dY, dy_bytes);

    const int a_k = k0 + (t & 31) * 4;
    const bool a_kok = a_k < K;
    const int a_tap = min(a_k, K - 1) / p.Ci;
    const int a_c = min(a_k, K - 1) % p.Ci, a_kh = a_tap / p.KW, a_kw = a_tap % p.KW;
    const int b_n4 = (t % B4) * 4, b_row = t / B4;
    const bool b_nok = n0 + b_n4 < p.Co;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // two register sets: chunk c + 2 is requested while chunk c + 1 waits in the other set for its LDS stage
    float4 ra[2][APASS], rb[2][BPASS];
    constexpr int NPIECE = APASS + BPASS;
    static_assert(NPIECE <= BKT / 2, "one loader piece per k-step");
    auto load_piece = [&](auto SET, int i, int mc) {                 // rows past m_end get the out-of-range offset (zeros)
        constexpr int S = decltype(SET)::value;
        if (i < APASS) {
            const int m = mc + (t >> 5) + 8 * i;
            const uint32_t mm = (uint32_t)min(m, M - 1);
            const uint32_t q = fast_div(mm, magic_wo);
            const int ox = (int)(mm - q * (uint32_t)p.Wo);
            const uint32_t b = fast_div(q, magic_ho);
            const int oy = (int)(q - b * (uint32_t)p.Ho);
            const int iy = oy * p.stride + a_kh - p.pad_h, ix = ox * p.stride + a_kw - p.pad_w;
            const bool ok = a_kok && m < m_end && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            ra[S][i] = dsf_buffer_load4(xbuf, ok ? (uint32_t)((((int)b * p.Hi + iy) * p.Wi + ix) * p.Ci + a_c) * 4u : OOB);
        } else {
            const int j = i - APASS;
            const int m = mc + b_row + BROWS * j;
            const bool ok = b_nok && m < m_end;
            rb[S][j] = dsf_buffer_load4(ybuf, ok ? (uint32_t)(m * p.Co + n0 + b_n4) * 4u : OOB);
        }
    };
    auto load_all = [&](auto SET, int mc) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) load_piece(SET, i, mc);
    };
    auto stage_piece = [&](auto SET, int buf, int i) {
        constexpr int S = decltype(SET)::value;
        if (i < APASS) *reinterpret_cast<float4*>(&As[buf][((t >> 5) + 8 * i) * KT + (t & 31) * 4]) = ra[S][i];
        else *reinterpret_cast<float4*>(&Bs[buf][(b_row + BROWS * (i - APASS)) * BN + b_n4]) = rb[S][i - APASS];
    };
    auto stage = [&](auto SET, int buf) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) stage_piece(SET, buf, i);
    };

    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    if (m_begin < m_end) {
        load_all(Set0{}, m_begin);
        load_all(Set1{}, m_begin + BKT);
        stage(Set0{}, 0);
    }
    __syncthreads();
    static_assert(2 * NPIECE <= BKT / 2, "loads and stage stores are handed out one per k-step");
    auto body = [&](auto SET, auto OTHER, int mc) {
        constexpr int buf = decltype(SET)::value;
        const bool live1 = mc + BKT < m_end;
        mma_chunk<BN, TM, TN, BKT, false>(As[buf], Bs[buf], wm * WM, wn * 64, lane, acc, [&](int s) {
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) {
                if (i == s) load_piece(SET, i, mc + 2 * BKT);
                if (i + NPIECE == s && live1) stage_piece(OTHER, buf ^ 1, i);
            }
        });
        __syncthreads();
    };
    for (int mc = m_begin; mc < m_end; mc += 2 * BKT) {
        body(Set0{}, Set1{}, mc);
        if (mc + BKT < m_end) body(Set1{}, Set0{}, mc + BKT);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            if (n >= p.Co) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (k < K) atomicAdd(dW + (int64_t)k * p.Co + n, acc[i][j][r]);
            }
        }
}

}  // namespace

static int conv_forward_impl(const float* X, const float* W, const float* bias, float* Y, int B, int Hi, int Wi, int Ci, int Ho,
                             int Wo, int Co, int KH, int KW, int stride, int dil, int pad_h, int pad_w, int w_fwd_layout,
                             dsf_stream_t stream) {
    DSF_CHECK_ARG(X && W && Y && B >= 0 && Hi > 0 && Wi > 0 && Ci > 0 && Ho > 0 && Wo > 0 && Co > 0 && KH > 0 && KW > 0);
    DSF_CHECK_ARG(stride >= 1 && dil >= 1 && (stride == 1 || dil == 1));
    if (B == 0) return DSF_OK;
    ConvP p = {B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, dil, pad_h, pad_w};
    const int64_t M = (int64_t)B * Ho * Wo;
    DSF_CHECK_ARG(M < (1ll << 31));
    const int bn = (Co > 64) ? 128 : 64;
    const int n_tiles = (Co + bn - 1) / bn;
    const bool flat = Ci < BK && dil == 1;
    const int64_t x_bytes = (int64_t)B * Hi * Wi * Ci * 4, w_bytes = (int64_t)KH * KW * Ci * Co * 4;
    const bool fast = dil == 1 && !flat && (Ci & 3) == 0 && (Co & 3) == 0 && x_bytes < 0xFFFFFFF0ll && w_bytes < 0xFFFFFFF0ll;
    // 64-row tiles (fast kernel, BN 128) when 128-row tiles would leave fewer than 2 per CU
    static const int bm_env = [] { const char* e = getenv("DSF_CONV_BM"); return e ? atoi(e) : 0; }();
    const int bmt = (fast && bn == 128 && (bm_env == 64 || (bm_env != 128 && ((M + BM - 1) / BM) * n_tiles < 512))) ? 64 : BM;
    const int m_tiles = (int)((M + bmt - 1) / bmt);
    const int perm = (dil > 1 && Ho % dil == 0 && Wo % dil == 0) ? 1 : 0;
    const int n_chunks = flat ? (KH * KW * Ci + BK - 1) / BK : KH * KW * ((Ci + BK - 1) / BK);
    const int live_chunks = perm ? n_chunks / (dil * dil) : n_chunks;
    const bool fast2 = dil == 2 && stride == 1 && perm && (Ci & 3) == 0 && (Co & 3) == 0 && Ci >= BK &&
                       x_bytes < 0xFFFFFFF0ll && w_bytes < 0xFFFFFFF0ll;
    static const int bk_env = [] { const char* e = getenv("DSF_CONV_BK"); return e ? atoi(e) : 0; }();
    // pipeline stage depth of the fast kernel: 16 (4-5 workgroups per CU) once there are >= 4 tiles per CU, 32 (2 per
    // CU, half the barriers, and half the split-K partials for small-M layers) below that.  Measured per layer on
    // MI355X (B=32 ResNet-18 shapes): 64x64 maps 124 vs 116 TFLOP/s with 16; 8x8..32x32 maps 85 vs 62 with 32.
    const int bkt = !(fast || fast2) ? 32 : (bk_env == 16 || bk_env == 32) ? bk_env
                    : (m_tiles * n_tiles >= 1024 && !(fast2 && bn == 64) ? 16 : 32);   // (dil2, BN 64: 75 vs 102 us with 32)
    int k_splits = 1;
    if (m_tiles * n_tiles < 384) {                      // fewer tiles than ~1.5 per CU: split K to fill the chip
        // resident workgroups per CU: 2 with 32-deep stages (64 KiB of LDS each), 4 with 16-deep ones
        const int slots = ((fast || fast2) && bkt == 16) ? 1024 : 512;
        k_splits = (slots + m_tiles * n_tiles / 2) / (m_tiles * n_tiles);
        if (k_splits > live_chunks / 4) k_splits = live_chunks / 4;
        if (k_splits < 1) k_splits = 1;
    }
    static const int ks_env = [] { const char* e = getenv("DSF_CONV_SPLITS"); return e ? atoi(e) : 0; }();   // tuning aid
    if (ks_env > 0) k_splits = ks_env > live_chunks ? live_chunks : ks_env;
    if (dsf_deterministic()) k_splits = 1;                               // no float atomics in the epilogue
    if (k_splits > 1 &&
        dsf_zero_async(Y, sizeof(float) * (size_t)M * Co, (hipStream_t)stream) != hipSuccess) return DSF_ERR_LAUNCH;
    const dim3 grid(m_tiles * n_tiles * k_splits);
    if (fast2) {
#define DSF_LAUNCH_DIL2(BNv, BKv, WTv) hipLaunchKernelGGL((igemm_fwd_dil2_kernel<BNv, BKv, WTv>), grid, dim3(256), 0, (hipStream_t)stream, \
                                                         X, W, bias, Y, p, m_tiles, n_tiles, k_splits, (uint32_t)x_bytes, (uint32_t)w_bytes)
#define DSF_LAUNCH_DIL2_WT(BNv, BKv) do { if (w_fwd_layout) DSF_LAUNCH_DIL2(BNv, BKv, true); else DSF_LAUNCH_DIL2(BNv, BKv, false); } while (0)
        if (bn == 128) { if (bkt == 16) DSF_LAUNCH_DIL2_WT(128, 16); else DSF_LAUNCH_DIL2_WT(128, 32); }
        else { if (bkt == 16) DSF_LAUNCH_DIL2_WT(64, 16); else DSF_LAUNCH_DIL2_WT(64, 32); }
#undef DSF_LAUNCH_DIL2_WT
#undef DSF_LAUNCH_DIL2
        return dsf_launch_status();
    }
    if (w_fwd_layout && !fast) return DSF_ERR_UNSUPPORTED;
    if (fast) {
        static const int lds_pad = [] { const char* e = getenv("DSF_CONV_LDS_PAD"); return e ? atoi(e) : 0; }();      // tuning aid
#define DSF_LAUNCH_FAST(BNv, WTv, BKv, BMv) hipLaunchKernelGGL((igemm_fwd_fast_kernel<BNv, WTv, BKv, BMv>), grid, dim3(256), (BMv == 64 ? lds_pad : 0), \
                                                              (hipStream_t)stream, X, W, bias, Y, p, m_tiles, n_tiles, k_splits,      \
                                                              (uint32_t)x_bytes, (uint32_t)w_bytes)
#define DSF_LAUNCH_FAST_BK(BNv, WTv, BMv) do { if (bkt == 16) DSF_LAUNCH_FAST(BNv, WTv, 16, BMv); else DSF_LAUNCH_FAST(BNv, WTv, 32, BMv); } while (0)
        if (bn == 128 && bmt == 64) { if (w_fwd_layout) DSF_LAUNCH_FAST_BK(128, true, 64); else DSF_LAUNCH_FAST_BK(128, false, 64); }
        else if (bn == 128) { if (w_fwd_layout) DSF_LAUNCH_FAST_BK(128, true, 128); else DSF_LAUNCH_FAST_BK(128, false, 128); }
        else { if (w_fwd_layout) DSF_LAUNCH_FAST_BK(64, true, 128); else DSF_LAUNCH_FAST_BK(64, false, 128); }
#undef DSF_LAUNCH_FAST_BK
#undef DSF_LAUNCH_FAST
        return dsf_launch_status();
    }
#define DSF_LAUNCH_FWD(BNv, FL) hipLaunchKernelGGL((igemm_fwd_kernel<BNv, FL>), grid, dim3(256), 0, (hipStream_t)stream, X, W, \
                                                   bias, Y, p, m_tiles, n_tiles, k_splits, perm)
    if (bn == 128) { if (flat) DSF_LAUNCH_FWD(128, true); else DSF_LAUNCH_FWD(128, false); }
    else { if (flat) DSF_LAUNCH_FWD(64, true); else DSF_LAUNCH_FWD(64, false); }
#undef DSF_LAUNCH_FWD
    return dsf_launch_status();
}

extern "C" int dsf_conv_igemm_forward(const float* X, const float* W, const float* bias, float* Y, int B, int Hi, int Wi,
                                      int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int dil, int pad_h,
                                      int pad_w, dsf_stream_t stream) {
    return conv_forward_impl(X, W, bias, Y, B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, dil, pad_h, pad_w, 0, stream);
}

// Backward-data of a stride-1 convolution straight from the layer's FORWARD weight operand W_fwd [KH][KW][Cin][Cout]:
// dX[b,y,x,ci] = sum_{kh,kw,co} dY[b, y+pad-kh, x+pad-kw, co] * W_fwd[kh][kw][ci][co].  (Ci = Cout, Co = Cin here.)
extern "C" int dsf_conv_igemm_bwd_data_s1(const float* dY, const float* W_fwd, float* dX, int B, int H, int Wd, int Cout,
                                          int Cin, int KH, int KW, int pad_h, int pad_w, dsf_stream_t stream) {
    const int Ho = H + 2 * pad_h - KH + 1, Wo = Wd + 2 * pad_w - KW + 1;          // dY spatial size
    return conv_forward_impl(dY, W_fwd, nullptr, dX, B, Ho, Wo, Cout, H, Wd, Cin, KH, KW, 1, 1, KH - 1 - pad_h, KW - 1 - pad_w, 1,
                             stream);
}

// Same GEMM with the weights given TRANSPOSED AND TAP-FLIPPED, Wt[KH][KW][Co][Ci] with tap (kh, kw) stored at
// (KH-1-kh, KW-1-kw): that is what a layer's own parameter memory looks like from its other direction, so
// ConvTranspose2d forward and the backward-data of strided convolutions need no re-laid weight copy.
// Only the vectorised paths implement it (DSF_ERR_UNSUPPORTED otherwise; callers then re-lay the weights).
extern "C" int dsf_conv_igemm_forward_wt(const float* X, const float* Wt, const float* bias, float* Y, int B, int Hi, int Wi,
                                         int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int dil, int pad_h,
                                         int pad_w, dsf_stream_t stream) {
    return conv_forward_impl(X, Wt, bias, Y, B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, dil, pad_h, pad_w, 1, stream);
}

extern "C" int dsf_conv_igemm_wrw(const float* X, const float* dY, float* dW, int B, int Hi, int Wi, int Ci, int Ho, int Wo,
                                  int Co, int KH, int KW, int stride, int pad_h, int pad_w, int accumulate,
                                  dsf_stream_t stream) {
    DSF_CHECK_ARG(X && dY && dW && B >= 0 && Hi > 0 && Wi > 0 && Ci > 0 && Ho > 0 && Wo > 0 && Co > 0 && KH > 0 && KW > 0);
    const int K = KH * KW * Ci;
    // accumulate != 0: dW += ... (the caller zeroed it, e.g. one memset over a whole gradient pool, or wants accumulation)
    if (!accumulate &&
        dsf_zero_async(dW, sizeof(float) * (size_t)K * Co, (hipStream_t)stream) != hipSuccess) return DSF_ERR_LAUNCH;
    if (B == 0) return DSF_OK;
    ConvP p = {B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, 1, pad_h, pad_w};
    const int64_t M = (int64_t)B * Ho * Wo;
    DSF_CHECK_ARG(M < (1ll << 31));
    const int k_tiles = (K + 127) / 128;
    const int bn = (Co > 64) ? 128 : 64;
    const int n_tiles = (Co + bn - 1) / bn;
    const int64_t x_bytes = (int64_t)B * Hi * Wi * Ci * 4, dy_bytes = M * Co * 4;
    const bool fast = (Ci & 3) == 0 && (Co & 3) == 0 && x_bytes < 0xFFFFFFF0ll && dy_bytes < 0xFFFFFFF0ll;
    static const int bk_env = [] { const char* e = getenv("DSF_WRW_BK"); return e ? atoi(e) : 0; }();
    // 16-pixel stages (4-5 workgroups per CU) pay off on long reductions; short ones (8x8, 16x16 maps) keep 32
    const int bkt = !fast ? 32 : (bk_env == 16 || bk_env == 32) ? bk_env : (M >= 32768 ? 16 : 32);
    // split the pixel reduction so that one round of resident workgroups covers the chip (4 per CU with 16-pixel
    // stages, 2 per CU with 32-pixel ones); at least 4 chunks per split
    static const int wg_env = [] { const char* e = getenv("DSF_WRW_WGS"); return e ? atoi(e) : 0; }();          // tuning aid
    int splits = (wg_env > 0 ? wg_env : (bkt == 16 ? 1024 : 512)) / (k_tiles * n_tiles);
    if (splits < 1 || dsf_deterministic()) splits = 1;                  // deterministic mode: one workgroup walks all pixels of its tile
    int64_t per = (M + splits - 1) / splits;
    per = ((per + BK - 1) / BK) * BK;
    if (per < 4 * BK) per = 4 * BK;
    splits = (int)((M + per - 1) / per);
    if (fast) {
        const uint64_t mwo = ((1ull << 40) + Wo - 1) / Wo, mho = ((1ull << 40) + Ho - 1) / Ho;
#define DSF_LAUNCH_WRW(BNv, BKv) hipLaunchKernelGGL((igemm_wrw_fast_kernel<BNv, BKv>), dim3(k_tiles * n_tiles * splits), dim3(256), 0, \
                                                   (hipStream_t)stream, X, dY, dW, p, k_tiles, n_tiles, splits, (int)per, mwo, mho,     \
                                                   (uint32_t)x_bytes, (uint32_t)dy_bytes)
        if (bn == 128) { if (bkt == 16) DSF_LAUNCH_WRW(128, 16); else DSF_LAUNCH_WRW(128, 32); }
        else { if (bkt == 16) DSF_LAUNCH_WRW(64, 16); else DSF_LAUNCH_WRW(64, 32); }
#undef DSF_LAUNCH_WRW
        return dsf_launch_status();
    }
    if (bn == 128)
        hipLaunchKernelGGL(igemm_wrw_kernel<128>, dim3(k_tiles * n_tiles * splits), dim3(256), 0, (hipStream_t)stream, X, dY,
                           dW, p, k_tiles, n_tiles, (int)per);
    else
        hipLaunchKernelGGL(igemm_wrw_kernel<64>, dim3(k_tiles * n_tiles * splits), dim3(256), 0, (hipStream_t)stream, X, dY,
                           dW, p, k_tiles, n_tiles, (int)per);
    return dsf_launch_status();
}
