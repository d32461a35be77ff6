// fp32 convolution on the bf16 matrix cores by exact operand splitting ("bf16x6").
//
// gfx950 has no TF32/xf32 MFMA and its fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate.  An fp32 value
// is the exact sum of three bf16 values  x = h + m + l  (8 + 8 + 8 significand bits: h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m); both subtractions are exact in fp32), so a product is  a b = ah bh + (ah bm + am bh) +
// (ah bl + al bh + am bm) + O(2^-26 |a b|):  six bf16 MFMAs with fp32 accumulation reproduce the fp32 product to
// better than one fp32 rounding (the dropped terms am bl, al bm, al bl are <= 2^-26 relative), at 16 / 6 = 2.7x the
// fp32 MFMA rate.  Every bf16 product is exact in the fp32 accumulator (8 x 8 bit significands); what remains is the
// same fp32 accumulation error the fp32 MFMA kernel has.  tests/test_gpu_conv.py compares both against float64.
//
// Data path of one workgroup (128 pixels x 128 channels of output, 4 waves of 64 x 64):
//  * activations stay fp32 in HBM; the loader reads float4 (4 consecutive channels of one pixel, buffer loads with
//    hardware zero fill for padding taps) and splits them into the three planes on its way to LDS;
//  * weights are split ONCE per optimizer step by x6_split_weights_kernel into an image that is already laid out as the
//    LDS tiles of this kernel (per tap, 16-channel chunk, 128-wide n tile: 3 planes x [k-group 2][n 128][8 bf16]), so a
//    tile's B operand is three contiguous 4 KiB reads per chunk and needs no conversion in the hot loop;
//  * LDS tiles are [plane][k-group][row] arrays of 16-byte granules (8 consecutive k of one row): the fragment of
//    v_mfma_f32_32x32x16_bf16 (lane = row + 32 k-group, 8 consecutive k) is one conflict-free ds_read_b128;
//  * two LDS stages of one 16-deep chunk each, two register sets for the global loads (chunk c + 2 in flight while
//    chunk c + 1 waits for its stage), loader pieces handed out between the 24 MFMAs of a chunk.
// Kernels: igemm_x6_kernel / igemm_x6b_kernel<BN, DIL2, BMT> (forward, backward-data, transposed convolution: one gather per tap),
// igemm_x6p_kernel (3 x 3 stride-1 layers and the parity classes of 4 x 4 stride-2 transposed ones: the input patch of a chunk is
// staged ONCE for all taps; bit-identical to the gather kernels; round 6: 4 x 4 stride-2 convolutions by INPUT parity classes), igemm_wrw_x6_kernel (backward-weights: both operands are
// activations and are split on the fly; LDS transposes with ds_read_b64_tr_b16), igemm_wrw_x6p_kernel (the same for 3 x 3 stride-1
// layers from a ring of input rows with halo, all nine taps per workgroup), x6_split_weights(_multi)_kernel.  MI355X, B = 32 ResNet-18 two-stage step: 175-200 TFLOP/s fp32-equivalent on the
// 64x64-map layers against 120-134 for the fp32 MFMA kernels of conv.hip (whose peak is 157.3); 28.0 -> 21.1 ms per step.
#include "common.h"

#include <type_traits>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct X6P { int B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad_h, pad_w; };

constexpr int XBK = 16;
constexpr uint32_t X_OOB = 0xFFFFFFFFu;
// Weight image (HBM): per (tap, 16-channel chunk, n tile) one block of 3 planes x [k-group 2][n BN] granules of 16 bytes
// (8 consecutive k of one n).  BN = 128 output channels per tile, 64 when the layer has at most 64.
__host__ __device__ constexpr int x6_bn(int Cn) { return Cn > 64 ? 128 : 64; }
// LDS copy of a plane: the second k-group starts 4 granules (16 banks) later, so that the 16-lane groups of the loader's
// ds_write_b64 (4 rows x 4 k-quads: both k-groups of a row) do not meet on a bank (they were 2-way: 55 M conflict cycles
// of 110 M LDS cycles on the 488->256 layer)
template <int ROWS> struct LdsPlane { static constexpr int KG = ROWS + 4, SIZE = 2 * KG; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t x6_buffer(const void* ptr, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 x6_load16(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
    const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
    return __builtin_bit_cast(u32x4, raw);
}
__device__ __forceinline__ int x6_xcd_contiguous(int bid, int total) {
    const int q = total >> 3, rem = total & 7, xcd = bid & 7, local = bid >> 3;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
}

#ifdef X6_STAMP      // diagnostic build only (tools/x6/wrw_stamps.py): per-workgroup wall-clock stamps; the shipped library has none of this
__device__ unsigned long long* x6_stamp_buf = nullptr;
__device__ __forceinline__ void x6_stamp(int slot) {
    if (x6_stamp_buf && threadIdx.x == 0) {
        unsigned long long* row = x6_stamp_buf + (size_t)blockIdx.x * 8;
        row[slot] = wall_clock64();                                     // s_memrealtime: 100 MHz, chip-wide
        if (slot == 0) { row[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); row[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20); }   // HW_ID, XCC_ID
    }
}
#else
__device__ __forceinline__ void x6_stamp(int) {}
#endif

// Dilation-2 gathers (transposed convolutions, backward-data of stride-2 layers) order their rows by output parity class, and the
// classes differ in live taps (k = 1: one class carries the whole GEMM, three store zeros; k = 3: 4 / 2 / 2 / 1 taps).  In
// launch order the heavy class would fill two of the eight XCDs and leave six idle (B = 192, 32x32x512 -> 64x64x256 k1: 1090 us):
// deal the tiles of the four classes round-robin instead (a bijection on [0, m_tiles)).
__device__ __forceinline__ int x6_dil2_tile(int j, int m_tiles) {
    const int q = m_tiles >> 2;
    return j < 4 * q ? (j & 3) * q + (j >> 2) : j;
}

// two floats -> packed bf16 pair (round to nearest even; v_cvt_pk_bf16_f32), and back
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float pk_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float pk_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// x = h + m + l exactly (to 2^-26 |x|): four floats -> three planes of four bf16
__device__ __forceinline__ void split4(const u32x4 raw, uint2& h, uint2& m, uint2& l) {
    const f32x4 v = __builtin_bit_cast(f32x4, raw);
    h.x = pk_bf16(v[0], v[1]); h.y = pk_bf16(v[2], v[3]);
    const float r0 = v[0] - pk_lo(h.x), r1 = v[1] - pk_hi(h.x), r2 = v[2] - pk_lo(h.y), r3 = v[3] - pk_hi(h.y);
    m.x = pk_bf16(r0, r1); m.y = pk_bf16(r2, r3);
    l.x = pk_bf16(r0 - pk_lo(m.x), r1 - pk_hi(m.x)); l.y = pk_bf16(r2 - pk_lo(m.y), r3 - pk_hi(m.y));
}

// ------------------------------------------------------------------------------------------------
// Weight image.  mode 0 (forward): n = output channel, k = input channel, taps in order;  W is the kernel layout
// [KH][KW][Ci][Co].  mode 1 (backward-data of a stride-1 convolution): n = input channel, k = output channel, taps
// flipped -- the image is that of the transposed, 180-degree-rotated filter, so the main kernel is the same.
// One thread per granule (8 consecutive k of one n), all three planes.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void x6_split_granule(const float* __restrict__ W, uint4* __restrict__ img, int KH, int KW, int Ci,
                                                 int Co, int mode, int chunks, int n_tiles, int bn, int64_t g) {
    const int nl = (int)(g % bn), kg = (int)((g / bn) & 1);
    int64_t blk = g / (2 * bn);
    const int n_tile = (int)(blk % n_tiles); blk /= n_tiles;
    const int chunk = (int)(blk % chunks); const int tap = (int)(blk / chunks);
    const int n = n_tile * bn + nl, k0 = chunk * XBK + kg * 8;
    const int Cn = mode ? Ci : Co, Ck = mode ? Co : Ci;
    const int kh = tap / KW, kw = tap % KW;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        float x = 0.f;
        if (n < Cn && k < Ck)
            x = mode ? W[((int64_t)((KH - 1 - kh) * KW + (KW - 1 - kw)) * Ci + n) * Co + k]
                     : W[((int64_t)(kh * KW + kw) * Ci + k) * Co + n];
        v[e] = x;
    }
    uint2 h0, m0, l0, h1, m1, l1;
    split4(__builtin_bit_cast(u32x4, (f32x4){v[0], v[1], v[2], v[3]}), h0, m0, l0);
    split4(__builtin_bit_cast(u32x4, (f32x4){v[4], v[5], v[6], v[7]}), h1, m1, l1);
    const int64_t base = (g / (2 * bn)) * (6 * bn) + kg * bn + nl;
    img[base] = make_uint4(h0.x, h0.y, h1.x, h1.y);
    img[base + 2 * bn] = make_uint4(m0.x, m0.y, m1.x, m1.y);
    img[base + 4 * bn] = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

__global__ __launch_bounds__(256) void x6_split_weights_kernel(const float* __restrict__ W, uint4* __restrict__ img, int KH,
                                                               int KW, int Ci, int Co, int mode, int chunks, int n_tiles,
                                                               int bn, int64_t granules) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < granules) x6_split_granule(W, img, KH, KW, Ci, Co, mode, chunks, n_tiles, bn, g);
}

// Four consecutive granules of an image (the same k-group and block: bn is a multiple of 4) with 16-byte loads (round 6; the
// per-granule form above issues eight 4-byte loads per granule: the whole-network launch ran at 2.7 TB/s).  mode 0: eight float4
// along n, one per k; mode 1: two float4 along k per granule.  The channel counts are multiples of 4 (the convolution's own
// requirement), so a float4 is inside the matrix or outside it as a whole.  Same values, same split: bit-identical images.
// wave-local exchange through LDS: the LDS unit serves one wave's instructions in order, so a ds_read issued after a ds_write of the same
// wave sees it; only the COMPILER must keep the order (a wavefront-scope fence would also wait for the global stores in flight)
#define X6_WAVE_LDS_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)
// (out[plane][q]: the high / middle / low granule q; the image index of out[plane][q] is base + q + 2 bn plane)
__device__ __forceinline__ int64_t x6_split_granules4(const float* __restrict__ W, int KH, int KW, int Ci, int Co,
                                                      int mode, int chunks, int n_tiles, int bn, int64_t g, uint4 (&out)[3][4]) {
    const int nl = (int)(g % bn), kg = (int)((g / bn) & 1);
    int64_t blk = g / (2 * bn);
    const int n_tile = (int)(blk % n_tiles); blk /= n_tiles;
    const int chunk = (int)(blk % chunks); const int tap = (int)(blk / chunks);
    const int n = n_tile * bn + nl, k0 = chunk * XBK + kg * 8;
    const int Cn = mode ? Ci : Co, Ck = mode ? Co : Ci;
    const int kh = tap / KW, kw = tap % KW;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 lo[4], hi[4];                                  // [granule]: k0 .. k0 + 3, k0 + 4 .. k0 + 7
    if (mode) {
        const float* wt = W + (int64_t)((KH - 1 - kh) * KW + (KW - 1 - kw)) * Ci * Co;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool row = n + q < Cn;
            lo[q] = (row && k0 < Ck) ? *reinterpret_cast<const f32x4*>(wt + (int64_t)(n + q) * Co + k0) : z;
            hi[q] = (row && k0 + 4 < Ck) ? *reinterpret_cast<const f32x4*>(wt + (int64_t)(n + q) * Co + k0 + 4) : z;
        }
    } else {
        const float* wt = W + (int64_t)(kh * KW + kw) * Ci * Co;
        f32x4 r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (n < Cn && k0 + e < Ck) ? *reinterpret_cast<const f32x4*>(wt + (int64_t)(k0 + e) * Co + n) : z;
#pragma unroll
        for (int q = 0; q < 4; ++q) { lo[q] = (f32x4){r[0][q], r[1][q], r[2][q], r[3][q]}; hi[q] = (f32x4){r[4][q], r[5][q], r[6][q], r[7][q]}; }
    }
    const int64_t base = (g / (2 * bn)) * (6 * bn) + kg * bn + nl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint2 h0, m0, l0, h1, m1, l1;
        split4(__builtin_bit_cast(u32x4, lo[q]), h0, m0, l0);
        split4(__builtin_bit_cast(u32x4, hi[q]), h1, m1, l1);
        out[0][q] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        out[1][q] = make_uint4(m0.x, m0.y, m1.x, m1.y);
        out[2][q] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
    return base;
}

// every image of a network in ONE launch (after an optimizer step): jobs[j] = {W, image, KH, KW, Ci, Co, mode, first granule
// of the job in the launch-wide numbering}, jobs[n_jobs][7] = the total; a thread takes FOUR consecutive granules (every job's
// granule count is a multiple of 4) and finds its job by bisection.
// Stores: a thread's four granules are 64 consecutive bytes per plane, so a store instruction written per thread touches 64
// separate segments per wave (the launch ran at 3.9 TB/s: 154 us for config 2's 30 M weights in both directions).  When a wave's
// 256 granules belong to one job -- job starts are multiples of 128 granules, a wave straddles two jobs at most once per job --
// they go through LDS ([plane][q][lane], rows 68 granules apart: both sides conflict-free) and every store instruction writes 64
// consecutive granules = 1 KB (a 64-granule run never crosses an image block: blocks are 2 bn >= 128 granules): 124-128 us (a
// probe with lane-contiguous stores and no exchange: 107; tools/x6/split_probe.py).  Same values, same places: bit-identical images.
#ifndef X6_SPLIT_OCC                                     // waves per SIMD the register allocation is held to (tools/x6/split_probe.py, config 2's 30 M weights:
#define X6_SPLIT_OCC 4                                   //  2: 137.5 us, 3: 136.0, 4: 124-128, 5: 119.3 but 80 bytes of scratch (the ISA lint refuses it), 6: 174.8;
#endif                                                   //  unbounded the kernel took 134 VGPRs = 3 waves)
__global__ __launch_bounds__(256, X6_SPLIT_OCC) void x6_split_weights_multi_kernel(const int64_t* __restrict__ jobs, int n_jobs) {
    __shared__ uint4 s_x[4][4 * 68];                     // per wave, one plane at a time (the exchange is wave-local: no workgroup barrier)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t total = jobs[(int64_t)n_jobs * 8 + 7];
    int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const bool live = g < total;
    if (!live) g = total - 4;                            // (keeps the wave's control flow uniform up to the stores; stores nothing)
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[(int64_t)mid * 8 + 7] <= g) lo = mid; else hi = mid - 1;
    }
    const int64_t* j = jobs + (int64_t)lo * 8;
    const int KH = (int)j[2], KW = (int)j[3], Ci = (int)j[4], Co = (int)j[5], mode = (int)j[6];
    const int Ck = mode ? Co : Ci, Cn = mode ? Ci : Co, bn = x6_bn(Cn);
    const float* W = reinterpret_cast<const float*>(j[0]);
    uint4* img = reinterpret_cast<uint4*>(j[1]);
    const bool vec = ((Ci | Co) & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
    const int64_t gl = g - j[7];                         // granule within the job
    // the wave's 256 granules: one job, all there, a vectorisable matrix, and 64-granule runs aligned in the job
    const bool uni = __all(live && vec && lo == __builtin_amdgcn_readfirstlane(lo) && ((gl - 4 * lane) & 63) == 0);
    uint4 out[3][4];
    int64_t base = 0;
    if (vec) base = x6_split_granules4(W, KH, KW, Ci, Co, mode, (Ck + XBK - 1) / XBK, (Cn + bn - 1) / bn, bn, gl, out);
    if (uni) {
        const int64_t g0 = gl - 4 * lane;                // the wave's first granule
        int64_t at[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {                    // run r: granules g0 + 64 r + lane, held by thread (64 r + lane) / 4 as its granule (lane & 3)
            const int64_t gi = g0 + 64 * r + lane;
            at[r] = (gi / (2 * bn)) * (6 * bn) + gi % (2 * bn);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s_x[wave][q * 68 + lane] = out[pl][q];
            X6_WAVE_LDS_ORDER();
#pragma unroll
            for (int r = 0; r < 4; ++r) img[at[r] + 2 * bn * pl] = s_x[wave][(lane & 3) * 68 + 16 * r + (lane >> 2)];
            X6_WAVE_LDS_ORDER();
        }
    } else if (live && vec) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            img[base + q] = out[0][q]; img[base + q + 2 * bn] = out[1][q]; img[base + q + 4 * bn] = out[2][q];
        }
    } else if (live) {
#pragma unroll 1
        for (int q = 0; q < 4; ++q)
            x6_split_granule(W, img, KH, KW, Ci, Co, mode, (Ck + XBK - 1) / XBK, (Cn + bn - 1) / bn, bn, gl + q);
    }
}

// ------------------------------------------------------------------------------------------------
// Y[m][n] = bias[n] + sum_{tap, k} X[pixel(m) + tap][k] * Wimage[tap][k][n]      (dil 1, stride s)
// ------------------------------------------------------------------------------------------------
// DIL2: the transposed-convolution gather (virtual input = X upsampled by 2 with zeros, stride 1) of ConvTranspose2d forward
// and of the backward-data pass of stride-2 convolutions.  Output pixels are then numbered parity-class-major
// ((oy & 1, ox & 1) classes of B * Ho/2 * Wo/2 pixels each), so a tile shares its parity, only the taps of matching parity
// can meet data and the walk visits just those (a quarter of a 4x4 filter's taps); Ho and Wo must be even.
// BMT: pixel rows per tile, 128 or (BN 128 only) 64 for the 8x8 .. 32x32 maps: twice the tiles, so those layers need no
// split-K or half of it (every split adds M x Co float atomics to the epilogue).
template <int BN, bool DIL2, int BMT = 128>
__global__ __launch_bounds__(256, 2) void igemm_x6_kernel(const float* __restrict__ X, const uint4* __restrict__ Wimg,
                                                         const float* __restrict__ bias, float* __restrict__ Y, X6P p,
                                                         int m_tiles, int n_tiles, int k_splits, uint32_t x_bytes,
                                                         uint32_t w_bytes) {
    static_assert(BMT == 128 || (BMT == 64 && BN == 128), "tile shapes");
    constexpr int WM = (BN == 128) ? BMT / 2 : 32;       // wave sub-tile WM x 64: 2 x 2 waves (BN 128) or 4 x 1 (BN 64)
    constexpr int TM = WM / 32, TN = 2;
    constexpr int APASS = BMT / 64;                      // float4 loads of the A tile per thread
    using LA = LdsPlane<BMT>;
    using LB = LdsPlane<BN>;
    constexpr int B_GRANULES = 2 * BN;                   // per plane of one image block
    __shared__ uint4 As[2][3 * LA::SIZE];
    __shared__ uint4 Bs[2][3 * LB::SIZE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (BN == 128) ? (wave >> 1) : wave, wn = (BN == 128) ? (wave & 1) : 0;
    int tile = x6_xcd_contiguous(blockIdx.x, m_tiles * n_tiles * k_splits);
    const int n_tile = tile % n_tiles; tile /= n_tiles;
    const int m_tile = DIL2 ? x6_dil2_tile(tile % m_tiles, m_tiles) : tile % m_tiles; const int ks = tile / m_tiles;
    const int m0 = m_tile * BMT, n0 = n_tile * BN;
    const bool b_thread = t < B_GRANULES;                // BN 64: waves 0-1 carry the B tile (wave-uniform)
    const int M = p.B * p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t xbuf = x6_buffer(X, x_bytes), wbuf = x6_buffer(Wimg, w_bytes);

    const int Hq = p.Ho >> 1, Wq = p.Wo >> 1, Mc = p.B * Hq * Wq;           // DIL2: pixels per parity class
    auto decode = [&](int m, int& b, int& oy, int& ox) {
        if (DIL2) {
            const int cls = m / Mc, r = m % Mc;
            const int qx = r % Wq, q = r / Wq;
            ox = qx * 2 + (cls & 1); oy = (q % Hq) * 2 + (cls >> 1); b = q / Hq;
        } else {
            ox = m % p.Wo; const int q = m / p.Wo; oy = q % p.Ho; b = q / p.Ho;
        }
    };
    int tile_py = -1, tile_px = -1;                      // DIL2: the tile's parity class when it has just one
    if (DIL2) {
        const int c0 = m0 / Mc, c1 = min(m0 + BMT - 1, M - 1) / Mc;
        if (c0 == c1) { tile_py = c0 >> 1; tile_px = c0 & 1; }
    }
    // A loader: thread -> (row (t >> 2) + 64 i, k-quad t & 3).  dil 1: a_iy / a_ix = input coordinates of tap (0, 0) and
    // a_base its element offset; DIL2: virtual (upsampled) coordinates and a_base = b * Hi.
    const int a_q = t & 3, a_k4 = a_q * 4, a_r = t >> 2;
    int a_base[APASS], a_iy[APASS], a_ix[APASS];
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int m = m0 + a_r + 64 * i;
        const bool ok = m < M;
        int b, oy, ox;
        decode(ok ? m : 0, b, oy, ox);
        if (DIL2) {
            a_iy[i] = ok ? oy - p.pad_h : -0x40000000;
            a_ix[i] = ox - p.pad_w;
            a_base[i] = b * p.Hi;
        } else {
            a_iy[i] = ok ? oy * p.stride - p.pad_h : -0x40000000;
            a_ix[i] = ox * p.stride - p.pad_w;
            a_base[i] = ((b * p.Hi + oy * p.stride - p.pad_h) * p.Wi + a_ix[i]) * p.Ci + a_k4;
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // K is walked in LIVE chunks: every (tap, 16-channel chunk), or with a uniform DIL2 parity only the taps
    // kh == (pad_h - py) mod 2, kw likewise; split-K ranges are cut in this live space so that they carry equal work
    const int chunks_per_tap = (p.Ci + XBK - 1) / XBK;
    const bool uniform = DIL2 && tile_py >= 0;
    const int kh0 = uniform ? ((p.pad_h - tile_py) & 1) : 0, kw0 = uniform ? ((p.pad_w - tile_px) & 1) : 0;
    const int kstep = uniform ? 2 : 1;
    const int cnt_h = (p.KH - kh0 + kstep - 1) / kstep, cnt_w = (p.KW - kw0 + kstep - 1) / kstep;
    const int n_chunks = cnt_h * cnt_w * chunks_per_tap;
    const int per_split = (n_chunks + k_splits - 1) / k_splits;
    const int chunk_lo = ks * per_split, chunk_hi = min(n_chunks, chunk_lo + per_split);
    const int vH = (p.Hi - 1) * 2 + 1, vW = (p.Wi - 1) * 2 + 1;

    u32x4 ra[2][APASS], rb[2][3];
    // wave-uniform walk state of the chunk being loaded
    // (taps of one channel chunk first, then the next chunk: see igemm_x6b_kernel)
    const int taps_n = cnt_h * cnt_w;
    const int l_lt = chunk_lo % taps_n;
    int l_c0 = (chunk_lo / taps_n) * XBK;
    int l_kh = kh0 + kstep * (l_lt / cnt_w), l_kw = kw0 + kstep * (l_lt % cnt_w);
    constexpr int NPIECE = APASS + 3;
    auto load_piece = [&](auto SET, int i, bool live) {
        constexpr int S = decltype(SET)::value;
        if (i < APASS) {
            bool ok = live && l_c0 + a_k4 < p.Ci;
            uint32_t off;
            if (DIL2) {
                const int vy = a_iy[i] + l_kh, vx = a_ix[i] + l_kw;
                ok = ok && (unsigned)vy < (unsigned)vH && (unsigned)vx < (unsigned)vW && ((vy | vx) & 1) == 0;
                off = (uint32_t)(((a_base[i] + (vy >> 1)) * p.Wi + (vx >> 1)) * p.Ci + l_c0 + a_k4);
            } else {
                ok = ok && (unsigned)(a_iy[i] + l_kh) < (unsigned)p.Hi && (unsigned)(a_ix[i] + l_kw) < (unsigned)p.Wi;
                off = (uint32_t)(a_base[i] + (l_kh * p.Wi + l_kw) * p.Ci + l_c0);
            }
            ra[S][i] = x6_load16(xbuf, ok ? off * 4u : X_OOB);
        } else {
            const int pl = i - APASS;
            const uint32_t dead = (live && b_thread) ? 0u : X_OOB;     // branch-free: (offset | ~0) is the out-of-range offset
            const int blk = (l_kh * p.KW + l_kw) * chunks_per_tap + (l_c0 >> 4);        // image block of (tap, chunk)
            rb[S][pl] = x6_load16(wbuf, ((uint32_t)(blk * n_tiles + n_tile) * (uint32_t)(3 * B_GRANULES * 16) +
                                         (uint32_t)(pl * B_GRANULES + t) * 16u) | dead);
        }
        if (i == NPIECE - 1) {                                         // advance (scalar) to the next live chunk
            l_kw += kstep;
            if (l_kw >= p.KW) {
                l_kw = kw0; l_kh += kstep;
                if (l_kh >= p.KH) { l_kh = kh0; l_c0 += XBK; }
            }
        }
    };
    auto stage_piece = [&](auto SET, int buf, int i) {
        constexpr int S = decltype(SET)::value;
        if (i < APASS) {
            uint2 h, m, l;
            split4(ra[S][i], h, m, l);
            // granule (k-group a_q >> 1, row), half a_q & 1
            uint2* dst = reinterpret_cast<uint2*>(&As[buf][(a_q >> 1) * LA::KG + a_r + 64 * i]) + (a_q & 1);
            dst[0] = h; dst[2 * LA::SIZE] = m; dst[4 * LA::SIZE] = l;
        } else {
            const int pl = i - APASS;
            if (BN == 128 || b_thread)
                Bs[buf][pl * LB::SIZE + (t / BN) * LB::KG + (t % BN)] = __builtin_bit_cast(uint4, rb[S][pl]);
        }
    };

    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    if (chunk_lo < chunk_hi) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) load_piece(Set0{}, i, true);
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) load_piece(Set1{}, i, chunk_lo + 1 < chunk_hi);
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) stage_piece(Set0{}, 0, i);
    }
    __syncthreads();

    const int a_frag = (lane >> 5) * LA::KG + (lane & 31) + wm * WM, b_frag = (lane >> 5) * LB::KG + (lane & 31) + wn * 64;
    auto body = [&](auto SET, auto OTHER, int chunk) {
        constexpr int buf = decltype(SET)::value;
        const bool live2 = chunk + 2 < chunk_hi;
        bf16x8 a[3][TM], b[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[pl][i] = __builtin_bit_cast(bf16x8, As[buf][pl * LA::SIZE + a_frag + i * 32]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[pl][j] = __builtin_bit_cast(bf16x8, Bs[buf][pl * LB::SIZE + b_frag + j * 32]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the six products; the ones on the low planes come last so that every fragment register stays live past the
        // global loads issued in this chunk (a dead fragment register handed to a load forces `s_waitcnt vmcnt(0)` at
        // the next fragment read).  Loader pieces in between: 5 loads of chunk c + 2, then 5 LDS stores of chunk c + 1.
        constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
        int slot = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][i], b[PB[q]][j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int pc = 0; pc < NPIECE; ++pc) {            // constant indices after unrolling
                        // 24 MFMAs (BN 128): a piece behind every second one; 12 (BN 64): behind every one
                        if (slot == (TM == 2 ? 2 * pc + 1 : pc)) load_piece(SET, pc, live2);
                        // chunk c + 1 past the end was loaded as zeros into a stage nobody reads: no branch needed
                        if (slot == (TM == 2 ? 2 * (pc + NPIECE) + 1 : pc + NPIECE)) stage_piece(OTHER, buf ^ 1, pc);
                    }
                    ++slot;
                    __builtin_amdgcn_sched_barrier(0);
                }
        __syncthreads();
    };
    for (int chunk = chunk_lo; chunk < chunk_hi; chunk += 2) {  // chunks come in pairs: one past chunk_hi is all zeros
        body(Set0{}, Set1{}, chunk);
        body(Set1{}, Set0{}, chunk + 1);
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
                int64_t row = m;
                if (DIL2) {
                    int b, oy, ox;
                    decode(m, b, oy, ox);
                    row = ((int64_t)b * p.Ho + oy) * p.Wo + ox;
                }
                if (k_splits > 1) atomicAdd(Y + row * p.Co + n, acc[i][j][r] + bv);
                else Y[row * p.Co + n] = acc[i][j][r] + bv;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// igemm_x6b_kernel: the same GEMM with the B operand (weight image) read STRAIGHT INTO THE MFMA FRAGMENT REGISTERS.
//
// Why: with both operands staged through LDS the 128 x 128 kernel moves, per 16-deep chunk and wave, 12 KB of fragment
// reads for 24 MFMAs (768 matrix-pipe cycles); four waves per chunk plus the 24 KB of tile stores are 72 KB per 768
// cycles = 94 of the CU's 128 LDS bytes per clock -- the kernel is LDS-bandwidth-bound at ~0.6 MFMA utilisation (measured
// 0.587; rocprof SQ_VALU_MFMA_BUSY_CYCLES, profiles/r01_summary.txt).  The weight image is already laid out granule by
// granule as the fragment of v_mfma_f32_32x32x16_bf16 (lane = n + 32 k-group, 8 consecutive k): lane l of a wave reads
// granule (k-group l >> 5, n = 64 wn + 32 j + (l & 31)) of plane pl with ONE coalesced 16-byte buffer load -- no LDS round
// trip, no conversion.  The image block of a (tap, chunk, n tile) is shared by every workgroup of the XCD that walks K in
// step, so these loads hit L2.  LDS then carries the A tile only: 4 x 6 KB of reads + 12 KB of stores per chunk = 47
// bytes per clock, and the B-side ds_write / ds_read instructions disappear from the issue stream.
// Three register sets for the B fragments (loads run two chunks ahead, as the A loads do): the chunk loop is unrolled by
// three, the A stage toggles at run time; a K range is padded with at most two all-zero chunks (out-of-range loads).
// Tiles: BN 128: 128 x 128 (2 x 2 waves of 64 x 64) or 64 x 128 (2 x 2 waves of 32 x 64: the 8x8 .. 32x32 maps);
//        BN 64:  256 x 64  (4 x 1 waves of 64 x 64): the 64-channel layers get the 64 x 64 wave tile too.
// ------------------------------------------------------------------------------------------------
struct X6Ep { const float* scale; const float* shift; const float* residual; int relu; int stats_acc; int live_cls = -1; };   // output epilogue (all null: none); stats_acc: see `stats`; live_cls: see igemm_x6b_kernel

template <int BN, bool DIL2, int BMT>
__global__ __launch_bounds__(256, 2) void igemm_x6b_kernel(const float* __restrict__ X, const uint4* __restrict__ Wimg,
                                                          const float* __restrict__ bias, float* __restrict__ Y, X6P p,
                                                          int m_tiles, int n_tiles, int k_splits, uint32_t x_bytes,
                                                          uint32_t w_bytes, float* __restrict__ stats, X6Ep ep) {
    static_assert((BN == 128 && (BMT == 128 || BMT == 64)) || (BN == 64 && BMT == 256), "tile shapes");
    constexpr int WM = (BN == 128) ? BMT / 2 : 64;       // wave sub-tile WM x 64: 2 x 2 waves (BN 128) or 4 x 1 (BN 64)
    constexpr int TM = WM / 32, TN = 2;
    constexpr int APASS = BMT / 64;                      // float4 loads of the A tile per thread
    using LA = LdsPlane<BMT>;
    constexpr int B_GRANULES = 2 * BN;                   // per plane of one image block
    __shared__ uint4 As[2][3 * LA::SIZE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (BN == 128) ? (wave >> 1) : wave, wn = (BN == 128) ? (wave & 1) : 0;
    int tile = x6_xcd_contiguous(blockIdx.x, m_tiles * n_tiles * k_splits);
    const int n_tile = tile % n_tiles; tile /= n_tiles;
    // DIL2, ep.live_cls >= 0 (a 1 x 1 filter: only that parity class of output pixels has a tap): the launch covers the rows of
    // that class only, and their epilogue also stores the zeros of the three sibling pixels
    const int live = DIL2 ? ep.live_cls : -1;
    const int m_tile = (DIL2 && live < 0) ? x6_dil2_tile(tile % m_tiles, m_tiles) : tile % m_tiles; const int ks = tile / m_tiles;
    const int m0 = m_tile * BMT, n0 = n_tile * BN;
    const __amdgpu_buffer_rsrc_t xbuf = x6_buffer(X, x_bytes), wbuf = x6_buffer(Wimg, w_bytes);

    const int Hq = p.Ho >> 1, Wq = p.Wo >> 1, Mc = p.B * Hq * Wq;           // DIL2: pixels per parity class
    const int M = live >= 0 ? Mc : p.B * p.Ho * p.Wo;                       // rows of the launch
    auto decode = [&](int m, int& b, int& oy, int& ox) {
        if (DIL2) {
            const int cls = live >= 0 ? live : m / Mc, r = live >= 0 ? m : m % Mc;
            const int qx = r % Wq, q = r / Wq;
            ox = qx * 2 + (cls & 1); oy = (q % Hq) * 2 + (cls >> 1); b = q / Hq;
        } else {
            ox = m % p.Wo; const int q = m / p.Wo; oy = q % p.Ho; b = q / p.Ho;
        }
    };
    int tile_py = -1, tile_px = -1;                      // DIL2: the tile's parity class when it has just one
    if (DIL2) {
        const int c0 = live >= 0 ? live : m0 / Mc, c1 = live >= 0 ? live : min(m0 + BMT - 1, M - 1) / Mc;
        if (c0 == c1) { tile_py = c0 >> 1; tile_px = c0 & 1; }
    }
    const int a_q = t & 3, a_k4 = a_q * 4, a_r = t >> 2;
    int a_base[APASS], a_iy[APASS], a_ix[APASS];
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int m = m0 + a_r + 64 * i;
        const bool ok = m < M;
        int b, oy, ox;
        decode(ok ? m : 0, b, oy, ox);
        if (DIL2) {
            a_iy[i] = ok ? oy - p.pad_h : -0x40000000;
            a_ix[i] = ox - p.pad_w;
            a_base[i] = b * p.Hi;
        } else {
            a_iy[i] = ok ? oy * p.stride - p.pad_h : -0x40000000;
            a_ix[i] = ox * p.stride - p.pad_w;
            a_base[i] = ((b * p.Hi + oy * p.stride - p.pad_h) * p.Wi + a_ix[i]) * p.Ci + a_k4;
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int chunks_per_tap = (p.Ci + XBK - 1) / XBK;
    const bool uniform = DIL2 && tile_py >= 0;
    const int kh0 = uniform ? ((p.pad_h - tile_py) & 1) : 0, kw0 = uniform ? ((p.pad_w - tile_px) & 1) : 0;
    const int kstep = uniform ? 2 : 1;
    const int cnt_h = (p.KH - kh0 + kstep - 1) / kstep, cnt_w = (p.KW - kw0 + kstep - 1) / kstep;
    const int n_chunks = cnt_h * cnt_w * chunks_per_tap;
    const int per_split = (n_chunks + k_splits - 1) / k_splits;
    const int chunk_lo = ks * per_split, chunk_hi = min(n_chunks, chunk_lo + per_split);
    const int vH = (p.Hi - 1) * 2 + 1, vW = (p.Wi - 1) * 2 + 1;

    u32x4 ra[2][APASS];
    u32x4 rbf[2][3][TN];                                 // [set][plane][n block]: B fragments as loaded
    // byte offset of this lane's granule inside a plane of an image block: (k-group, n)
    const uint32_t b_lane = (uint32_t)((lane >> 5) * BN + wn * 64 + (lane & 31)) * 16u;
    // wave-uniform walk states: `la` = the chunk whose A tile is being loaded (two ahead), `lb` = the chunk whose B fragments
    // are being loaded (one ahead: they go straight to registers, there is no LDS stage to wait for)
    struct Walk { int c0, kh, kw; };
    // reduction order: the taps of one 16-channel chunk first, then the next chunk -- consecutive iterations re-read the same
    // input rows shifted by one pixel (L1 / L2 hits) instead of coming back to them a whole channel sweep later
    auto advance = [&](Walk& w) {
        w.kw += kstep;
        if (w.kw >= p.KW) {
            w.kw = kw0; w.kh += kstep;
            if (w.kh >= p.KH) { w.kh = kh0; w.c0 += XBK; }
        }
    };
    const int taps_n = cnt_h * cnt_w;
    const int l_lt = chunk_lo % taps_n;
    Walk la = {(chunk_lo / taps_n) * XBK, kh0 + kstep * (l_lt / cnt_w), kw0 + kstep * (l_lt % cnt_w)};
    Walk lb = la;
    auto load_a = [&](auto SET, int i, bool live) {
        constexpr int S = decltype(SET)::value;
        bool ok = live && la.c0 + a_k4 < p.Ci;
        uint32_t off;
        if (DIL2) {
            const int vy = a_iy[i] + la.kh, vx = a_ix[i] + la.kw;
            ok = ok && (unsigned)vy < (unsigned)vH && (unsigned)vx < (unsigned)vW && ((vy | vx) & 1) == 0;
            off = (uint32_t)(((a_base[i] + (vy >> 1)) * p.Wi + (vx >> 1)) * p.Ci + la.c0 + a_k4);
        } else {
            ok = ok && (unsigned)(a_iy[i] + la.kh) < (unsigned)p.Hi && (unsigned)(a_ix[i] + la.kw) < (unsigned)p.Wi;
            off = (uint32_t)(a_base[i] + (la.kh * p.Wi + la.kw) * p.Ci + la.c0);
        }
        ra[S][i] = x6_load16(xbuf, ok ? off * 4u : X_OOB);
        if (i == APASS - 1) advance(la);
    };
    auto load_b = [&](auto SET, int f, bool live) {
        constexpr int S = decltype(SET)::value;
        const int pl = f / TN, j = f % TN;
        const uint32_t dead = live ? 0u : X_OOB;                            // (offset | ~0) is the out-of-range offset: zeros
        const int blk = (lb.kh * p.KW + lb.kw) * chunks_per_tap + (lb.c0 >> 4);
        rbf[S][pl][j] = x6_load16(wbuf, ((uint32_t)(blk * n_tiles + n_tile) * (uint32_t)(3 * B_GRANULES * 16) +
                                         (uint32_t)(pl * B_GRANULES * 16 + j * 32 * 16) + b_lane) | dead);
        if (f == 3 * TN - 1) advance(lb);
    };
    auto stage_piece = [&](auto SET, int buf, int i) {
        constexpr int S = decltype(SET)::value;
        uint2 h, m, l;
        split4(ra[S][i], h, m, l);
        uint2* dst = reinterpret_cast<uint2*>(&As[buf][(a_q >> 1) * LA::KG + a_r + 64 * i]) + (a_q & 1);
        dst[0] = h; dst[2 * LA::SIZE] = m; dst[4 * LA::SIZE] = l;
    };

    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    if (chunk_lo < chunk_hi) {
#pragma unroll
        for (int i = 0; i < APASS; ++i) load_a(Set0{}, i, true);
#pragma unroll
        for (int f = 0; f < 3 * TN; ++f) load_b(Set0{}, f, true);
#pragma unroll
        for (int i = 0; i < APASS; ++i) load_a(Set1{}, i, chunk_lo + 1 < chunk_hi);
#pragma unroll
        for (int i = 0; i < APASS; ++i) stage_piece(Set0{}, 0, i);
    }
    __syncthreads();

    const int a_frag = (lane >> 5) * LA::KG + (lane & 31) + wm * WM;
    // chunk c (parity SET): A fragments from LDS stage SET, B fragments = register set SET.  Handed out between its MFMAs:
    // the B fragment loads of chunk c + 1 (into set OTHER, free since chunk c - 1 finished), the A loads of chunk c + 2
    // (into ra[SET], whose contents went to LDS during chunk c - 1), then the LDS stores of chunk c + 1's A tile.
    auto body = [&](auto SET, auto OTHER, int chunk) {
        constexpr int buf = decltype(SET)::value;
        const bool live1 = chunk + 1 < chunk_hi, live2 = chunk + 2 < chunk_hi;
        bf16x8 a[3][TM];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[pl][i] = __builtin_bit_cast(bf16x8, As[buf][pl * LA::SIZE + a_frag + i * 32]);
        constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
        constexpr int NM = 6 * TM * TN;                                   // MFMAs of the chunk: 24 or 12
        constexpr int NB = 3 * TN, NP = NB + 2 * APASS;                   // pieces: B loads, A loads, A stores
        __builtin_amdgcn_sched_barrier(0);
        int slot = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][i], __builtin_bit_cast(bf16x8, rbf[buf][PB[q]][j]),
                                                                        acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc) {
                        // piece pc goes behind MFMA (pc * NM / NP): spread evenly over the chunk (constant after unrolling)
                        if (slot == (pc * NM) / NP) {
                            if (pc < NB) load_b(OTHER, pc, live1);
                            else if (pc < NB + APASS) load_a(SET, pc - NB, live2);
                            else stage_piece(OTHER, buf ^ 1, pc - NB - APASS);
                        }
                    }
                    ++slot;
                    __builtin_amdgcn_sched_barrier(0);
                }
        __syncthreads();
    };
    for (int chunk = chunk_lo; chunk < chunk_hi; chunk += 2) {  // chunks come in pairs: one past chunk_hi is all zeros
        body(Set0{}, Set1{}, chunk);
        body(Set1{}, Set0{}, chunk + 1);
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            if (n >= p.Co) continue;
            const float bv = (bias && ks == 0) ? bias[n] : 0.f;
            // fused output epilogue (unsplit launches only): y = max((acc + bias) * scale[n] + shift[n] + residual, 0) -- the
            // frozen-statistics BatchNorm (+ skip connection)(+ ReLU) that follows the convolution in evaluation mode
            const bool affine = ep.scale != nullptr;
            const float es = affine ? ep.scale[n] : 1.f, et = affine ? ep.shift[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= M) continue;
                int64_t row = m;
                if (DIL2) {
                    int b, oy, ox;
                    decode(m, b, oy, ox);
                    row = ((int64_t)b * p.Ho + oy) * p.Wo + ox;
                    if (live >= 0) {                     // the siblings (oy ^ 1, ox), (oy, ox ^ 1), (oy ^ 1, ox ^ 1) have no tap
                        const int64_t rb = (int64_t)b * p.Ho;
                        Y[((rb + (oy ^ 1)) * p.Wo + ox) * p.Co + n] = 0.f;
                        Y[((rb + oy) * p.Wo + (ox ^ 1)) * p.Co + n] = 0.f;
                        Y[((rb + (oy ^ 1)) * p.Wo + (ox ^ 1)) * p.Co + n] = 0.f;
                    }
                }
                if (k_splits > 1) atomicAdd(Y + row * p.Co + n, acc[i][j][r] + bv);
                else if (affine) {
                    float v = fmaf(acc[i][j][r] + bv, es, et);
                    if (ep.residual) v += ep.residual[row * p.Co + n];
                    Y[row * p.Co + n] = ep.relu ? fmaxf(v, 0.f) : v;
                } else Y[row * p.Co + n] = acc[i][j][r] + bv;
            }
        }

    // BatchNorm batch statistics of this tile (stats != nullptr: unsplit, bias-free convolution followed by a BatchNorm):
    // per-channel sum and sum of squares over the tile's rows as ONE partial row [m_tile][2][Co] -- the rows bn_finalize_kernel
    // folds in double (rows past M hold exact zeros: their A rows were zero-filled).  Saves the BN reduce pass over Y.
    if (stats) {
        __shared__ float s_st[4][2][64];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float su = 0.f, sq = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { su += acc[i][j][r]; sq = fmaf(acc[i][j][r], acc[i][j][r], sq); }
            su += __shfl_xor(su, 32, 64); sq += __shfl_xor(sq, 32, 64);
            if (lane < 32) { s_st[wave][0][j * 32 + lane] = su; s_st[wave][1][j * 32 + lane] = sq; }
        }
        __syncthreads();
        constexpr int WAVES_M = (BN == 128) ? 2 : 4;                     // waves stacked along the rows of the tile
        for (int e = t; e < 2 * BN; e += 256) {
            const int which = e / BN, c = e % BN;
            const int n = n0 + c;
            if (n >= p.Co) continue;
            const int wcol = (BN == 128) ? (c >> 6) : 0, lc = c & 63;
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES_M; ++w) tot += s_st[(BN == 128) ? (w * 2 + wcol) : w][which][lc];
            // ep.stats_acc == 0: partial row m_tile (folded in a fixed order by bn_finalize_kernel);  > 0: added into row
            // (m_tile mod stats_acc) of a zeroed block of that many rows of DOUBLES with global_atomic_add_f64 (folded by the
            // BatchNorm apply kernel's own prologue: dsf_bn_forward_acc) -- ~1 MB of atomics per launch
            if (ep.stats_acc > 0)
                __hip_atomic_fetch_add(reinterpret_cast<double*>(stats) + ((int64_t)(m_tile % ep.stats_acc) * 2 + which) * p.Co + n, (double)tot,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else stats[((int64_t)m_tile * 2 + which) * p.Co + n] = tot;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// igemm_x6p_kernel: igemm_x6b_kernel for the 3 x 3, stride 1, pad 1 layers on W-wide maps, with the A operand staged as a PATCH.
//
// Why: igemm_x6b_kernel gathers, splits and stores the A tile once per (tap, chunk): every input value of a 3 x 3 layer crosses
// the loader nine times (~22 VALU + 3 ds_write per float4 each time), and on the 64-channel layers that work is amortised over
// only 24 MFMAs -- the kernel is issue-bound at 0.35-0.68 MFMA utilisation.  A tile of BMT = R W output pixels (R whole rows of
// one image) needs the (R + 2) x (W + 2) input pixels around it; this kernel loads and splits THAT once per 16-channel chunk
// ([plane][k-group][patch pixel] granules, pixel pitch W + 2) and reads all nine taps' fragments out of it: the fragment of tap
// (kh, kw) is the same ds_read_b128 moved by the constant kh (W + 2) + kw granules.  Loader work per output tile falls
// (9 BMT) / ((R + 2)(W + 2)) = 5.8x (R = 4) / 4.4x (R = 2); barriers fall 9x.  The reduction order (taps of one chunk, then the
// next chunk; the six products of a step in the same order) is igemm_x6b_kernel's, so results are bit-identical to it.
// One LDS stage = 3 planes x 2 k-groups x KGS granules (38.8 KB at R = 4): two stages as dynamic LDS (77.6 KB, 2 workgroups per
// CU); the B fragments alternate between two register sets per tap; the A loads of chunk c + 1 are issued during taps 0-2 of
// chunk c and split / stored during taps 5-8; the fragments of tap t + 1 are read during tap t.
// ------------------------------------------------------------------------------------------------
template <int BMT, int W, int NT = 9> struct X6Patch {
    // NT = 9: 3 x 3 taps, halo 2;  NT = 4: the 2 x 2 taps one output-parity class of a 4 x 4 stride-2 transposed convolution has, halo 1.
    // pixel pitch of a patch row: W + halo; 8-wide maps 24 (a 32-row fragment block spans four image rows: the 16-lane halves of
    // the ds_read_b128 then land on disjoint bank halves)
    static constexpr int HALO = (NT == 9) ? 2 : 1, TW = (NT == 9) ? 3 : 2;
    static constexpr int R = BMT / W, PW = (W == 8) ? 24 : W + HALO, NPIX = (R + HALO) * (W + HALO);
    static constexpr int KGS = (((R + HALO) * PW + 11) / 16) * 16 + 4;    // k-group pitch = 4 mod 16 granules: see LdsPlane
    static constexpr int PLANE = 2 * KGS, STAGE = 3 * PLANE;             // granules
    static constexpr int NPASS = (NPIX + 63) / 64;                       // float4 loads of the patch per thread
    static constexpr int LDS_BYTES = 2 * STAGE * 16;
};

template <int BN, int BMT, int W, int NW = 2, int BD = 2, int NT = 9, bool IP = false>
__global__ __launch_bounds__(256, 2) void igemm_x6p_kernel(const float* __restrict__ X, const uint4* __restrict__ Wimg,
                                                          const float* __restrict__ bias, float* __restrict__ Y, X6P p, int m_tiles,
                                                          int n_tiles, int k_splits, uint32_t x_bytes, uint32_t w_bytes,
                                                          float* __restrict__ stats, X6Ep ep) {
    static_assert((BN == 128 && (BMT == 128 || BMT == 64)) || (BN == 64 && BMT == 256), "tile shapes");
    static_assert(BMT % W == 0 && (W % 32 == 0 || 32 % W == 0), "fragment blocks are whole image rows or lie in one");
    static_assert(NT == 9 || (NT == 4 && BD == 2), "taps");
    static_assert(!IP || NT == 4, "input-parity classes are 2 x 2 convolutions");
    using PT = X6Patch<BMT, W, NT>;
    // waves: WMW x WN, each WM rows x WNC columns: 2 x 2 of (BMT / 2) x 64 (BN 128), 4 x 1 of 64 x 64 (BN 64), or -- NW = 4, the
    // 64-row tiles -- 1 x 4 of 64 x 32: every wave then reads its own quarter of the weight block (half the weight traffic of the
    // 2 x 2 arrangement, whose two wave rows pull the same fragments) and all of the A tile
    constexpr int WN = (BN == 128) ? NW : 1, WMW = 4 / WN, WNC = BN / WN;
    constexpr int WM = BMT / WMW, TM = WM / 32, TN = WNC / 32;
    static_assert(NW == 2 || (NW == 4 && BN == 128 && BMT == 64), "wave arrangements");
    constexpr int NPASS = PT::NPASS;
    constexpr int B_GRANULES = 2 * BN;
    extern __shared__ uint4 x6p_lds[];                   // [stage 2][plane 3][k-group 2][KGS]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    x6_stamp(0);
    const int wm = wave / WN, wn = wave % WN;
    int tile = x6_xcd_contiguous(blockIdx.x, m_tiles * n_tiles * k_splits);
    const int n_tile = tile % n_tiles; tile /= n_tiles;
    const int m_tile = tile % m_tiles; const int ks = tile / m_tiles;
    const int m0 = m_tile * BMT, n0 = n_tile * BN;
    const int M = p.B * p.Ho * p.Wo;                     // a multiple of BMT (launcher)
    const __amdgpu_buffer_rsrc_t xbuf = x6_buffer(X, x_bytes), wbuf = x6_buffer(Wimg, w_bytes);

    // patch pixel (py, px), LDS granule py PW + px  <->  input pixel (y0 + oy0 + py, ox0 + px) of image `img`.
    // NT = 9: rows are output pixels in order, (oy0, ox0) = (-pad, -pad): pad 1 (Hi x W input), or pad 0 over an input that was
    // padded beforehand (reflection padding: Hi = Ho + 2, Wi = W + 2; every patch pixel is then inside the input).  NT = 4 (dilation-2 gather of a 4 x 4 stride-2 transposed
    // convolution): rows are ordered by output parity class (cpy, cpx), then as the W-wide class image = the input grid; an output
    // pixel (2 qy + cpy, 2 qx + cpx) has the taps kh = kh0 + 2 ty, kw = kw0 + 2 tx at input (qy + oy0 + ty, qx + ox0 + tx)
    // IP (round 6; a 4 x 4, stride 2, pad 1 convolution = the input gradient of ConvTranspose2d(4, 2, 1)): rows are output pixels in
    // order; the INPUT is taken apart into its four parity classes (iy & 1, ix & 1), each a (Ho + 1) x (W + 1)-reachable image on which
    // the layer is a 2 x 2 stride-1 convolution: class (cy, cx), patch pixel (py, px) <-> input pixel (2 (y0 + py) - cy, 2 px - cx),
    // tap (ty, tx) <-> filter tap (2 ty + 1 - cy, 2 tx + 1 - cx).  The chunk loop runs over (class, channel chunk) pairs -- "virtual
    // chunks" -- and all four classes accumulate into the same output tile.
    const int OH = (NT == 4 && !IP) ? p.Hi : p.Ho;       // rows of the (class) image the tile's rows index
    const int Mc = p.B * OH * W;                         // NT = 4: rows per parity class
    const int cls = (NT == 4 && !IP) ? m0 / Mc : 0, r0 = (NT == 4 && !IP) ? m0 % Mc : m0;
    const int cpy = cls >> 1, cpx = cls & 1;
    const int kh0 = (NT == 4) ? ((p.pad_h - cpy) & 1) : 0, kw0 = (NT == 4) ? ((p.pad_w - cpx) & 1) : 0;
    const int oy0 = IP ? 0 : (NT == 4) ? (cpy - p.pad_h + kh0) / 2 : -p.pad_h, ox0 = IP ? 0 : (NT == 4) ? (cpx - p.pad_w + kw0) / 2 : -p.pad_w;
    const int img = r0 / (OH * W), y0 = (r0 % (OH * W)) / W;
    const int a_q = t & 3, a_k4 = a_q * 4, a_r = t >> 2;
    int a_off[NPASS];                                    // element offset of (pixel, channel quad) at chunk 0; -1: zeros
    int a_lds[NPASS];                                    // granule of the pixel in k-group 0; -1: no such pixel
    int a_cok[IP ? NPASS : 1];                           // IP: bit (2 cy + cx) = the pixel exists in class (cy, cx); a_off is that of class (0, 0)
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int pidx = a_r + 64 * i;
        const int py = pidx / (W + PT::HALO), px = pidx % (W + PT::HALO);
        if (IP) {
            const int y2 = 2 * (y0 + py), x2 = 2 * px;   // class (cy, cx): input pixel (y2 - cy, x2 - cx)
            const bool in = pidx < PT::NPIX;
            const int vy0 = y2 < p.Hi, vy1 = y2 >= 1 && y2 - 1 < p.Hi, vx0 = x2 < p.Wi, vx1 = x2 >= 1 && x2 - 1 < p.Wi;
            a_cok[i] = in ? ((vy0 & vx0) | (vy0 & vx1) << 1 | (vy1 & vx0) << 2 | (vy1 & vx1) << 3) : 0;
            a_off[i] = ((img * p.Hi + y2) * p.Wi + x2) * p.Ci + a_k4;
        } else {
            const int y = y0 + oy0 + py, x = ox0 + px;
            const bool ok = pidx < PT::NPIX && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
            a_off[i] = ok ? ((img * p.Hi + y) * p.Wi + x) * p.Ci + a_k4 : -1;
        }
        a_lds[i] = pidx < PT::NPIX ? py * PT::PW + px : -1;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int n_chunks = (p.Ci + XBK - 1) / XBK;       // K splits are ranges of channel chunks (all nine taps of each)
    const int v_chunks = IP ? 4 * n_chunks : n_chunks;   // IP: virtual chunk v = class * n_chunks + channel chunk
    const int per_split = (v_chunks + k_splits - 1) / k_splits;
    const int chunk_lo = ks * per_split, chunk_hi = min(v_chunks, chunk_lo + per_split);
    // IP: (class, channel chunk) of the virtual chunk the body works on and of the next one (uniform; stepped, not divided)
    int ip_cls = IP ? chunk_lo / n_chunks : 0, ip_ch = IP ? chunk_lo % n_chunks : 0, ip_cls1 = 0, ip_ch1 = 0, ip_v = chunk_lo;
    auto ip_next = [&]() { ip_ch1 = ip_ch + 1; ip_cls1 = ip_cls; if (ip_ch1 == n_chunks) { ip_ch1 = 0; ip_cls1 = ip_cls + 1; } };
    if (IP) ip_next();
    u32x4 ra[NPASS];
    u32x4 rbf[BD][3][TN];                                // [set][plane][n block]: B fragments as loaded, BD - 1 steps ahead
    bf16x8 af[2][3][TM];                                 // [set][plane][m block]: A fragments of the current / next tap
    const uint32_t b_lane = (uint32_t)((lane >> 5) * BN + wn * WNC + (lane & 31)) * 16u;
    auto load_a = [&](int i, int chunk) {                // pass i of the patch of `chunk` (past the end: zeros)
        if (IP) {                                        // `chunk` is the virtual chunk: the current one (prologue) or the next
            const bool cur = chunk == ip_v;
            const int c = cur ? ip_cls : ip_cls1, ch = cur ? ip_ch : ip_ch1;
            const bool ok = ((a_cok[i] >> c) & 1) && chunk < chunk_hi && ch * XBK + a_k4 < p.Ci;
            ra[i] = x6_load16(xbuf, ok ? (uint32_t)(a_off[i] - ((c >> 1) * p.Wi + (c & 1)) * p.Ci + ch * XBK) * 4u : X_OOB);
            return;
        }
#if defined(X6_STAMP) && defined(X6P_KO_A)                 // knock-out (diagnostic build): patch loads after the first chunk are issued out of range (no memory access)
        const bool ok = a_off[i] >= 0 && chunk <= chunk_lo && chunk * XBK + a_k4 < p.Ci;
#else
        const bool ok = a_off[i] >= 0 && chunk < chunk_hi && chunk * XBK + a_k4 < p.Ci;
#endif
        ra[i] = x6_load16(xbuf, ok ? (uint32_t)(a_off[i] + chunk * XBK) * 4u : X_OOB);
    };
    auto stage_piece = [&](int buf, int i) {
        if (NPASS * 64 > PT::NPIX && a_lds[i] < 0) return;               // the tail of the last pass
        uint2 h, m, l;
#if defined(X6_STAMP) && defined(X6P_KO_SPLIT)             // knock-out: the patch is stored unsplit
        h = make_uint2(ra[i][0], ra[i][1]); m = make_uint2(ra[i][2], ra[i][3]); l = h;
#else
        split4(ra[i], h, m, l);
#endif
        uint2* dst = reinterpret_cast<uint2*>(&x6p_lds[buf * PT::STAGE + (a_q >> 1) * PT::KGS + a_lds[i]]) + (a_q & 1);
        dst[0] = h; dst[2 * PT::PLANE] = m; dst[4 * PT::PLANE] = l;
    };
    auto load_b = [&](int S, int f, int tap, int chunk) {                // fragment f of step (chunk, tap) (past the end: zeros)
        const int pl = f / TN, j = f % TN;
#if defined(X6_STAMP) && defined(X6P_KO_B)                 // knock-out: weight-fragment loads after the first chunk are issued out of range
        const uint32_t dead = chunk <= chunk_lo ? 0u : X_OOB;
#else
        const uint32_t dead = chunk < chunk_hi ? 0u : X_OOB;
#endif
        int wtap = (NT == 4) ? (kh0 + 2 * (tap >> 1)) * p.KW + kw0 + 2 * (tap & 1) : tap;      // tap of the weight image
        int wch = chunk;
        if (IP) {                                        // `chunk`: the virtual chunk of the body, or the one after it
            const bool cur = chunk == ip_v;
            const int c = cur ? ip_cls : ip_cls1;
            wch = cur ? ip_ch : ip_ch1;
            wtap = (2 * (tap >> 1) + 1 - (c >> 1)) * p.KW + 2 * (tap & 1) + 1 - (c & 1);
        }
        const int blk = wtap * n_chunks + wch;
        rbf[S][pl][j] = x6_load16(wbuf, ((uint32_t)(blk * n_tiles + n_tile) * (uint32_t)(3 * B_GRANULES * 16) +
                                         (uint32_t)(pl * B_GRANULES * 16 + j * 32 * 16) + b_lane) | dead);
    };
    // this lane's granule of output row block i at tap (0, 0): rows wm WM + 32 i + (lane & 31) of the tile
    int a_frag[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wm * WM + i * 32 + (lane & 31);
        a_frag[i] = (lane >> 5) * PT::KGS + (ml / W) * PT::PW + (ml % W);
    }
    auto read_frag = [&](auto SET, int buf, int tap, int pl, int i) {
        constexpr int S = decltype(SET)::value;
        af[S][pl][i] = __builtin_bit_cast(bf16x8, x6p_lds[buf * PT::STAGE + pl * PT::PLANE + a_frag[i] + (tap / PT::TW) * PT::PW + tap % PT::TW]);
    };

    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) load_a(i, chunk_lo);
#pragma unroll
    for (int f = 0; f < 3 * TN; ++f) load_b(0, f, 0, chunk_lo);
    if (BD == 3) {
#pragma unroll
        for (int f = 0; f < 3 * TN; ++f) load_b(1, f, 1, chunk_lo);
    }
    // the patch loader's schedule inside a chunk: three loads per tap from tap 0, two stores per tap from tap ST
    constexpr int ST = (NT == 9) ? 5 : 2;
    static_assert(NPASS <= 2 * (NT - ST) && NPASS <= 3 * ST, "patch passes fit the tap schedule");
#pragma unroll
    for (int i = 0; i < NPASS; ++i) stage_piece(0, i);
    __syncthreads();

    // chunk c (parity PAR): A fragments from LDS stage PAR; tap t uses B register set and fragment set (PAR + t) & 1 (nine taps:
    // the next chunk starts on the other set).  Handed out between the NM = 24 or 12 MFMAs of a tap: the B loads of the next step,
    // the A fragment reads of the next tap, and this chunk's share of the patch of chunk c + 1 (loads in taps 0-2, stores in 5-8).
    auto body = [&](auto PARITY, int chunk) {
        constexpr int PAR = decltype(PARITY)::value;
        if (IP && chunk != ip_v) { ip_cls = ip_cls1; ip_ch = ip_ch1; ip_v = chunk; ip_next(); }      // (uniform: scalar registers)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (PAR && (NT & 1)) read_frag(Set1{}, PAR, 0, pl, i); else read_frag(Set0{}, PAR, 0, pl, i);
            }
        constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
        constexpr int NM = 6 * TM * TN, BSTEP = 2 * TM, ASTEP = 2 * TN;   // NM / (3 TN) B loads, NM / (3 TM) fragment reads
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            // fragment set: alternates per step (nine taps: a chunk starts on its parity; four: always on set 0);  B set: the same
            // (BD 2), tap mod 3 (BD 3: nine taps)
            const int S = ((NT & 1) ? PAR + tap : tap) & 1;
            const int SB = BD == 3 ? tap % 3 : S;
            __builtin_amdgcn_sched_barrier(0);
            int slot = 0;
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[S][PA[q]][i], __builtin_bit_cast(bf16x8, rbf[SB][PB[q]][j]),
                                                                            acc[i][j], 0, 0, 0);
                        // B fragments of the next step: behind MFMAs 0, 4, .. 20 (NM 12: 0, 2, .. 10)
                        if (slot % BSTEP == 0) {
                            const int f = slot / BSTEP, ahead = tap + BD - 1;
                            load_b(BD == 3 ? ahead % 3 : S ^ 1, f, ahead % NT, chunk + ahead / NT);
                        }
                        // A fragments of the next tap: behind MFMAs 1, 5, .. (the planes in the order the MFMAs want them)
                        if (slot % ASTEP == 1 && tap < NT - 1) {
                            const int f = slot / ASTEP, pl = f / TM, i = f % TM;
                            if (S) read_frag(Set0{}, PAR, tap + 1, pl, i); else read_frag(Set1{}, PAR, tap + 1, pl, i);
                        }
                        // the patch of chunk + 1: three loads per tap in taps 0-2, two stores per tap in taps 5-8
                        if (tap < ST && slot % (NM / 3) == 2) {
                            const int i = tap * 3 + slot / (NM / 3);
                            if (i < NPASS) load_a(i, chunk + 1);
                        }
                        if (tap >= ST && slot % (NM / 2) == NM / 4) {
                            const int i = (tap - ST) * 2 + slot / (NM / 2);
                            if (i < NPASS) stage_piece(PAR ^ 1, i);
                        }
                        ++slot;
                        __builtin_amdgcn_sched_barrier(0);
                    }
        }
        __syncthreads();
    };
    x6_stamp(1);
    for (int chunk = chunk_lo; chunk < chunk_hi; chunk += 2) {
        body(Set0{}, chunk);
        if (chunk + 1 < chunk_hi) body(Set1{}, chunk + 1);
    }
    x6_stamp(2);

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WNC + j * 32 + (lane & 31);
            if (n >= p.Co) continue;
            const float bv = (bias && ks == 0) ? bias[n] : 0.f;
            const bool affine = ep.scale != nullptr;     // the output epilogue of igemm_x6b_kernel
            const float es = affine ? ep.scale[n] : 1.f, et = affine ? ep.shift[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int64_t row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (NT == 4 && !IP) {                    // class row -> output pixel (2 qy + cpy, 2 qx + cpx)
                    const int rc = (int)row - cls * Mc, qx = rc % W, q = rc / W;
                    row = ((int64_t)(q / OH) * p.Ho + 2 * (q % OH) + cpy) * p.Wo + 2 * qx + cpx;
                }
                if (row >= M) continue;
                if (k_splits > 1) atomicAdd(Y + row * p.Co + n, acc[i][j][r] + bv);
                else if (affine) {
                    float v = fmaf(acc[i][j][r] + bv, es, et);
                    if (ep.residual) v += ep.residual[row * p.Co + n];
                    Y[row * p.Co + n] = ep.relu ? fmaxf(v, 0.f) : v;
                } else Y[row * p.Co + n] = acc[i][j][r] + bv;
            }
        }

    x6_stamp(3);
    // BatchNorm batch statistics of this tile: as in igemm_x6b_kernel (the partial sums meet in the first 2 KB of the stage)
    if (stats) {
        float (*s_st)[2][64] = reinterpret_cast<float (*)[2][64]>(x6p_lds);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float su = 0.f, sq = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { su += acc[i][j][r]; sq = fmaf(acc[i][j][r], acc[i][j][r], sq); }
            su += __shfl_xor(su, 32, 64); sq += __shfl_xor(sq, 32, 64);
            if (lane < 32) { s_st[wave][0][j * 32 + lane] = su; s_st[wave][1][j * 32 + lane] = sq; }
        }
        __syncthreads();
        for (int e = t; e < 2 * BN; e += 256) {
            const int which = e / BN, c = e % BN;
            const int n = n0 + c;
            if (n >= p.Co) continue;
            const int wcol = c / WNC, lc = c % WNC;
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < WMW; ++w) tot += s_st[w * WN + wcol][which][lc];
            if (ep.stats_acc > 0)
                __hip_atomic_fetch_add(reinterpret_cast<double*>(stats) + ((int64_t)(m_tile % ep.stats_acc) * 2 + which) * p.Co + n, (double)tot,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else stats[((int64_t)m_tile * 2 + which) * p.Co + n] = tot;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward-weights:  dW[(tap, ci)][co] += sum_{m in split} X[pixel(m) + tap][ci] * dY[m][co]
// The reduction runs over pixels, so an MFMA lane needs 8 consecutive PIXELS of one channel, while HBM (and the loader's
// float4) hold consecutive CHANNELS of one pixel.  Both tiles are therefore staged as loaded -- [pixel 16][channel 128]
// bf16 rows of 256 bytes per plane, split along the channel quad exactly as in the forward kernel -- and transposed on
// the way out of LDS by gfx950's ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered channel-major):
// two such reads make one 8-pixel fragment.  Rows are XOR-swizzled in 16-byte chunks (chunk ^ ((row & 3) << 2 | row >> 2))
// so that the loader's ds_write_b64 and the transposed reads are both bank-conflict-free.
// Both operands are activations, so both are split on the fly (2 + 2 float4 per thread and 16-pixel chunk).
// Pixel splits meet in dW by float atomics.
// ------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t x6_fast_div(uint32_t n, uint64_t magic) { return (uint32_t)(((uint64_t)n * magic) >> 40); }
// byte offset of 16-byte chunk `ch` (0..15) of pixel row `row` (0..15) inside one plane of a [16][128] bf16 tile
__device__ __forceinline__ int x6_tr_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// Loader placement (round 3): the ~120 VALU, 12 LDS stores and 4 buffer loads of a chunk are spread EVENLY over the 24 MFMA gaps by
// the compiler's group-barrier pipeline (5 VALU per gap).  Rounds 1-2 handed out whole pieces (one 22-instruction address
// decode or split behind one MFMA, nothing behind the next): a 32-cycle MFMA hides ~6 single-issue instructions of its own wave,
// a clump of 22 stalls the wave's next MFMA for ~90 cycles, six times per chunk (MI355X_MICROARCH.md, "vector-instruction ISSUE
// cost").  Measured B = 32: 488->256 at 64x64 1627 -> 1513 us (181 -> 195 TFLOP/s), 256->256 k4 s2 404 -> 366, the 8x8 .. 32x32
// maps 71-77 -> 66-72 us (profiles/r03_wrw_variants_ab.txt).  The forward-type kernels keep the piece placement: their loaders
// are lighter (2.6 VALU per gap) and the pipeline hoists their eight loads in front of the first MFMA (197 -> 186 TFLOP/s).
template <int BN, bool BIAS = false>
__global__ __launch_bounds__(256, 2) void igemm_wrw_x6_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                             float* __restrict__ dW, X6P p, int k_tiles, int n_tiles,
                                                             int n_splits, int m_per_split, uint64_t magic_wo,
                                                             uint64_t magic_ho, uint32_t x_bytes, uint32_t dy_bytes,
                                                             float* __restrict__ partial, float* __restrict__ dbias) {
    // BIAS: the workgroups of k tile 0 also add the column sums of their dY tiles into dbias (the bias gradient: every dY element
    // passes through exactly one of them per n tile) -- float atomics into a zeroed (or accumulating) vector.  A template
    // parameter, not a run-time test: a branch inside the chunk body would cut the basic block the sched_group_barrier pipeline
    // below interleaves (measured: every backward-weights launch 10 % slower, 144 -> 129 TFLOP/s on the dominant kernel)
    // partial != nullptr (deterministic mode): every pixel split stores its tile to partial[split][K][Co] (plain stores,
    // one writer per element); x6_wrw_reduce_kernel adds the splits in order.  Else: float atomics into dW.
    // BN = 128: 2 x 2 waves of 64 (k rows) x 64 (channels);  BN = 64 (layers with <= 64 output channels): 4 x 1 waves of
    // 32 x 64 -- half the MFMAs instead of multiplying zero columns; the dY tile keeps its 256-byte LDS rows, half used
    constexpr int WM = (BN == 128) ? 64 : 32;
    constexpr int TM = WM / 32, TN = 2, PLANE = 16 * 256;      // bytes per plane of a tile
    __shared__ __attribute__((aligned(16))) char As[2][3 * PLANE];
    __shared__ __attribute__((aligned(16))) char Bs[2][3 * PLANE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    x6_stamp(0);
    const int wm = (BN == 128) ? (wave >> 1) : wave, wn = (BN == 128) ? (wave & 1) : 0;
    int tile = x6_xcd_contiguous(blockIdx.x, k_tiles * n_tiles * n_splits);
    const int k_tile = tile % k_tiles; tile /= k_tiles;
    const int n_tile = tile % n_tiles; const int split = tile / n_tiles;
    const int k0 = k_tile * 128, n0 = n_tile * BN;
    const int M = p.B * p.Ho * p.Wo, K = p.KH * p.KW * p.Ci;
    const int m_begin = split * m_per_split, m_end = min(M, m_begin + m_per_split);
    const __amdgpu_buffer_rsrc_t xbuf = x6_buffer(X, x_bytes), ybuf = x6_buffer(dY, dy_bytes);

    // loader: thread -> (channel quad l_q of the tile's 128 rows, pixel (t >> 5) + 8 i of the chunk), for both tiles
    const int l_q = t & 31, l_p = t >> 5;
    const int a_k = k0 + l_q * 4;
    const bool a_kok = a_k < K;
    const int a_tap = min(a_k, K - 1) / p.Ci;
    const int a_c = min(a_k, K - 1) % p.Ci, a_kh = a_tap / p.KW, a_kw = a_tap % p.KW;
    // dY tile: BN 128: as the X tile (channel quad l_q, pixels l_p and l_p + 8: two pieces);  BN 64: the tile holds 16 x 16 float4,
    // exactly one per thread (channel quad t & 15, pixel t >> 4: ONE piece -- round 2 issued two half-masked ones and split both)
    constexpr int NPC = (BN == 128) ? 4 : 3;             // loader pieces per chunk
    const int b_q = (BN == 128) ? l_q : (t & 15), b_p = (BN == 128) ? l_p : (t >> 4);
    const int b_n = n0 + b_q * 4;
    const bool b_nok = b_n < p.Co;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 rl[3][4];                                      // [set][A px 0, A px 8, B px 0, B px 8]: chunk c lives in set c % 3
    auto load_piece = [&](auto SET, int j, int mc, auto INLOOP) __attribute__((always_inline)) {     // rows past m_end get the out-of-range offset (zeros)
        constexpr bool in_loop = decltype(INLOOP)::value; (void)in_loop;
        constexpr int S = decltype(SET)::value;
        const int m = (j < 2) ? mc + l_p + 8 * j : mc + b_p + 8 * (j - 2);
#if defined(X6_STAMP) && defined(X6_KO_ADDR)               // knock-out: neither the address arithmetic nor the load after the prologue
        if (in_loop) return;
#endif
        if (j < 2) {
            const uint32_t mm = (uint32_t)min(m, M - 1);
            const uint32_t q = x6_fast_div(mm, magic_wo);
            const int ox = (int)(mm - q * (uint32_t)p.Wo);
            const uint32_t b = x6_fast_div(q, magic_ho);
            const int oy = (int)(q - b * (uint32_t)p.Ho);
            const int iy = oy * p.stride + a_kh - p.pad_h, ix = ox * p.stride + a_kw - p.pad_w;
            // bitwise & (no short-circuit): straight-line code, one select
            const bool ok = a_kok & (m < m_end) & ((unsigned)iy < (unsigned)p.Hi) & ((unsigned)ix < (unsigned)p.Wi);
            const uint32_t off = (uint32_t)((((int)b * p.Hi + iy) * p.Wi + ix) * p.Ci + a_c) * 4u;
#if defined(X6_STAMP) && defined(X6_KO_LOAD)              // knock-out (diagnostic build): the address is computed, the load is not issued after the prologue
            if (in_loop) { asm volatile("" ::"v"(ok ? off : X_OOB)); return; }
#endif
            rl[S][j] = x6_load16(xbuf, ok ? off : X_OOB);
        } else {
            const bool ok = b_nok & (m < m_end);
#if defined(X6_STAMP) && defined(X6_KO_LOAD)
            if (in_loop) { asm volatile("" ::"v"(ok ? (uint32_t)(m * p.Co + b_n) * 4u : X_OOB)); return; }
#endif
            rl[S][j] = x6_load16(ybuf, ok ? (uint32_t)(m * p.Co + b_n) * 4u : X_OOB);
        }
    };
    const int st_off0 = x6_tr_off(l_p, l_q >> 1) + 8 * (l_q & 1), st_off1 = x6_tr_off(l_p + 8, l_q >> 1) + 8 * (l_q & 1);
    const int st_offb0 = x6_tr_off(b_p, b_q >> 1) + 8 * (b_q & 1), st_offb1 = x6_tr_off((b_p + 8) & 15, b_q >> 1) + 8 * (b_q & 1);
    const float bias_on = (BIAS && k_tile == 0) ? 1.f : 0.f;            // (a factor, not a branch: see above)
    float bsum0 = 0.f, bsum1 = 0.f, bsum2 = 0.f, bsum3 = 0.f;
    auto stage_piece = [&](auto SET, int buf, int j) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        uint2 h, m, l;
#if defined(X6_STAMP) && defined(X6_KO_SPLIT)              // knock-out: the tile is stored unsplit (wrong values, same LDS traffic)
        h = make_uint2(rl[S][j][0], rl[S][j][1]); m = make_uint2(rl[S][j][2], rl[S][j][3]); l = h;
#else
        split4(rl[S][j], h, m, l);
#endif
        if (BIAS && j >= 2) {                            // (compile-time; rows past m_end were loaded as zeros; scalar FMAs: no packed FP32)
            const f32x4 v = __builtin_bit_cast(f32x4, rl[S][j]);
            bsum0 = fmaf(v[0], bias_on, bsum0); bsum1 = fmaf(v[1], bias_on, bsum1);
            bsum2 = fmaf(v[2], bias_on, bsum2); bsum3 = fmaf(v[3], bias_on, bsum3);
        }
        char* base = (j < 2) ? As[buf] + (j ? st_off1 : st_off0) : Bs[buf] + ((j - 2) ? st_offb1 : st_offb0);
        *reinterpret_cast<uint2*>(base) = h;
        *reinterpret_cast<uint2*>(base + PLANE) = m;
        *reinterpret_cast<uint2*>(base + 2 * PLANE) = l;
    };

    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    using Set2 = std::integral_constant<int, 2>;
    if (m_begin < m_end) {
#pragma unroll
        for (int j = 0; j < NPC; ++j) load_piece(Set0{}, j, m_begin, std::false_type{});
#pragma unroll
        for (int j = 0; j < NPC; ++j) load_piece(Set1{}, j, m_begin + XBK, std::false_type{});
#pragma unroll
        for (int j = 0; j < NPC; ++j) load_piece(Set2{}, j, m_begin + 2 * XBK, std::false_type{});
#pragma unroll
        for (int j = 0; j < NPC; ++j) stage_piece(Set0{}, 0, j);
    }
    __syncthreads();

    // transposed fragment reads: 16-lane group g = channels 16 g .. 16 g + 15 of the 32-channel MFMA tile; lane 4 q + c of
    // the group addresses pixel row q, channels 4 c .. 4 c + 3 of the block and receives channel (lane & 15) of 4 pixels
    const int f_row = 8 * (lane >> 5) + ((lane & 15) >> 2);                    // + 4 for the second read
    const int f_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);                 // first channel addressed, within the MFMA tile
    auto frag_off = [&](int cbase, int r) __attribute__((always_inline)) {                                     // cbase: first channel of the MFMA tile
        const int c = cbase + f_col;
        return x6_tr_off(f_row + 4 * r, c >> 3) + 2 * (c & 7);
    };
    int fa[TM][2], fb[TN][2];
#pragma unroll
    for (int i = 0; i < TM; ++i) { fa[i][0] = frag_off(wm * WM + i * 32, 0); fa[i][1] = frag_off(wm * WM + i * 32, 1); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { fb[j][0] = frag_off(wn * 64 + j * 32, 0); fb[j][1] = frag_off(wn * 64 + j * 32, 1); }
    auto tr_read = [&](const char* base, int off0, int off1) __attribute__((always_inline)) {
        struct { s16x4 lo, hi; } v;
        v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off0));
        v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off1));
        return __builtin_bit_cast(bf16x8, v);
    };

    // chunk c: fragments from LDS stage c % 2; loads chunk c + 3 into register set c % 3 (free since chunk c was staged during chunk
    // c - 1); splits and stores chunk c + 1 (register set (c + 1) % 3, loaded during chunk c - 2) into the other stage.  Round 6: the
    // loads used to run TWO chunks ahead of their split (two register sets) -- ~1.2 chunk times = 1.1-1.4 us at these kernels' pace,
    // less than a load that misses L2 needs under load: with the loop's buffer loads knocked out (tools/x6/wrw_stamps.py --variant
    // ko_load, profiles/r06_wrw_knockouts.txt) the main loop of the small-map layers ran 27 % shorter.
    auto body = [&](auto SET, auto OTHER, auto BUF, int mc) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value;
        bf16x8 a[3][TM], b[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[pl][i] = tr_read(As[buf] + pl * PLANE, fa[i][0], fa[i][1]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[pl][j] = tr_read(Bs[buf] + pl * PLANE, fb[j][0], fb[j][1]);
        }
        constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
        {
            // program order: fragment reads, the whole loader, the MFMAs; the pipeline below interleaves them
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) load_piece(SET, pc, mc + 3 * XBK, std::true_type{});
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) stage_piece(OTHER, buf ^ 1, pc);
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][i], b[PB[q]][j], acc[i][j], 0, 0, 0);
            constexpr int NM = 6 * TM * TN;                              // 24 or 12 gaps
            constexpr int VPG = (TM == 2) ? 5 : 8;                       // VALU per gap
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);           // plane-0 fragments first
#pragma unroll
            for (int g = 0; g < NM; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (g < 2 * (TM + TN)) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // the other planes' fragments, two per gap
                __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0);
                // the four buffer loads early (behind their address arithmetic, the first VALU in program order): they have the
                // rest of this chunk and most of the next to land before the split arithmetic, which comes last, needs them
                if (g == NM / 6 || g == NM / 3 || g == NM / 3 + 1 || g == NM / 3 + 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    x6_stamp(1);
    // three register sets x two LDS stages: the pattern repeats after six chunks.  The main loop is straight-line code (the compiler's
    // s_waitcnt vmcnt(N) are then exact: with a condition in front of every chunk it assumed the fewest loads in flight at each join
    // and waited for ALL of them at the top of two chunks in six); the last 0-5 chunks of a split run from a nest of conditions
    int mc = m_begin;
    for (; mc + 6 * XBK <= m_end; mc += 6 * XBK) {
        body(Set0{}, Set1{}, Set0{}, mc);
        body(Set1{}, Set2{}, Set1{}, mc + XBK);
        body(Set2{}, Set0{}, Set0{}, mc + 2 * XBK);
        body(Set0{}, Set1{}, Set1{}, mc + 3 * XBK);
        body(Set1{}, Set2{}, Set0{}, mc + 4 * XBK);
        body(Set2{}, Set0{}, Set1{}, mc + 5 * XBK);
    }
    if (mc < m_end) {
        body(Set0{}, Set1{}, Set0{}, mc);
        if (mc + XBK < m_end) {
            body(Set1{}, Set2{}, Set1{}, mc + XBK);
            if (mc + 2 * XBK < m_end) {
                body(Set2{}, Set0{}, Set0{}, mc + 2 * XBK);
                if (mc + 3 * XBK < m_end) {
                    body(Set0{}, Set1{}, Set1{}, mc + 3 * XBK);
                    if (mc + 4 * XBK < m_end) body(Set1{}, Set2{}, Set0{}, mc + 4 * XBK);
                }
            }
        }
    }
    x6_stamp(2);

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            if (n >= p.Co) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (k >= K) continue;
                if (partial) partial[((int64_t)split * K + k) * p.Co + n] = acc[i][j][r];
                else atomicAdd(dW + (int64_t)k * p.Co + n, acc[i][j][r]);
            }
        }
    x6_stamp(3);
    if (BIAS && k_tile == 0) {                           // (workgroup-uniform) fold the pixel lanes through LDS, one atomic per channel
        float* s_red = reinterpret_cast<float*>(As[0]);  // [pixel lane][BN]: the last chunk's barrier has passed, the tiles are dead
        constexpr int LANES = (BN == 128) ? 8 : 16;
        __syncthreads();
        s_red[b_p * BN + b_q * 4 + 0] = bsum0; s_red[b_p * BN + b_q * 4 + 1] = bsum1;
        s_red[b_p * BN + b_q * 4 + 2] = bsum2; s_red[b_p * BN + b_q * 4 + 3] = bsum3;
        __syncthreads();
        if (t < BN && n0 + t < p.Co) {
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < LANES; ++q) tot += s_red[q * BN + t];
            atomicAdd(dbias + n0 + t, tot);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// igemm_wrw_x6p_kernel: backward-weights of the 3 x 3, stride 1, pad 1 layers with the input staged as ROWS WITH HALO.
//
// igemm_wrw_x6_kernel gathers, splits and stores a 16-pixel x 128-(tap, channel) X tile and a 16 x BN dY tile per 24 MFMAs of a
// wave: ~120 loader VALU per chunk, 5 per MFMA gap -- the kernel is issue-bound at 0.47 MFMA utilisation.  Here a workgroup owns
// 32 input channels x ALL NINE TAPS x 128 output channels (wave w: output channels 32 w .. 32 w + 31, nine 32 x 32 accumulators)
// and walks whole output rows of the maps: the three input rows an output row needs live in a ring of four row buffers
// ([plane][pixel -1 .. W][32 channels] bf16, 64 bytes per pixel and plane), each input row is loaded and split ONCE and read by
// three output rows x three kw shifts (the fragment of tap (kh, kw) is the transposed read of ring row kh moved by kw pixels = 64 kw
// bytes; 64-byte pixels make any 4-pixel window of the ds_read_b64_tr_b16 bank-conflict-free without a swizzle), and a dY tile
// feeds 54 MFMAs per wave instead of 24.  Loader work per MFMA falls ~4x (1.3 VALU per gap).  Rows outside the image (kh = 0
// above the first row, kh = 2 below the last) read a plane of zeros instead of the ring; the halo pixels are zeroed once.
// Pixel splits are ranges of whole rows; they meet in dW by float atomics, or in `partial` in deterministic mode, as in
// igemm_wrw_x6_kernel.
// ------------------------------------------------------------------------------------------------
template <int W> struct X6WrwPatch {
    static constexpr int A_PLANE = (W + 2) * 64, A_SLOT = 3 * A_PLANE;   // bytes: one plane of a row buffer, one row buffer
    static constexpr int B_PLANE = 16 * 256;                             // dY tile plane: as igemm_wrw_x6_kernel
    static constexpr int RING = 0, ZERO = 4 * A_SLOT, BS = ZERO + A_PLANE, LDS_BYTES = BS + 2 * 3 * B_PLANE;
    static constexpr int CPR = W / 16;                                   // 16-pixel chunks per row
    static constexpr int NXP = (W * 8 + 255) / 256;                      // float4 loads of an input row per thread
};

template <int W, int BN = 128>
__global__ __launch_bounds__(256, 2) void igemm_wrw_x6p_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                              float* __restrict__ dW, X6P p, int c_tiles, int n_tiles, int n_splits,
                                                              int rows_per_split, uint32_t x_bytes, uint32_t dy_bytes,
                                                              float* __restrict__ partial) {
    using PT = X6WrwPatch<W>;
    constexpr int CPR = PT::CPR, NXP = PT::NXP, U = CPR > 2 ? CPR : 2;   // chunks per trip of the main loop
    // BN 128: wave w = output channels 32 w .., all nine taps.  BN 64 (layers with <= 64 output channels): two channel blocks x two
    // tap groups (taps 0-4 and 5-8; the second group's fifth accumulator multiplies the zero plane: idle either way)
    constexpr int WN = BN / 32, TAPS = (BN == 128) ? 9 : 5, NPY = (BN == 128) ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) char wrw_lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int tile = x6_xcd_contiguous(blockIdx.x, c_tiles * n_tiles * n_splits);
    const int cb = tile % c_tiles; tile /= c_tiles;
    const int n_tile = tile % n_tiles; const int split = tile / n_tiles;
    const int c0 = cb * 32, n0 = n_tile * BN;
    const int wn = wave % WN, t0 = __builtin_amdgcn_readfirstlane(wave / WN) * TAPS;          // first tap of this wave
    const int RT = p.B * p.Hi, K = 9 * p.Ci;             // rows of all maps; the stride-1 layer has Ho = Hi, Wo = Wi = W
    const int r_begin = split * rows_per_split, r_end = min(RT, r_begin + rows_per_split);
    const __amdgpu_buffer_rsrc_t xbuf = x6_buffer(X, x_bytes), ybuf = x6_buffer(dY, dy_bytes);

    // zero plane and halo pixels (pixel -1 and W of every plane of every ring slot)
    for (int e = t; e < PT::A_PLANE / 16; e += 256) reinterpret_cast<uint4*>(wrw_lds + PT::ZERO)[e] = make_uint4(0, 0, 0, 0);
    for (int e = t; e < 4 * 3 * 2 * 4; e += 256) {
        const int q = e & 3, side = (e >> 2) & 1, pl = (e >> 3) % 3, slot = e / 24;
        reinterpret_cast<uint4*>(wrw_lds + slot * PT::A_SLOT + pl * PT::A_PLANE + side * (W + 1) * 64)[q] = make_uint4(0, 0, 0, 0);
    }

    // X loader: thread -> (pixel x_px, channel quad x_q) of an input row, NXP passes of 32 pixels
    const int x_q = t & 7, x_c = c0 + x_q * 4;
    const bool x_cok = x_c < p.Ci;
    // dY loader: thread -> (channel quad b_q, pixels b_p and b_p + 8 of the chunk);  BN 64: one float4 per thread (pixel t >> 4)
    const int b_q = (BN == 128) ? (t & 31) : (t & 15), b_p = (BN == 128) ? (t >> 5) : (t >> 4), b_n = n0 + b_q * 4;
    const bool b_nok = b_n < p.Co;

    f32x16 acc[TAPS];
#pragma unroll
    for (int i = 0; i < TAPS; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    u32x4 rx[NXP], rl[2][NPY];
    auto load_x = [&](int pass, int g) {                 // input row g (all maps stacked); outside [0, RT): zeros
        const int px = (t >> 3) + 32 * pass;
        const bool ok = x_cok & (px < W) & ((unsigned)g < (unsigned)RT);
        rx[pass] = x6_load16(xbuf, ok ? (uint32_t)((g * W + px) * p.Ci + x_c) * 4u : X_OOB);
    };
    auto stage_x = [&](int pass, int g) {
        const int px = (t >> 3) + 32 * pass;
        if (NXP * 32 > W && px >= W) return;
        uint2 h, m, l;
        split4(rx[pass], h, m, l);
        char* base = wrw_lds + (g & 3) * PT::A_SLOT + (px + 1) * 64 + x_q * 8;
        *reinterpret_cast<uint2*>(base) = h;
        *reinterpret_cast<uint2*>(base + PT::A_PLANE) = m;
        *reinterpret_cast<uint2*>(base + 2 * PT::A_PLANE) = l;
    };
    auto load_y = [&](auto SET, int j, int q) {          // piece j of chunk q of the split (past the end: zeros)
        constexpr int S = decltype(SET)::value;
        const int r = r_begin + q / CPR, x0 = (q % CPR) * 16;
        const int m = r * W + x0 + b_p + 8 * j;
        const bool ok = b_nok & (r < r_end);
        rl[S][j] = x6_load16(ybuf, ok ? (uint32_t)(m * p.Co + b_n) * 4u : X_OOB);
    };
    const int st_offb0 = x6_tr_off(b_p, b_q >> 1) + 8 * (b_q & 1), st_offb1 = x6_tr_off(b_p + 8, b_q >> 1) + 8 * (b_q & 1);
    auto stage_y = [&](auto SET, int buf, int j) {
        constexpr int S = decltype(SET)::value;
        uint2 h, m, l;
        split4(rl[S][j], h, m, l);
        char* base = wrw_lds + PT::BS + buf * 3 * PT::B_PLANE + (j ? st_offb1 : st_offb0);
        *reinterpret_cast<uint2*>(base) = h;
        *reinterpret_cast<uint2*>(base + PT::B_PLANE) = m;
        *reinterpret_cast<uint2*>(base + 2 * PT::B_PLANE) = l;
    };

    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    // prologue: input rows r_begin - 1 .. r_begin + 1 into the ring, dY chunk 0 staged, chunk 1 in flight
    if (r_begin < r_end) {
#pragma unroll
        for (int d = -1; d <= 1; ++d) {
#pragma unroll
            for (int ps = 0; ps < NXP; ++ps) load_x(ps, r_begin + d);
#pragma unroll
            for (int ps = 0; ps < NXP; ++ps) stage_x(ps, r_begin + d);
        }
#pragma unroll
        for (int j = 0; j < NPY; ++j) load_y(Set0{}, j, 0);
#pragma unroll
        for (int j = 0; j < NPY; ++j) load_y(Set1{}, j, 1);
#pragma unroll
        for (int j = 0; j < NPY; ++j) stage_y(Set0{}, 0, j);
    }
    __syncthreads();

    // transposed fragment reads (see igemm_wrw_x6_kernel): lane -> pixel row f_row (+ 4 for the second read), channels f_col ..
    const int f_row = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int f_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int la = f_row * 64 + f_col * 2;               // X rows: 64-byte pixels, no swizzle
    int fb[2];
    {
        const int c = wn * 32 + f_col;
        fb[0] = x6_tr_off(f_row, c >> 3) + 2 * (c & 7);
        fb[1] = x6_tr_off(f_row + 4, c >> 3) + 2 * (c & 7);
    }
    auto tr_read = [&](const char* a0, const char* a1) {
        struct { s16x4 lo, hi; } v;
        v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
        v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
        return __builtin_bit_cast(bf16x8, v);
    };

    // chunk q (dY register set / LDS stage SET = q & 1; CI = its place in the row): nine taps x 6 MFMAs.  Handed out between them:
    // the fragment reads of the next tap, the dY loads of chunk q + 2 and the LDS stores of chunk q + 1, and this chunk's share
    // of input row r + 2 (loads in the first chunks of row r, stores in the last ones; W = 16: load first, store last).
    auto body = [&](auto SET, auto OTHER, auto CI_, int q) {
        constexpr int buf = decltype(SET)::value, CI = decltype(CI_)::value;
        const int r = r_begin + q / CPR, x0 = (q % CPR) * 16;
        const int oy = r % p.Hi;
        // fragment base of plane pl of tap t0 + tt: the ring slot of input row r + kh - 1 moved by kw pixels, or the zero plane
        // outside the map (and for the tap past the ninth)
        // (BN 128: nine taps = three ring rows x three constant kw shifts -- nine base registers, not 27)
        constexpr int NBASE = (BN == 128) ? 3 : TAPS;
        int abase[NBASE][3];
#pragma unroll
        for (int tt = 0; tt < NBASE; ++tt) {
            const int tap = (BN == 128) ? 3 * tt : t0 + tt, kh = tap / 3, kw = tap - 3 * kh;
            const bool inside = tap < 9 && (kh == 1 || (kh == 0 ? oy > 0 : oy < p.Hi - 1));
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                abase[tt][pl] = (inside ? ((r + kh - 1) & 3) * PT::A_SLOT + pl * PT::A_PLANE + kw * 64 : PT::ZERO) + x0 * 64 + la;
        }
        auto frag = [&](int tt, int pl) {                // byte offset of the fragment of this wave's tap tt
            return (BN == 128) ? abase[tt / 3][pl] + (tt % 3) * 64 : abase[tt][pl];
        };
        const char* bbase = wrw_lds + PT::BS + buf * 3 * PT::B_PLANE;
        bf16x8 b[3], a[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) b[pl] = tr_read(bbase + pl * PT::B_PLANE + fb[0], bbase + pl * PT::B_PLANE + fb[1]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[0][pl] = tr_read(wrw_lds + frag(0, pl), wrw_lds + frag(0, pl) + 256);
        constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int S = tap & 1;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[S][PA[s]], b[PB[s]], acc[tap], 0, 0, 0);
                // fragments of the next tap: plane 0 first (the first MFMAs of a tap want it)
                if (tap < TAPS - 1 && s < 3) a[S ^ 1][s] = tr_read(wrw_lds + frag(tap + 1, s), wrw_lds + frag(tap + 1, s) + 256);
                if (s == 4) {
                    // (nine taps: pieces behind taps 0, 1, 2, 3, 5, 7 / 8;  five: 0, 1, 2, 3, 4)
                    if (tap == 0) load_y(SET, 0, q + 2);
                    if (tap == 1 && NPY == 2) load_y(SET, 1, q + 2);
                    if (tap == 3) stage_y(OTHER, buf ^ 1, 0);
                    if (tap == (TAPS == 9 ? 5 : 4) && NPY == 2) stage_y(OTHER, buf ^ 1, 1);
                    constexpr int TX = TAPS == 9 ? 7 : 4;                // the tap behind which an X row piece is stored
                    if (CPR == 4) {
                        if (tap == 2 && CI < 2) load_x(CI, r + 2);
                        if (tap == TX && CI >= 2) stage_x(CI - 2, r + 2);
                    } else if (CPR == 2) {
                        if (tap == 2 && CI == 0) load_x(0, r + 2);
                        if (tap == TX && CI == 1) stage_x(0, r + 2);
                    } else {
                        if (tap == 0) load_x(0, r + 2);
                        if (tap == TAPS - 1) stage_x(0, r + 2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    };
    const int n_chunks = (r_end - r_begin) * CPR;
    for (int q = 0; q < n_chunks; q += U) {              // (W = 16: rows come in pairs; one past the end has an all-zero dY tile)
        if (CPR == 4) {
            body(Set0{}, Set1{}, std::integral_constant<int, 0>{}, q);
            body(Set1{}, Set0{}, std::integral_constant<int, 1>{}, q + 1);
            body(Set0{}, Set1{}, std::integral_constant<int, 2>{}, q + 2);
            body(Set1{}, Set0{}, std::integral_constant<int, 3>{}, q + 3);
        } else if (CPR == 2) {
            body(Set0{}, Set1{}, std::integral_constant<int, 0>{}, q);
            body(Set1{}, Set0{}, std::integral_constant<int, 1>{}, q + 1);
        } else {
            body(Set0{}, Set1{}, std::integral_constant<int, 0>{}, q);
            body(Set1{}, Set0{}, std::integral_constant<int, 0>{}, q + 1);
        }
    }

    const int n = n0 + wn * 32 + (lane & 31);
    if (n < p.Co) {
#pragma unroll
        for (int tt = 0; tt < TAPS; ++tt) {
            if (t0 + tt >= 9) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = c0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (ci >= p.Ci) continue;
                const int k = (t0 + tt) * p.Ci + ci;
                if (partial) partial[((int64_t)split * K + k) * p.Co + n] = acc[tt][r];
                else atomicAdd(dW + (int64_t)k * p.Co + n, acc[tt][r]);
            }
        }
    }
}

// dW[e] = sum over the pixel splits, ascending (deterministic mode)
// (n is a multiple of 4: K * Co with Co % 4 == 0; four elements per lane, the splits still added one by one in ascending order)
// accumulate != 0: the ordered sum starts from what dW holds (the launcher's `accumulate` contract; still one fixed order)
__global__ __launch_bounds__(256) void x6_wrw_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, int64_t n,
                                                           int splits, int accumulate) {
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= n) return;
    float4 s = accumulate ? *reinterpret_cast<const float4*>(dW + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    int q = 0;
    for (; q + 4 <= splits; q += 4) {                    // four independent loads in flight, added in order
        const float4 a = *reinterpret_cast<const float4*>(partial + (int64_t)q * n + e);
        const float4 b = *reinterpret_cast<const float4*>(partial + (int64_t)(q + 1) * n + e);
        const float4 c = *reinterpret_cast<const float4*>(partial + (int64_t)(q + 2) * n + e);
        const float4 d = *reinterpret_cast<const float4*>(partial + (int64_t)(q + 3) * n + e);
        s.x = (((s.x + a.x) + b.x) + c.x) + d.x; s.y = (((s.y + a.y) + b.y) + c.y) + d.y;
        s.z = (((s.z + a.z) + b.z) + c.z) + d.z; s.w = (((s.w + a.w) + b.w) + c.w) + d.w;
    }
    for (; q < splits; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(partial + (int64_t)q * n + e);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    *reinterpret_cast<float4*>(dW + e) = s;
}

// Measurement aid (bench.py): a bare v_mfma_f32_32x32x16_bf16 loop on pseudo-random operands, four accumulators per wave,
// no memory traffic -- the matrix-pipe rate this chip sustains at the clock it holds under that load, i.e. the practical
// ceiling of the split-operand kernels (their nominal peak, 2500 / 6 TFLOP/s, assumes the 2.4 GHz boost clock).
__global__ __launch_bounds__(256) void mfma_bf16_probe_kernel(const u32x4* __restrict__ in, float* __restrict__ out, int iters) {
    u32x4 ra[6], rb[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { ra[i] = in[(threadIdx.x + 64 * i) & 1023]; rb[i] = in[(threadIdx.x * 3 + 17 * i) & 1023]; }
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[(q + t) % 6]),
                                                                 __builtin_bit_cast(bf16x8, rb[(q * 2 + t) % 6]), acc[t], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace

// More than 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize, and the attribute is set for the device that is
// current at the call: once per (kernel instantiation, device), a failure is not remembered (the next launch tries again).
constexpr int X6_MAX_DEVICES = 16;
static inline bool x6_arm_dynamic_lds(const void* fn, int bytes, bool* armed) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (dev >= 0 && dev < X6_MAX_DEVICES && armed[dev]) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (dev >= 0 && dev < X6_MAX_DEVICES) armed[dev] = true;
    return true;
}

extern "C" {

#ifdef X6_STAMP
int dsf_x6_stamp_buffer(unsigned long long* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(x6_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : 1; }
#endif

// operands: 1024 x 16 bytes of bf16 pairs; out: workgroups * 256 floats (ignored values); returns the MFMA count issued
int64_t dsf_mfma_bf16_probe(const void* operands, float* out, int workgroups, int iters, dsf_stream_t stream) {
    if (!operands || !out || workgroups <= 0 || iters <= 0) return -1;
    hipLaunchKernelGGL(mfma_bf16_probe_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (const u32x4*)operands, out,
                       iters);
    if (dsf_launch_status() != DSF_OK) return -1;
    return (int64_t)workgroups * 4 * iters * 24;
}

int64_t dsf_conv_x6_image_bytes(int KH, int KW, int Ck, int Cn) {
    const int bn = x6_bn(Cn);
    const int64_t chunks = (Ck + XBK - 1) / XBK, n_tiles = (Cn + bn - 1) / bn;
    return (int64_t)KH * KW * chunks * n_tiles * (3 * 2 * bn * 16);
}

int dsf_conv_x6_split_weights(const float* W, void* image, int KH, int KW, int Ci, int Co, int mode, dsf_stream_t stream) {
    DSF_CHECK_ARG(W && image && KH > 0 && KW > 0 && Ci > 0 && Co > 0 && (mode == 0 || mode == 1));
    const int Ck = mode ? Co : Ci, Cn = mode ? Ci : Co;
    const int bn = x6_bn(Cn);
    const int chunks = (Ck + XBK - 1) / XBK, n_tiles = (Cn + bn - 1) / bn;
    const int64_t granules = (int64_t)KH * KW * chunks * n_tiles * 2 * bn;
    hipLaunchKernelGGL(x6_split_weights_kernel, dim3((unsigned)((granules + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                       (uint4*)image, KH, KW, Ci, Co, mode, chunks, n_tiles, bn, granules);
    return dsf_launch_status();
}

int64_t dsf_conv_x6_image_granules(int KH, int KW, int Ck, int Cn) { return dsf_conv_x6_image_bytes(KH, KW, Ck, Cn) / 48; }

int dsf_conv_x6_split_weights_multi(const int64_t* jobs, int n_jobs, int64_t total_granules, dsf_stream_t stream) {
    DSF_CHECK_ARG(jobs && n_jobs >= 0 && total_granules >= 0);
    if (n_jobs == 0 || total_granules == 0) return DSF_OK;
    hipLaunchKernelGGL(x6_split_weights_multi_kernel, dim3((unsigned)((total_granules + 1023) / 1024)), dim3(256), 0,
                       (hipStream_t)stream, jobs, n_jobs);
    return dsf_launch_status();
}

// Tiling of one forward launch: tile rows, tile counts and the number of K splits (k_splits < 1: chosen here).
struct X6Plan { int bn, n_tiles, bdirect, bmt, m_tiles, k_splits; };
// the geometry igemm_x6p_kernel serves: 3 x 3, stride 1, pad 1 on 64 / 32 / 16 / 8-wide maps (DSF_X6_PATCH=0 switches the kernel
// off, 1 keeps it to the 64-wide maps; read per call: tests/test_gpu_conv.py compares the kernels in one process)
// -> taps of the patch kernel: 9 (that geometry), 4 (a 4 x 4 stride-2 transposed convolution as its dilation-2 gather: each output
// parity class is a 2 x 2 convolution over the input grid; 32 / 16 / 8-wide inputs; DSF_X6_PATCH=3 keeps these on the gather kernel), 0
static int x6_patch_geometry(int Hi, int Wi, int Ho, int Wo, int KH, int KW, int stride, int dil, int pad_h, int pad_w) {
    const char* e = getenv("DSF_X6_PATCH");
    int level = e ? atoi(e) : 2;
    if (level <= 0) return 0;
    const bool no_ip = level == 4;                       // 4: everything of level 2 but the input-parity launch
    if (no_ip) level = 2;
    if (dil == 2 && KH == 4 && KW == 4 && stride == 1 && Ho == 2 * Hi && Wo == 2 * Wi && pad_h >= 0 && pad_h <= 3 && pad_w >= 0 &&
        pad_w <= 3 && level == 2 && (Wi == 32 || Wi == 16 || Wi == 8)) return 4;
    // 5: a 4 x 4, stride 2, pad 1 convolution (the input gradient of that transposed convolution) by input parity classes (round 6;
    // DSF_X6_PATCH=4 keeps it on the gather kernel)
    if (dil == 1 && KH == 4 && KW == 4 && stride == 2 && pad_h == 1 && pad_w == 1 && Hi == 2 * Ho && Wi == 2 * Wo && level == 2 && !no_ip &&
        (Wo == 32 || Wo == 16 || Wo == 8)) return 5;
    // 3 x 3, stride 1: pad 1, or pad 0 over an input that carries its own (reflection) padding
    if (!(dil == 1 && KH == 3 && KW == 3 && stride == 1 && pad_h == pad_w && (pad_h == 0 || pad_h == 1) && Hi == Ho + 2 - 2 * pad_h &&
          Wi == Wo + 2 - 2 * pad_w)) return 0;
    return (Wo == 64 || (level >= 2 && (Wo == 32 || Wo == 16 || Wo == 8))) ? 9 : 0;
}
// patch_w: the map width when the layer has igemm_x6p_kernel's geometry (x6_patch_geometry), else 0
static X6Plan x6_forward_plan(int64_t M, int Ci, int Co, int KH, int KW, int dil, int k_splits, int patch_w = 0) {
    const int bn = x6_bn(Co);
    const int n_tiles = (Co + bn - 1) / bn;
    // DSF_X6_BDIRECT=0: the first-generation kernels (both operands through LDS); default: igemm_x6b_kernel (B operand straight
    // from the L2-resident weight image into the MFMA fragment registers)
    static const int bdirect = [] { const char* e = getenv("DSF_X6_BDIRECT"); return e ? atoi(e) : 1; }();
    // 64-row tiles when 128-row ones would leave most of the chip idle or force split-K (8x8 .. 32x32 maps)
    static const int bm_env = [] { const char* e = getenv("DSF_X6_BM"); return e ? atoi(e) : 0; }();             // tuning aid
    int bmt = (bn == 128 && (bm_env == 64 || (bm_env != 128 && ((M + 127) / 128) * n_tiles < 384))) ? 64 : 128;
    if (bdirect && bn == 64) bmt = 256;                                  // 4 x 1 waves of 64 x 64
    if (bdirect && bn == 128 && patch_w == 8) bmt = 64;                  // the patch kernel's tiles lie inside one (8 x 8) image
    const int m_tiles = (int)((M + bmt - 1) / bmt);
    const int n_chunks = (KH * KW / (dil * dil)) * ((Ci + XBK - 1) / XBK);          // live chunks of a tile
    const char* ks_e = getenv("DSF_X6_KSPLIT_WGS");                      // tuning aid, read per call: workgroup target of the K splits
    const int ks_target = (ks_e && atoi(ks_e) > 0) ? atoi(ks_e) : 512;
    if (k_splits < 1 && patch_w && bdirect && bmt == 64) {
        // the patch kernel's 64-row tiles: one workgroup per CU already runs at the rate of two half-length ones (B = 32, 16x16x256:
        // 256 tiles unsplit 48 us, 2-way 54) and an unsplit launch needs no zero fill, keeps the BatchNorm-statistics epilogue and is
        // deterministic; 128 tiles: unsplit 67 us, 4-way 54.  Splits are ranges of channel chunks, at least two each.
        const int tiles = m_tiles * n_tiles, ch = (Ci + XBK - 1) / XBK;
        k_splits = tiles < 256 ? (ks_target + tiles - 1) / tiles : 1;
        if (k_splits > ch / 2) k_splits = ch / 2;
        if (k_splits < 1) k_splits = 1;
    }
    if (k_splits < 1) {
        // auto: fewer than ~0.8 tiles per CU -> split K towards 2 workgroups per CU.  Each split adds M x Co float atomics
        // (~1.3 TB/s chip-wide), so 256 tiles run unsplit (69 vs 83 us on the 32x32x128 layers), 128 tiles 4-way, 64 8-way.
        // 64-row tiles (measured, B = 32): 512 tiles unsplit 63 us (128-row: 68), 256 tiles 2-way 70 (75), 128 tiles 4-way 68 (74).
        const int tiles = m_tiles * n_tiles;
        k_splits = bmt == 64 ? (tiles < 512 ? (ks_target + tiles - 1) / tiles : 1) : (tiles < 200 ? (ks_target + tiles / 2) / tiles : 1);
        if (k_splits > n_chunks / 8) k_splits = n_chunks / 8;
        if (k_splits < 1) k_splits = 1;
    }
    if (k_splits > n_chunks) k_splits = n_chunks > 0 ? n_chunks : 1;
    if (dsf_deterministic()) k_splits = 1;                               // no float atomics in the epilogue
    return X6Plan{bn, n_tiles, bdirect, bmt, m_tiles, k_splits};
}

// igemm_x6p_kernel's conditions on top of x6_patch_geometry: tiles of whole image rows inside one image, K splits no finer than
// channel chunks
// (ph x pw: the image the tile rows index -- the output map, or the input grid = one parity class of a transposed convolution)
static bool x6_patch_applies(const X6Plan& plan, int patch_geo, int ph, int pw, int Ci) {
    if (!patch_geo || !plan.bdirect) return false;
    if ((ph * pw) % plan.bmt != 0 || plan.k_splits > (Ci + XBK - 1) / XBK) return false;
    if (pw == 64) return plan.bn == 64 ? plan.bmt == 256 : true;
    return plan.bn == 128 && (pw == 32 || pw == 16 || plan.bmt == 64);
}

// Ci / Co are the reduction / output channel counts of the IMAGE (mode 1: those of the backward-data GEMM).
static int x6_forward_impl(const float* X, const void* image, const float* bias, float* Y, int B, int Hi, int Wi, int Ci, int Ho,
                           int Wo, int Co, int KH, int KW, int stride, int dil, int pad_h, int pad_w, int k_splits,
                           float* bn_stats, int* bn_rows, X6Ep ep, int* ep_applied, dsf_stream_t stream, bool y_ready = false) {
    if (bn_rows) *bn_rows = 0;
    DSF_CHECK_ARG(X && image && Y && B >= 0 && Hi > 0 && Wi > 0 && Ci > 0 && Ho > 0 && Wo > 0 && Co > 0 && KH > 0 && KW > 0);
    DSF_CHECK_ARG(stride >= 1 && (Ci & 3) == 0 && (dil == 1 || (dil == 2 && stride == 1)));
    if (dil == 2 && ((Ho | Wo) & 1)) return DSF_ERR_UNSUPPORTED;
    if (B == 0) return DSF_OK;
    X6P p = {B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad_h, pad_w};
    const int64_t M = (int64_t)B * Ho * Wo;
    const int64_t x_bytes = (int64_t)B * Hi * Wi * Ci * 4, w_bytes = dsf_conv_x6_image_bytes(KH, KW, Ci, Co);
    DSF_CHECK_ARG(M < (1ll << 31) && x_bytes < 0xFFFFFFF0ll && w_bytes < 0xFFFFFFF0ll);
    const int patch_geo = x6_patch_geometry(Hi, Wi, Ho, Wo, KH, KW, stride, dil, pad_h, pad_w);
    const int ph = patch_geo == 4 ? Hi : Ho, pw = patch_geo == 4 ? Wi : Wo;       // the image the patch kernel's tile rows index
    X6Plan plan = x6_forward_plan(M, Ci, Co, KH, KW, dil, k_splits, patch_geo ? pw : 0);
    // a 1 x 1 filter under dilation 2 (backward-data of a 1 x 1 stride-2 shortcut): one output pixel in four has a tap.  Launch the
    // rows of that parity class only -- their epilogue stores the zeros of the three siblings -- instead of 4x the tiles, three
    // quarters of which only store zeros (B = 192, 32x32x512 -> 64x64x256: 812 -> ... us).  DSF_X6_LIVE=0: off.
    int live_cls = -1;
    if (dil == 2 && KH == 1 && KW == 1 && !bias && !ep.scale && !bn_stats && !y_ready && k_splits < 2) {
        const char* live_e = getenv("DSF_X6_LIVE");                       // read per call: the test compares both launches
        const int live_env = live_e ? atoi(live_e) : 1;
        const X6Plan lp = x6_forward_plan(M / 4, Ci, Co, 1, 1, 1, k_splits);
        if (live_env && lp.k_splits == 1 && lp.bdirect && !(lp.bmt == 64 && lp.n_tiles >= 2)) {
            plan = lp;
            live_cls = ((pad_h & 1) << 1) | (pad_w & 1);
        }
    }
    const int bn = plan.bn, n_tiles = plan.n_tiles, bdirect = plan.bdirect, bmt = plan.bmt, m_tiles = plan.m_tiles;
    k_splits = plan.k_splits;
    if (y_ready && k_splits < 2) return DSF_ERR_UNSUPPORTED;             // only the split launches ADD into Y
    if (k_splits > 1 && !y_ready &&
        dsf_zero_async(Y, sizeof(float) * (size_t)M * Co, (hipStream_t)stream) != hipSuccess) return DSF_ERR_LAUNCH;
    const dim3 grid(m_tiles * n_tiles * k_splits);
    // BatchNorm statistics in the epilogue: only the B-direct kernels, unsplit, without a bias
    const bool patch = x6_patch_applies(plan, patch_geo, ph, pw, Ci);
    const bool direct_pre = patch || (bdirect && !(bmt == 64 && n_tiles >= 2));
    float* stats = (bn_stats && bn_rows && direct_pre && k_splits == 1 && !bias) ? bn_stats : nullptr;
    if (bn_rows) *bn_rows = stats ? m_tiles : 0;
    // the affine output epilogue has the same conditions (a split reduction meets in Y by atomics; the staged kernel has none)
    const bool ep_on = ep.scale && direct_pre && k_splits == 1;
    if (!ep_on) ep = X6Ep{nullptr, nullptr, nullptr, 0, ep.stats_acc};
    ep.live_cls = live_cls;
    if (ep_applied) *ep_applied = ep_on ? 1 : 0;
#define DSF_LAUNCH_X6(KERNEL, BNv, DILv, BMv) hipLaunchKernelGGL((KERNEL<BNv, DILv, BMv>), grid, dim3(256), 0, (hipStream_t)stream, \
                                                       X, (const uint4*)image, bias, Y, p, m_tiles, n_tiles, k_splits,              \
                                                       (uint32_t)x_bytes, (uint32_t)w_bytes)
#define DSF_LAUNCH_X6B(BNv, DILv, BMv) hipLaunchKernelGGL((igemm_x6b_kernel<BNv, DILv, BMv>), grid, dim3(256), 0, (hipStream_t)stream, \
                                                       X, (const uint4*)image, bias, Y, p, m_tiles, n_tiles, k_splits,              \
                                                       (uint32_t)x_bytes, (uint32_t)w_bytes, stats, ep)
    // 64-row tiles with several n tiles (16x16x256, 8x8x512 maps): every wave of the many small tiles would pull its own copy of
    // the weight block from L2 -- the LDS-staged kernel is faster there (measured 136 / 139 vs 133 / 133 TFLOP/s)
    const bool direct = bdirect && !(bmt == 64 && n_tiles >= 2);
    if (patch) {
#define DSF_LAUNCH_X6P_(BNv, BMv, Wv, NWv, BDv, NTv, IPv)                                                                         \
    do {                                                                                                                          \
        using PT = X6Patch<BMv, Wv, NTv>;                                                                                         \
        static bool armed[X6_MAX_DEVICES];                   /* per device: the attribute belongs to the device that was current */ \
        if (!x6_arm_dynamic_lds(reinterpret_cast<const void*>(&igemm_x6p_kernel<BNv, BMv, Wv, NWv, BDv, NTv, IPv>), PT::LDS_BYTES, armed)) \
            return DSF_ERR_LAUNCH;                                                                                                 \
        hipLaunchKernelGGL((igemm_x6p_kernel<BNv, BMv, Wv, NWv, BDv, NTv, IPv>), grid, dim3(256), PT::LDS_BYTES, (hipStream_t)stream,  \
                           X, (const uint4*)image, bias, Y, p, m_tiles, n_tiles, k_splits, (uint32_t)x_bytes, (uint32_t)w_bytes,  \
                           stats, ep);                                                                                            \
    } while (0)
#define DSF_LAUNCH_X6P(BNv, BMv, Wv, NWv, BDv, NTv) DSF_LAUNCH_X6P_(BNv, BMv, Wv, NWv, BDv, NTv, false)
        // 64-row tiles: 1 x 4 waves and weight fragments two taps ahead (B = 32, 16x16x256 unsplit: 2 x 2 waves 55 us, 1 x 4 51,
        // + two taps ahead 48; no gain from either on the taller tiles, which have two workgroups per CU to hide the latency)
        if (patch_geo == 5) {                                            // 4 x 4 stride 2 by input parity classes: pw = the output map's width
            if (bmt == 128) { if (pw == 32) DSF_LAUNCH_X6P_(128, 128, 32, 2, 2, 4, true); else DSF_LAUNCH_X6P_(128, 128, 16, 2, 2, 4, true); }
            else if (pw == 32) DSF_LAUNCH_X6P_(128, 64, 32, 4, 2, 4, true);
            else if (pw == 16) DSF_LAUNCH_X6P_(128, 64, 16, 4, 2, 4, true);
            else DSF_LAUNCH_X6P_(128, 64, 8, 4, 2, 4, true);
        }
        else if (patch_geo == 4) {                                       // transposed 4 x 4 stride 2: pw = the class image's width
            if (bmt == 128) { if (pw == 32) DSF_LAUNCH_X6P(128, 128, 32, 2, 2, 4); else DSF_LAUNCH_X6P(128, 128, 16, 2, 2, 4); }
            else if (pw == 32) DSF_LAUNCH_X6P(128, 64, 32, 4, 2, 4);
            else if (pw == 16) DSF_LAUNCH_X6P(128, 64, 16, 4, 2, 4);
            else DSF_LAUNCH_X6P(128, 64, 8, 4, 2, 4);
        }
        else if (bn == 64) DSF_LAUNCH_X6P(64, 256, 64, 2, 2, 9);
        else if (bmt == 128) {
            if (pw == 64) DSF_LAUNCH_X6P(128, 128, 64, 2, 2, 9); else if (pw == 32) DSF_LAUNCH_X6P(128, 128, 32, 2, 2, 9);
            else DSF_LAUNCH_X6P(128, 128, 16, 2, 2, 9);
        }
        else if (pw == 64) DSF_LAUNCH_X6P(128, 64, 64, 4, 3, 9);
        else if (pw == 32) DSF_LAUNCH_X6P(128, 64, 32, 4, 3, 9);
        else if (pw == 16) DSF_LAUNCH_X6P(128, 64, 16, 4, 3, 9);
        else DSF_LAUNCH_X6P(128, 64, 8, 4, 3, 9);
#undef DSF_LAUNCH_X6P
#undef DSF_LAUNCH_X6P_
        return dsf_launch_status();
    }
    if (direct) {
        if (dil == 2) { if (bn == 64) DSF_LAUNCH_X6B(64, true, 256); else if (bmt == 64) DSF_LAUNCH_X6B(128, true, 64); else DSF_LAUNCH_X6B(128, true, 128); }
        else { if (bn == 64) DSF_LAUNCH_X6B(64, false, 256); else if (bmt == 64) DSF_LAUNCH_X6B(128, false, 64); else DSF_LAUNCH_X6B(128, false, 128); }
    } else {
        if (dil == 2) { if (bn == 64) DSF_LAUNCH_X6(igemm_x6_kernel, 64, true, 128); else if (bmt == 64) DSF_LAUNCH_X6(igemm_x6_kernel, 128, true, 64); else DSF_LAUNCH_X6(igemm_x6_kernel, 128, true, 128); }
        else { if (bn == 64) DSF_LAUNCH_X6(igemm_x6_kernel, 64, false, 128); else if (bmt == 64) DSF_LAUNCH_X6(igemm_x6_kernel, 128, false, 64); else DSF_LAUNCH_X6(igemm_x6_kernel, 128, false, 128); }
    }
#undef DSF_LAUNCH_X6
#undef DSF_LAUNCH_X6B
    return dsf_launch_status();
}

static int x6_wrw_plan(int B, int Ho, int Wo, int Ci, int Co, int KH, int KW, int& k_tiles, int& n_tiles, int64_t& per) {
    const int64_t M = (int64_t)B * Ho * Wo;
    const int K = KH * KW * Ci;
    const int bn = x6_bn(Co);
    k_tiles = (K + 127) / 128; n_tiles = (Co + bn - 1) / bn;
    // split the pixel reduction towards ONE workgroup per CU (256), >= 4 chunks per split -- towards two per CU (512) only where
    // that still leaves a workgroup 1024 chunks or more (the large-batch layers of config 4).  Inside a step these launches run BESIDE
    // the main queue's kernels: (probably) two of these workgroups hold 96 KB of a CU's LDS and a patch-staged forward workgroup cannot
    // join them, one leaves it room; and every workgroup ends with 64 KB of float atomics (~13 us of its CU's atomic path:
    // tools/x6/wrw_stamps.py), half as many at half the workgroups.  Whole steps, one box, alternating blocks
    // (profiles/r06_wrw_split_target.txt): config 2 17.07 -> 16.62 ms, config 3 15.94 -> 15.34; config 4 159.2 vs 159.7 the other way.
    // ALONE, a small-map launch is ~10 % slower at 256 (one wave per SIMD): the isolated optimum was the wrong target for rounds 3-5.
    const char* wg_e = getenv("DSF_X6_WRW_WGS");                         // tuning aid, read per call (tools/ab_env.py alternates it in one process)
    const int wg_env = wg_e ? atoi(wg_e) : 0;
    int target = wg_env > 0 ? wg_env : 512;
    if (wg_env <= 0) {
        const int s512 = 512 / (k_tiles * n_tiles) > 0 ? 512 / (k_tiles * n_tiles) : 1;
        if (M / s512 < 1024 * XBK) target = 256;
    }
    int splits = target / (k_tiles * n_tiles);
    if (splits < 1) splits = 1;
    per = (M + splits - 1) / splits;
    per = ((per + 2 * XBK - 1) / (2 * XBK)) * (2 * XBK);
    if (per < 4 * XBK) per = 4 * XBK;
    return (int)((M + per - 1) / per);
}

// bytes of scratch dsf_conv_x6_wrw needs in deterministic mode (0 otherwise): one dW-sized tile per pixel split
int dsf_conv_x6_forward(const float* X, const void* image, const float* bias, float* Y, int B, int Hi, int Wi, int Ci, int Ho,
                        int Wo, int Co, int KH, int KW, int stride, int dil, int pad_h, int pad_w, int k_splits,
                        dsf_stream_t stream) {
    return x6_forward_impl(X, image, bias, Y, B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, dil, pad_h, pad_w, k_splits, nullptr, nullptr,
                           X6Ep{nullptr, nullptr, nullptr, 0, 0}, nullptr, stream);
}

// The K splits dsf_conv_x6_forward would choose for this shape (1: an unsplit launch that stores Y).
int dsf_conv_x6_forward_splits(int B, int Ho, int Wo, int Ci, int Co, int KH, int KW, int dil) {
    if (B <= 0 || Ho <= 0 || Wo <= 0 || Ci <= 0 || Co <= 0 || KH <= 0 || KW <= 0 || (dil != 1 && dil != 2)) return 1;
    return x6_forward_plan((int64_t)B * Ho * Wo, Ci, Co, KH, KW, dil, 0).k_splits;
}

// What dsf_conv_x6_forward (k_splits <= 0) launches for this layer.  *variant: 0 igemm_x6_kernel (both operands through LDS),
// 1 igemm_x6b_kernel (weights straight into the fragment registers), 2 igemm_x6p_kernel (the same with a patch-staged input);
// *k_splits: its K splits (1: an unsplit launch that stores Y).  Either pointer may be null.
int dsf_conv_x6_forward_plan(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int dil, int pad_h,
                             int pad_w, int* variant, int* k_splits) {
    DSF_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && Ci > 0 && Co > 0 && KH > 0 && KW > 0 && (dil == 1 || dil == 2));
    const int patch_geo = x6_patch_geometry(Hi, Wi, Ho, Wo, KH, KW, stride, dil, pad_h, pad_w);
    const int ph = patch_geo == 4 ? Hi : Ho, pw = patch_geo == 4 ? Wi : Wo;
    const X6Plan plan = x6_forward_plan((int64_t)B * Ho * Wo, Ci, Co, KH, KW, dil, 0, patch_geo ? pw : 0);
    if (variant)
        *variant = x6_patch_applies(plan, patch_geo, ph, pw, Ci) ? 2 : (plan.bdirect && !(plan.bmt == 64 && plan.n_tiles >= 2)) ? 1 : 0;
    if (k_splits) *k_splits = plan.k_splits;
    // a 1 x 1 filter under dilation 2 that x6_forward_impl would launch over its live parity class only (unsplit, no zero fill): report
    // THAT launch (variant 3, one split), so that a caller holding pooled zeros does not route the layer into the split gather
    // (dsf_conv_x6_forward_into disables the live launch) -- the advisor's round-5 finding; bias / epilogue users get the general plan
    // from the launcher itself either way
    if (dil == 2 && KH == 1 && KW == 1 && !((Ho | Wo) & 1)) {
        const char* live_e = getenv("DSF_X6_LIVE");
        const X6Plan lp = x6_forward_plan((int64_t)B * Ho * Wo / 4, Ci, Co, 1, 1, 1, 0);
        if (!(live_e && atoi(live_e) == 0) && lp.k_splits == 1 && lp.bdirect && !(lp.bmt == 64 && lp.n_tiles >= 2)) {
            if (variant) *variant = 3;
            if (k_splits) *k_splits = 1;
        }
    }
    return DSF_OK;
}

// dsf_conv_x6_forward as a split launch that ADDS into a Y the caller has initialised (zeros from one pooled fill instead of a
// fill per layer, or a residual): k_splits >= 2 as reported by dsf_conv_x6_forward_splits; DSF_ERR_UNSUPPORTED otherwise and in
// deterministic mode (which never splits).
int dsf_conv_x6_forward_into(const float* X, const void* image, const float* bias, float* Y, int B, int Hi, int Wi, int Ci, int Ho,
                             int Wo, int Co, int KH, int KW, int stride, int dil, int pad_h, int pad_w, int k_splits,
                             dsf_stream_t stream) {
    if (k_splits < 2) return DSF_ERR_UNSUPPORTED;
    return x6_forward_impl(X, image, bias, Y, B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, dil, pad_h, pad_w, k_splits, nullptr, nullptr,
                           X6Ep{nullptr, nullptr, nullptr, 0, 0}, nullptr, stream, true);
}

// partial rows a BatchNorm-statistics epilogue may write for an (M = B Ho Wo)-row output: one per 64 rows at most
int dsf_conv_x6_bn_stats_rows(int B, int Ho, int Wo) { return (int)(((int64_t)B * Ho * Wo + 63) / 64); }

int dsf_conv_x6_forward_bn(const float* X, const void* image, float* Y, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                           int KH, int KW, int stride, int dil, int pad_h, int pad_w, float* bn_stats, int* bn_rows,
                           dsf_stream_t stream) {
    DSF_CHECK_ARG(bn_stats && bn_rows);
    return x6_forward_impl(X, image, nullptr, Y, B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, dil, pad_h, pad_w, 0, bn_stats, bn_rows,
                           X6Ep{nullptr, nullptr, nullptr, 0, 0}, nullptr, stream);
}

// As dsf_conv_x6_forward_bn, but the tile sums are ADDED (double atomics) into the caller-zeroed block `acc` of
// dsf_bn_acc_rows() rows [row][2][Co] of doubles that dsf_bn_forward_acc folds in its own prologue (no finalise launch).  *filled = 1
// when the launch this shape takes wrote them (else Y is the plain convolution and `acc` is untouched).  Not in
// deterministic mode (DSF_ERR_UNSUPPORTED).
int dsf_conv_x6_forward_bn_acc(const float* X, const void* image, float* Y, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                               int KH, int KW, int stride, int dil, int pad_h, int pad_w, double* acc, int acc_rows, int* filled,
                               dsf_stream_t stream) {
    DSF_CHECK_ARG(acc && filled && acc_rows > 0);
    if (dsf_deterministic()) return DSF_ERR_UNSUPPORTED;
    int rows = 0;
    const int rc = x6_forward_impl(X, image, nullptr, Y, B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, dil, pad_h, pad_w, 0,
                                   reinterpret_cast<float*>(acc), &rows,
                                   X6Ep{nullptr, nullptr, nullptr, 0, acc_rows}, nullptr, stream);
    *filled = rows > 0 ? 1 : 0;
    return rc;
}

// Convolution with a fused per-channel output epilogue: Y = act((conv + bias) * scale[c] + shift[c] (+ residual)), act = ReLU
// when relu != 0 -- a frozen-statistics BatchNorm (scale = gamma * invstd, shift = beta - mean * scale), the block's skip
// connection and its activation without a second pass over Y.  *applied = 1 when the epilogue ran; 0 when the launch this
// shape takes cannot carry it (split reduction, staged kernel): Y then holds the plain convolution (+ bias) and the caller
// applies the rest (dsf_bn_apply).
int dsf_conv_x6_forward_affine(const float* X, const void* image, const float* bias, float* Y, int B, int Hi, int Wi, int Ci,
                               int Ho, int Wo, int Co, int KH, int KW, int stride, int dil, int pad_h, int pad_w,
                               const float* scale, const float* shift, const float* residual, int relu, int* applied,
                               dsf_stream_t stream) {
    DSF_CHECK_ARG(scale && shift && applied);
    return x6_forward_impl(X, image, bias, Y, B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, dil, pad_h, pad_w, 0, nullptr, nullptr,
                           X6Ep{scale, shift, residual, relu, 0}, applied, stream);
}

// igemm_wrw_x6p_kernel's tiling: 32-channel blocks x 128-wide n tiles x row splits (whole output rows, at least four each, pairs on
// 16-wide maps) towards one workgroup per CU (rounds 4-5: two)
static int x6_wrw_patch_plan(int B, int H, int W, int Ci, int Co, int& c_tiles, int& n_tiles, int& rows_per_split) {
    c_tiles = (Ci + 31) / 32; n_tiles = (Co + 127) / 128;
    const int RT = B * H;
    const char* wg_e = getenv("DSF_X6_WRWP_WGS");                        // tuning aid, read per call
    const int wg_target = (wg_e && atoi(wg_e) > 0) ? atoi(wg_e) : 256;   // one workgroup per CU: see x6_wrw_plan (config 2, alternating blocks: 16.28 -> 16.20 ms)
    int splits = wg_target / (c_tiles * n_tiles);
    if (splits < 1) splits = 1;
    rows_per_split = (RT + splits - 1) / splits;
    if (rows_per_split < 4) rows_per_split = 4;
    // 64-wide maps: longer splits of 1024 pixels where that still leaves a workgroup per CU (the 64 -> 64 layers at B = 32:
    // 256 x 16 rows, 80.7 -> 73.8 us; on the narrower maps the old kernel wins such cases)
    const int want = 1024 / W;
    if (W == 64 && rows_per_split < want && c_tiles * n_tiles * ((RT + want - 1) / want) >= 256) rows_per_split = want;
    if (W == 16) rows_per_split += rows_per_split & 1;
    return (RT + rows_per_split - 1) / rows_per_split;
}
// the layers it serves (DSF_X6_WRW_PATCH=0 switches it off, 2 drops the size rule; read per call): 3 x 3, stride 1, pad 1 on 64 / 32 / 16-wide maps with at
// least 1024 pixels per split -- every split adds its 288 x 128 tile into dW with float atomics, and on short splits that traffic
// (B = 32, 32x32x128: 128 splits of 256 pixels, 19 M atomics) costs more than the loader saves (84 us against 66)
static bool x6_wrw_patch_applies(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad_h,
                                 int pad_w) {
    const char* e = getenv("DSF_X6_WRW_PATCH");
    const int level = e ? atoi(e) : 1;                                   // 0: off;  2: wherever the geometry fits (tests)
    if (level <= 0) return false;
    if (!(KH == 3 && KW == 3 && stride == 1 && pad_h == 1 && pad_w == 1 && Ho == Hi && Wo == Wi &&
          (Wi == 64 || Wi == 32 || Wi == 16))) return false;
    int c_tiles, n_tiles, rows;
    x6_wrw_patch_plan(B, Hi, Wi, Ci, Co, c_tiles, n_tiles, rows);
    return level >= 2 || rows * Wi >= 1024;
}

int64_t dsf_conv_x6_wrw_workspace_bytes(int B, int Ho, int Wo, int Ci, int Co, int KH, int KW) {
    if (!dsf_deterministic() || B <= 0) return 0;
    int k_tiles, n_tiles; int64_t per;
    int splits = x6_wrw_plan(B, Ho, Wo, Ci, Co, KH, KW, k_tiles, n_tiles, per);
    if (KH == 3 && KW == 3 && (Wo == 64 || Wo == 32 || Wo == 16)) {     // (stride and padding unknown here: the larger of the two)
        int c_tiles, rows;
        const int ps = x6_wrw_patch_plan(B, Ho, Wo, Ci, Co, c_tiles, n_tiles, rows);
        if (ps > splits) splits = ps;
    }
    return (int64_t)splits * KH * KW * Ci * Co * 4;
}

static int x6_wrw_impl(const float* X, const float* dY, float* dW, float* dbias, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                       int KH, int KW, int stride, int pad_h, int pad_w, int accumulate, float* workspace, dsf_stream_t stream) {
    DSF_CHECK_ARG(X && dY && dW && B >= 0 && Hi > 0 && Wi > 0 && Ci > 0 && Ho > 0 && Wo > 0 && Co > 0 && KH > 0 && KW > 0);
    DSF_CHECK_ARG(stride >= 1 && (Ci & 3) == 0 && (Co & 3) == 0);
    const int K = KH * KW * Ci;
    const bool det = dsf_deterministic() != 0;
    DSF_CHECK_ARG(!det || workspace || B == 0);
    if (!accumulate && !(det && B > 0) &&
        dsf_zero_async(dW, sizeof(float) * (size_t)K * Co, (hipStream_t)stream) != hipSuccess) return DSF_ERR_LAUNCH;
    if (B == 0) return DSF_OK;
    X6P p = {B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad_h, pad_w};
    const int64_t M = (int64_t)B * Ho * Wo;
    const int64_t x_bytes = (int64_t)B * Hi * Wi * Ci * 4, dy_bytes = M * Co * 4;
    DSF_CHECK_ARG(M < (1ll << 31) && x_bytes < 0xFFFFFFF0ll && dy_bytes < 0xFFFFFFF0ll);
    const int bn = x6_bn(Co);
    float* partial = det ? workspace : nullptr;
    if (x6_wrw_patch_applies(B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad_h, pad_w)) {
        int c_tiles, pn_tiles, rows;
        const int psplits = x6_wrw_patch_plan(B, Hi, Wi, Ci, Co, c_tiles, pn_tiles, rows);
#define DSF_LAUNCH_WRWP(Wv, BNv)                                                                                                  \
    do {                                                                                                                          \
        using PT = X6WrwPatch<Wv>;                                                                                                \
        static bool armed[X6_MAX_DEVICES];                                                                                        \
        if (!x6_arm_dynamic_lds(reinterpret_cast<const void*>(&igemm_wrw_x6p_kernel<Wv, BNv>), PT::LDS_BYTES, armed))             \
            return DSF_ERR_LAUNCH;                                                                                                 \
        hipLaunchKernelGGL((igemm_wrw_x6p_kernel<Wv, BNv>), dim3(c_tiles * pn_tiles * psplits), dim3(256), PT::LDS_BYTES,        \
                           (hipStream_t)stream, X, dY, dW, p, c_tiles, pn_tiles, psplits, rows, (uint32_t)x_bytes,               \
                           (uint32_t)dy_bytes, partial);                                                                          \
    } while (0)
        if (bn == 128) { if (Wi == 64) DSF_LAUNCH_WRWP(64, 128); else if (Wi == 32) DSF_LAUNCH_WRWP(32, 128); else DSF_LAUNCH_WRWP(16, 128); }
        else { if (Wi == 64) DSF_LAUNCH_WRWP(64, 64); else if (Wi == 32) DSF_LAUNCH_WRWP(32, 64); else DSF_LAUNCH_WRWP(16, 64); }
#undef DSF_LAUNCH_WRWP
        if (det) {
            const int64_t n = (int64_t)K * Co;
            hipLaunchKernelGGL(x6_wrw_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial,
                               dW, n, psplits, accumulate);
        }
        return dsf_launch_status();
    }
    int k_tiles, n_tiles; int64_t per;
    const int splits = x6_wrw_plan(B, Ho, Wo, Ci, Co, KH, KW, k_tiles, n_tiles, per);
    const uint64_t mwo = ((1ull << 40) + Wo - 1) / Wo, mho = ((1ull << 40) + Ho - 1) / Ho;
#define DSF_LAUNCH_WRW(BNv, BIASv)                                                                                                \
    hipLaunchKernelGGL((igemm_wrw_x6_kernel<BNv, BIASv>), dim3(k_tiles * n_tiles * splits), dim3(256), 0, (hipStream_t)stream, X, dY, \
                       dW, p, k_tiles, n_tiles, splits, (int)per, mwo, mho, (uint32_t)x_bytes, (uint32_t)dy_bytes, partial, dbias)
    if (bn == 128) { if (dbias) DSF_LAUNCH_WRW(128, true); else DSF_LAUNCH_WRW(128, false); }
    else { if (dbias) DSF_LAUNCH_WRW(64, true); else DSF_LAUNCH_WRW(64, false); }
#undef DSF_LAUNCH_WRW
    if (det) {
        const int64_t n = (int64_t)K * Co;
        hipLaunchKernelGGL(x6_wrw_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial, dW,
                           n, splits, accumulate);
    }
    return dsf_launch_status();
}

int dsf_conv_x6_wrw_ws(const float* X, const float* dY, float* dW, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                       int KW, int stride, int pad_h, int pad_w, int accumulate, float* workspace, dsf_stream_t stream) {
    return x6_wrw_impl(X, dY, dW, nullptr, B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad_h, pad_w, accumulate, workspace, stream);
}

// dsf_conv_x6_wrw with the bias gradient from the same launch: dbias[co] += sum over pixels of dY[.][co] (float atomics: dbias is
// zeroed, or holds what it accumulates into, as dW with accumulate != 0).  DSF_ERR_UNSUPPORTED in deterministic mode (its ordered
// column sums are dsf_col_sum's) -- nothing is launched then.
int dsf_conv_x6_wrw_bias(const float* X, const float* dY, float* dW, float* dbias, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                         int KH, int KW, int stride, int pad_h, int pad_w, int accumulate, dsf_stream_t stream) {
    DSF_CHECK_ARG(dbias && B >= 0 && Ho > 0 && Wo > 0 && Ci > 0 && Co > 0 && KH > 0 && KW > 0);
    if (dsf_deterministic()) return DSF_ERR_UNSUPPORTED;
    {   // every pixel split adds its column sums into the same Co addresses: with hundreds of splits (the 64x64-map layers: 256) the
        // atomics queue up behind each other (+250 us on the 256 -> 84 head, measured) -- those layers keep dsf_col_sum
        // (the layers of the row-staged kernel are large ones: a column-sum pass is noise beside them, and its nine accumulators
        //  leave no registers for the sums)
        if (x6_wrw_patch_applies(B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad_h, pad_w)) return DSF_ERR_UNSUPPORTED;
        int k_tiles, n_tiles; int64_t per;
        const int splits = x6_wrw_plan(B, Ho, Wo, Ci, Co, KH, KW, k_tiles, n_tiles, per);
        const char* ms_e = getenv("DSF_WRW_BIAS_MAX_SPLITS");            // tuning aid
        if (splits > (ms_e ? atoi(ms_e) : 64)) return DSF_ERR_UNSUPPORTED;
    }
    return x6_wrw_impl(X, dY, dW, dbias, B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad_h, pad_w, accumulate, nullptr, stream);
}

int dsf_conv_x6_wrw(const float* X, const float* dY, float* dW, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                    int KW, int stride, int pad_h, int pad_w, int accumulate, dsf_stream_t stream) {
    if (dsf_deterministic() && B > 0) return DSF_ERR_INVALID_ARG;        // deterministic mode needs the scratch: dsf_conv_x6_wrw_ws
    return dsf_conv_x6_wrw_ws(X, dY, dW, B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad_h, pad_w, accumulate, nullptr, stream);
}

}  // extern "C"
