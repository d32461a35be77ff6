// Stand-alone reproducer (no torch, no libdsf_hip.so) of a wrong-result condition on MI355X (gfx950, ROCm 7.2), round 5:
//
//   A packed-FP32 instruction (v_pk_mul_f32, v_pk_fma_f32, v_pk_add_f32) whose LOW result takes the LOW register of source 0
//   and the HIGH register of source 1 -- `op_sel:[0,1]` / `op_sel:[0,1,0]`, e.g.
//   `v_pk_mul_f32 v[22:23], v[24:25], v[20:21] op_sel:[0,1]` -- occasionally computes that low result with source 1 read
//   as 0.0 in lanes 48-63 of the wave, while a wave issuing v_mfma_f32_32x32x16_bf16 runs on the same SIMD.
//   Never on an idle GPU; never beside fp32-input MFMAs (v_mfma_f32_32x32x2_f32) or LDS traffic alone; never with op_sel
//   [1,0], [1,1], [0,0,1], op_sel_hi variants or the default selects; v_pk_mul_f16 op_sel:[0,1], v_div_fmas_f32 and
//   v_cndmask_b32 with VCC all ones are not affected.  Measured (profiles/r05_pk_opsel_erratum.txt): 18-20 of 20 launches of
//   1024 workgroups wrong for each of the three instructions, always exactly lanes 48-63, always the low half, 10-50 of the
//   4096 iterations of a lane; every other cell 0 of 20.
//
//   hipcc --offload-arch=gfx950 -O3 tools/platform/pk_opsel_beside_mfma_lds.hip -o /tmp/pk_opsel && /tmp/pk_opsel [launches]
//
// How it was found: hipcc's SLP vectoriser turned the 3x3 transform at the end of the MANO skinning backward
// (dsf_amd/csrc/mano.hip) into v_pk_mul_f32 / v_pk_add_f32; one multiply carries op_sel:[0,1], and d/d(v_posed).x of vertices
// 48-63 of every wave lost exactly that instruction's term (T[3] * g1: the damaged outputs equal the sum without it to all
// digits) whenever conv_x6 workgroups (bf16 MFMAs) shared the CU -- 100 of 100 calls (tools/platform/war_bisect.py on assembly
// variants from war_variants.py: rewriting ONLY the packed multiplies / adds of that epilogue as single-float instructions
// removes the fault; s_nop / s_waitcnt padding, splitting the LDS reads, rewriting the loop's packed FMAs do not).
// Consequence for the product: built with -fno-slp-vectorize -fno-vectorize; dsf_amd/csrc/isa_lint.py (run by build.sh and by
// tests/test_isa_lint.py) fails on any packed-FP32 instruction in any code object, so the claim is checked, not assumed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// src0 = (v24, v25) = (1, 2); src1 = (v20, v21) = (3, 5); src2 = (v26, v27) = (7, 11); result pair (v22, v23) is added to
// (v28, v29) with plain v_add_f32 n times.  All values are small integers: every sum is exact.
#define VICTIM(NAME, INSTR)                                                                                               \
    __global__ __launch_bounds__(256, 3) void NAME(float* __restrict__ out, int n) {                                       \
        float o0, o1;                                                                                                     \
        asm volatile(                                                                                                     \
            "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 2.0\n v_mov_b32 v20, 0x40400000\n v_mov_b32 v21, 0x40a00000\n"              \
            "v_mov_b32 v26, 0x40e00000\n v_mov_b32 v27, 0x41300000\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n s_mov_b32 s20, %[n]\n"                   \
            "v_mov_b32 v30, 0x40003c00\n v_mov_b32 v31, 0x44004200\n s_mov_b64 vcc, -1\n"  \
            "1:\n"                                                                                                        \
            "v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n s_nop 1\n"                                                              \
            INSTR "\n"                                                                                                    \
            "s_nop 1\n v_add_f32 v28, v28, v22\n v_add_f32 v29, v29, v23\n s_mov_b64 vcc, -1\n"                                                \
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"                                            \
            "s_nop 4\n v_mov_b32 %[o0], v28\n v_mov_b32 %[o1], v29\n"                                                      \
            : [o0] "=v"(o0), [o1] "=v"(o1) : [n] "s"(n)                                                                   \
            : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "s20", "scc", "vcc");                           \
        out[(size_t)blockIdx.x * 512 + threadIdx.x * 2] = o0;                                                             \
        out[(size_t)blockIdx.x * 512 + threadIdx.x * 2 + 1] = o1;                                                         \
    }
VICTIM(victim0, "v_pk_mul_f32 v[22:23], v[24:25], v[20:21] op_sel:[0,1]")          // lo = 1*5, hi = 2*5
VICTIM(victim1, "v_pk_mul_f32 v[22:23], v[24:25], v[20:21] op_sel:[1,0]")          // lo = 2*3, hi = 2*5
VICTIM(victim2, "v_pk_mul_f32 v[22:23], v[24:25], v[20:21] op_sel_hi:[1,0]")       // lo = 1*3, hi = 2*3
VICTIM(victim3, "v_pk_mul_f32 v[22:23], v[24:25], v[20:21]")                       // lo = 1*3, hi = 2*5
VICTIM(victim4, "v_pk_fma_f32 v[22:23], v[24:25], v[20:21], v[26:27] op_sel:[0,1,0]")   // lo = 1*5+7, hi = 2*5+11
VICTIM(victim5, "v_pk_add_f32 v[22:23], v[24:25], v[20:21] op_sel:[0,1]")          // lo = 1+5, hi = 2+5
VICTIM(victim6, "v_pk_mul_f32 v[22:23], v[24:25], v[20:21] op_sel:[1,1]")          // lo = 2*5, hi = 2*5
VICTIM(victim7, "v_pk_fma_f32 v[22:23], v[24:25], v[20:21], v[26:27] op_sel:[0,0,1]")   // lo = 1*3+11, hi = 2*5+11
// v_div_fmas_f32 reads VCC per lane (set to all ones by the loop): fma(1, 3, 7) scaled by 2^32 in every lane
VICTIM(victim8, "v_div_fmas_f32 v22, v24, v20, v26\n v_mov_b32 v23, v22")
// v_cndmask_b32 reading a 64-bit lane mask from VCC
VICTIM(victim9, "v_cndmask_b32 v22, v24, v21, vcc\n v_cndmask_b32 v23, v25, v20, vcc")
// packed fp16 with the same source-1 high select: halves of v20 / v21 as fp16 pairs; (v20 = 0x40400000: hi half 2.125 fp16, lo 0)
VICTIM(victim10, "v_pk_mul_f16 v22, v30, v31 op_sel:[0,1]\n v_cvt_f32_f16 v23, v22\n v_cvt_f32_f16_sdwa v22, v22 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
static const float EXPECT[11][2] = {{5, 10}, {6, 10}, {3, 6}, {3, 10}, {12, 21}, {6, 7}, {10, 10}, {14, 21}, {42949672960.f, 42949672960.f}, {5, 3}, {8, 4}};
static const char* WHAT[11] = {"v_pk_mul_f32 op_sel:[0,1]", "v_pk_mul_f32 op_sel:[1,0]", "v_pk_mul_f32 op_sel_hi:[1,0]", "v_pk_mul_f32 (default)",
                               "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_add_f32 op_sel:[0,1]", "v_pk_mul_f32 op_sel:[1,1]", "v_pk_fma_f32 op_sel:[0,0,1]",
                               "v_div_fmas_f32 (VCC all ones)", "v_cndmask_b32 (VCC all ones)", "v_pk_mul_f16 op_sel:[0,1]"};

// the side load: MFMA = bf16 / fp32 / none, LDS reads (ds_read_b128) on or off; 256 threads, 32 KB of LDS
template <int MFMA, bool LDS>
__global__ __launch_bounds__(256, 2) void side_kernel(const u32x4* __restrict__ in, float* __restrict__ out, int iters) {
    __shared__ u32x4 tile[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) tile[i] = in[i & 1023];
    __syncthreads();
    u32x4 ra = in[threadIdx.x], rb = in[256 + threadIdx.x];
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    uint32_t x = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (LDS) { rb = tile[(threadIdx.x + 64 * t + it * 7) & 2047]; x ^= rb[3]; }
            if (MFMA == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra), __builtin_bit_cast(bf16x8, rb), acc[t], 0, 0, 0);
            if (MFMA == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, (ra[0] & 0x007fffffu) | 0x3f800000u),
                                                                         __builtin_bit_cast(float, (rb[0] & 0x007fffffu) | 0x3f800000u), acc[t], 0, 0, 0);
        }
    }
    float s = (float)x;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

typedef void (*victim_fn)(float*, int);
static victim_fn VICT[11] = {victim0, victim1, victim2, victim3, victim4, victim5, victim6, victim7, victim8, victim9, victim10};

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 20, WG = 1024, N = 4096, SIDE_WG = 2048;
    hipStream_t s1, s2;
    CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreate(&s2));
    float *out, *side_out; u32x4* ops;
    CHECK(hipMalloc(&out, (size_t)WG * 512 * 4)); CHECK(hipMalloc(&side_out, (size_t)SIDE_WG * 256 * 4)); CHECK(hipMalloc(&ops, 1024 * 16));
    std::vector<uint32_t> h(4096);
    uint32_t r = 12345u;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (r & 0x807fffffu) | 0x3f000000u; v = (v & 0xffff0000u) | ((v >> 16) & 0xbfffu) | 0x3f00u; }
    CHECK(hipMemcpy(ops, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    std::vector<float> host((size_t)WG * 512);
    const char* SIDE[5] = {"nothing", "bf16 MFMA + ds_read_b128", "bf16 MFMA only", "ds_read_b128 only", "fp32 MFMA + ds_read_b128"};
    printf("%d launches of 1024 workgroups x 256 threads per cell; a launch counts as wrong when any lane's sum differs from n x the exact result\n", launches);
    for (int m = 0; m < 11; ++m) {
        printf("%-30s expect (%g, %g):", WHAT[m], EXPECT[m][0], EXPECT[m][1]);
        float ref0 = 0.f, ref1 = 0.f;
        for (int sd = 0; sd < 5; ++sd) {
            int bad = 0; uint64_t lanes = 0; int halves = 0; float sample = 0.f;
            for (int l = 0; l < launches; ++l) {
                if (sd == 1) hipLaunchKernelGGL((side_kernel<1, true>), dim3(SIDE_WG), dim3(256), 0, s2, ops, side_out, 600);
                if (sd == 2) hipLaunchKernelGGL((side_kernel<1, false>), dim3(SIDE_WG), dim3(256), 0, s2, ops, side_out, 600);
                if (sd == 3) hipLaunchKernelGGL((side_kernel<0, true>), dim3(SIDE_WG), dim3(256), 0, s2, ops, side_out, 2400);
                if (sd == 4) hipLaunchKernelGGL((side_kernel<2, true>), dim3(SIDE_WG), dim3(256), 0, s2, ops, side_out, 300);
                hipLaunchKernelGGL(VICT[m], dim3(WG), dim3(256), 0, s1, out, N);
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(host.data(), out, host.size() * 4, hipMemcpyDeviceToHost));
                bool any = false;
                if (sd == 0 && l == 0) { ref0 = host[0]; ref1 = host[1]; if (m < 8 && (ref0 != EXPECT[m][0] * N || ref1 != EXPECT[m][1] * N)) printf(" [idle result (%g, %g) per iteration differs from the expectation]", ref0 / N, ref1 / N); }
                for (size_t i = 0; i < host.size(); ++i)
                    if (host[i] != ((i & 1) ? ref1 : ref0)) { any = true; lanes |= 1ull << ((i >> 1) & 63); halves |= 1 << (i & 1); sample = host[i] / N; }
                bad += any;
            }
            printf("  | %s: %d", SIDE[sd], bad);
            if (bad) printf(" (lanes %016llx, halves %s, e.g. %g per iteration)", (unsigned long long)lanes, halves == 1 ? "low" : halves == 2 ? "high" : "both", sample);
        }
        printf("\n");
    }
    return 0;
}
