// Synthetic victims and aggressors for the co-residency bisect (tools/platform/war_bisect.py).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/platform/war_kernels.hip -o tools/platform/_war/libwar.so
// Victims: hand-written instruction pairs in inline assembly; every lane accumulates n times a constant that is exact in
// fp32, so a single wrong operand read shows up as a wrong integer in that lane.
// Aggressors: loops with one ingredient of the conv_x6 kernels each (vector ALU, matrix + vector ALU, matrix + LDS, LDS + memory),
// sized like them (256 threads, ~170 registers by launch bound) so that they share a SIMD with the victim.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------- victims
// Every victim: acc (v22, v23) += 1.0 * src (v20, v21) n times with src = a; behind the accumulating instruction src is
// overwritten with b.  Correct result: n * a in both halves.  A lane that reads b instead lands on a different integer.
#define VICTIM(NAME, ACCUM, GAP)                                                                                          \
    __global__ __launch_bounds__(256, 3) void NAME(float* __restrict__ out, int n, float a, float b) {                      \
        __shared__ float lds[64];                                                                                         \
        if (threadIdx.x < 64) lds[threadIdx.x] = b;                                                                       \
        __syncthreads();                                                                                                  \
        float o0, o1;                                                                                                     \
        asm volatile(                                                                                                     \
            "v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 0\n s_mov_b32 s20, %[n]\n"        \
            "1:\n"                                                                                                        \
            "v_mov_b32 v20, %[a]\n v_mov_b32 v21, %[a]\n ds_read_b96 v[28:30], v26\n s_nop 4\n"                             \
            ACCUM GAP                                                                                                     \
            "s_nop 4\n s_waitcnt lgkmcnt(0)\n"                                                                            \
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"                                            \
            "s_nop 4\n v_mov_b32 %[o0], v22\n v_mov_b32 %[o1], v23\n"                                                      \
            : [o0] "=v"(o0), [o1] "=v"(o1)                                                                                \
            : [a] "v"(a), [b] "v"(b), [n] "s"(n)                                                                          \
            : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v28", "v29", "v30", "s20", "scc", "memory");                \
        out[(size_t)blockIdx.x * 512 + threadIdx.x * 2] = o0;                                                             \
        out[(size_t)blockIdx.x * 512 + threadIdx.x * 2 + 1] = o1;                                                         \
    }
#define PKFMA "v_pk_fma_f32 v[22:23], v[24:25], v[20:21], v[22:23]\n"
#define FMA2 "v_fma_f32 v22, v24, v20, v22\n v_fma_f32 v23, v25, v21, v23\n"
// 0: packed FMA, source overwritten by the NEXT instruction (low half first)
VICTIM(victim0, PKFMA, "v_mov_b32 v20, %[b]\n v_mov_b32 v21, %[b]\n")
// 1: one idle state between
VICTIM(victim1, PKFMA, "s_nop 0\n v_mov_b32 v20, %[b]\n v_mov_b32 v21, %[b]\n")
// 2: control: plain FMAs
VICTIM(victim2, FMA2, "v_mov_b32 v20, %[b]\n v_mov_b32 v21, %[b]\n")
// 3: the real kernel's shape: an LDS read issued before the packed op, waited for behind it, then the overwrite from the LDS data
VICTIM(victim3, PKFMA, "s_waitcnt lgkmcnt(0)\n v_mov_b32 v20, v29\n v_mov_b32 v21, v30\n")
// 4: high half overwritten first
VICTIM(victim4, PKFMA, "v_mov_b32 v21, %[b]\n v_mov_b32 v20, %[b]\n")
// 5: an independent vector instruction between the packed op and the overwrite
VICTIM(victim5, PKFMA, "v_mov_b32 v26, 0\n v_mov_b32 v20, %[b]\n v_mov_b32 v21, %[b]\n")

// ---------------------------------------------------------------------------------------------- aggressors
// 0: vector ALU only: 16 independent FMA chains
__global__ __launch_bounds__(256, 2) void agg_valu(float* __restrict__ out, int iters) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], 0.999f, 0.001f * i);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// 1: matrix + vector ALU: per MFMA a handful of conversions / subtractions (the operand split of conv_x6's loaders)
__global__ __launch_bounds__(256, 2) void agg_mfma_valu(const u32x4* __restrict__ in, float* __restrict__ out, int iters) {
    u32x4 ra = in[threadIdx.x], rb = in[256 + threadIdx.x];
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 0.37f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra), __builtin_bit_cast(bf16x8, rb), acc[t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {                        // split two floats: h = bf16(x), m = bf16(x - h)
                const float x = f[(t * 2 + i) & 7];
                const __bf16 h = (__bf16)x;
                const float r = x - (float)h;
                const __bf16 m = (__bf16)r;
                f[(t * 2 + i) & 7] = r * 1.0009765625f + (float)m;
                ra[i] ^= (uint32_t)__builtin_bit_cast(unsigned short, h) & 1u;
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[t][e];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// 2: matrix + LDS: per MFMA one ds_read_b128 fragment read and a ds_write_b64 (conv_x6's tile traffic), no global memory
__global__ __launch_bounds__(256, 2) void agg_mfma_lds(const u32x4* __restrict__ in, float* __restrict__ out, int iters) {
    __shared__ u32x4 tile[2048];                                 // 32 KB
    for (int i = threadIdx.x; i < 2048; i += 256) tile[i] = in[i & 1023];
    __syncthreads();
    u32x4 ra = in[threadIdx.x];
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32x4 rb = tile[(threadIdx.x + 64 * t + it * 7) & 2047];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra), __builtin_bit_cast(bf16x8, rb), acc[t], 0, 0, 0);
            uint2 w; w.x = rb[0] + it; w.y = rb[1];
            *(uint2*)&tile[(threadIdx.x * 3 + t * 257 + it) & 2047] = w;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// 3: LDS + global memory + vector ALU, no matrix instructions: a staging loop (16-byte loads, split, ds_write_b64, ds_read_b128)
__global__ __launch_bounds__(256, 2) void agg_stage(const u32x4* __restrict__ in, size_t n16, float* __restrict__ out, int iters) {
    __shared__ u32x4 tile[2048];
    float s = 0.f;
    size_t p = ((size_t)blockIdx.x * 256 + threadIdx.x) % n16;
    for (int it = 0; it < iters; ++it) {
        const u32x4 v = in[p];
        p = (p + (size_t)gridDim.x * 256) % n16;
        uint2 w; w.x = v[0] ^ v[2]; w.y = v[1] + v[3];
        *(uint2*)&tile[(threadIdx.x * 5 + it) & 2047] = w;
        const u32x4 r = tile[(threadIdx.x + it * 64) & 2047];
        s += __builtin_bit_cast(float, (r[0] & 0x007fffffu) | 0x3f800000u) * 0.5f;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// 4 / 5 / 6: agg_mfma_lds taken apart: matrix + LDS reads only, matrix + LDS writes only, LDS reads + writes without matrix ops
template <bool MFMA, bool RD, bool WR>
__global__ __launch_bounds__(256, 2) void agg_parts(const u32x4* __restrict__ in, float* __restrict__ out, int iters) {
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
            if (RD) { rb = tile[(threadIdx.x + 64 * t + it * 7) & 2047]; x ^= rb[3]; }
            if (MFMA) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra), __builtin_bit_cast(bf16x8, rb), acc[t], 0, 0, 0);
            if (WR) { uint2 w; w.x = ra[0] + it; w.y = ra[1] ^ t; *(uint2*)&tile[(threadIdx.x * 3 + t * 257 + it) & 2047] = w; }
        }
    }
    float s = (float)x;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    if (WR) s += (float)tile[threadIdx.x][0];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// 7: fp32-input MFMA (32x32x2) + LDS reads and writes
__global__ __launch_bounds__(256, 2) void agg_mfma32_lds(const u32x4* __restrict__ in, float* __restrict__ out, int iters) {
    __shared__ u32x4 tile[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) tile[i] = in[i & 1023];
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const float fa = __builtin_bit_cast(float, (in[threadIdx.x][0] & 0x007fffffu) | 0x3f800000u);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32x4 rb = tile[(threadIdx.x + 64 * t + it * 7) & 2047];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, __builtin_bit_cast(float, (rb[0] & 0x007fffffu) | 0x3f800000u), acc[t], 0, 0, 0);
            uint2 w; w.x = rb[0] + it; w.y = rb[1];
            *(uint2*)&tile[(threadIdx.x * 3 + t * 257 + it) & 2047] = w;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" {
int war_victim(int mode, float* out, int workgroups, int n, float a, float b, hipStream_t st) {
    switch (mode) {
    case 0: hipLaunchKernelGGL(victim0, dim3(workgroups), dim3(256), 0, st, out, n, a, b); break;
    case 1: hipLaunchKernelGGL(victim1, dim3(workgroups), dim3(256), 0, st, out, n, a, b); break;
    case 2: hipLaunchKernelGGL(victim2, dim3(workgroups), dim3(256), 0, st, out, n, a, b); break;
    case 3: hipLaunchKernelGGL(victim3, dim3(workgroups), dim3(256), 0, st, out, n, a, b); break;
    case 4: hipLaunchKernelGGL(victim4, dim3(workgroups), dim3(256), 0, st, out, n, a, b); break;
    case 5: hipLaunchKernelGGL(victim5, dim3(workgroups), dim3(256), 0, st, out, n, a, b); break;
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int war_aggressor(int kind, const void* in, size_t n16, float* out, int workgroups, int iters, hipStream_t st) {
    switch (kind) {
    case 0: hipLaunchKernelGGL(agg_valu, dim3(workgroups), dim3(256), 0, st, out, iters); break;
    case 1: hipLaunchKernelGGL(agg_mfma_valu, dim3(workgroups), dim3(256), 0, st, (const u32x4*)in, out, iters); break;
    case 2: hipLaunchKernelGGL(agg_mfma_lds, dim3(workgroups), dim3(256), 0, st, (const u32x4*)in, out, iters); break;
    case 3: hipLaunchKernelGGL(agg_stage, dim3(workgroups), dim3(256), 0, st, (const u32x4*)in, n16, out, iters); break;
    case 4: hipLaunchKernelGGL((agg_parts<true, true, false>), dim3(workgroups), dim3(256), 0, st, (const u32x4*)in, out, iters); break;
    case 5: hipLaunchKernelGGL((agg_parts<true, false, true>), dim3(workgroups), dim3(256), 0, st, (const u32x4*)in, out, iters); break;
    case 6: hipLaunchKernelGGL((agg_parts<false, true, true>), dim3(workgroups), dim3(256), 0, st, (const u32x4*)in, out, iters); break;
    case 7: hipLaunchKernelGGL(agg_mfma32_lds, dim3(workgroups), dim3(256), 0, st, (const u32x4*)in, out, iters); break;
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
}
