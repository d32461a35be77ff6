// 32x32x16 vs 16x16x32 bf16 MFMA at the same 64 x 64 output tile per wave, operands in registers and re-read from LDS every
// iteration (ds_read_b128), random full-range operands: the FLOP/s each shape sustains at the clock the chip holds under it
// (MI355X_MICROARCH.md, DVFS give-back item 7).   hipcc --offload-arch=gfx950 -O3 tools/x6/mfma_shapes.hip -o /tmp/ms && /tmp/ms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool LDS>
__global__ __launch_bounds__(256, 2) void k32(const u32x4* in, float* out, int iters) {
    __shared__ u32x4 s[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) s[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x & 63;
    u32x4 a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = s[(l + 64 * i) & 1023]; b[i] = s[(l * 3 + 17 * i + 128) & 1023]; }
    f32x16 acc[2][2] = {};
    for (int it = 0; it < iters; ++it) {
        if (LDS) { for (int i = 0; i < 2; ++i) { a[i] = s[(l + 64 * i + it * 5) & 1023]; b[i] = s[(l + 64 * i + it * 7 + 512) & 1023]; } }
#pragma unroll
        for (int q = 0; q < 2; ++q)        // two 16-k steps = 32 k
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) r += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <bool LDS>
__global__ __launch_bounds__(256, 2) void k16(const u32x4* in, float* out, int iters) {
    __shared__ u32x4 s[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) s[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x & 63;
    u32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = s[(l + 64 * i) & 1023]; b[i] = s[(l * 3 + 17 * i + 128) & 1023]; }
    f32x4 acc[4][4] = {};
    for (int it = 0; it < iters; ++it) {
        if (LDS) { for (int i = 0; i < 4; ++i) { a[i] = s[(l + 64 * i + it * 5) & 1023]; b[i] = s[(l + 64 * i + it * 7 + 512) & 1023]; } }
#pragma unroll
        for (int i = 0; i < 4; ++i)        // one 32-k step
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) r += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <typename K> double run(K kern, const u32x4* in, float* out, int wgs, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, in, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 6; ++r) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 6.0 * wgs * 4 * iters * (64.0 * 64 * 32 * 2) / ms / 1e9;      // TFLOP/s (one iteration = a 64 x 64 x 32 wave tile step)
}
int main() {
    std::vector<unsigned> h(4096);
    srand(3);
    for (auto& v : h) { unsigned short a = (unsigned short)(rand() & 0xffff), b = (unsigned short)(rand() & 0xffff);
        a = (a & 0x807f) | (((a >> 7) % 16 + 120) << 7); b = (b & 0x807f) | (((b >> 7) % 16 + 120) << 7); v = a | (b << 16); }   // random sign / mantissa, exponents 2^-7 .. 2^8
    u32x4* in; float* out; hipMalloc(&in, 16384); hipMalloc(&out, 4 * 256 * 2048);
    hipMemcpy(in, h.data(), 16384, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep)
        for (int wgs : {512}) {
            const int iters = 6000;
            printf("wgs %d: regs 32x32x16 %.0f | 16x16x32 %.0f || LDS 32x32x16 %.0f | 16x16x32 %.0f  TFLOP/s bf16\n", wgs,
                   run(k32<false>, in, out, wgs, iters), run(k16<false>, in, out, wgs, iters), run(k32<true>, in, out, wgs, iters),
                   run(k16<true>, in, out, wgs, iters));
        }
    return 0;
}
