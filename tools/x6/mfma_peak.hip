// Sustained v_mfma_f32_32x32x16_bf16 rate on random operands (the clock the chip holds under this instruction):
// hipcc --offload-arch=gfx950 -O3 tools/x6/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const u32x4* in, float* out, int iters) {
    u32x4 ra[6], rb[6];
    for (int i = 0; i < 6; ++i) { ra[i] = in[(threadIdx.x + 64 * i) & 1023]; rb[i] = in[(threadIdx.x * 3 + 17 * i) & 1023]; }
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[(q + t) % 6]),
                                                                 __builtin_bit_cast(bf16x8, rb[(q * 2 + t) % 6]), acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    std::vector<unsigned> h(4096);
    for (auto& v : h) { unsigned short a = 0x3f00 | (rand() & 0xff), b = 0x3f00 | (rand() & 0xff); v = a | (b << 16); }   // bf16 in [0.5, 1)
    u32x4* in; float* out; hipMalloc(&in, 16384); hipMalloc(&out, 4 * 256 * 4096);
    hipMemcpy(in, h.data(), 16384, hipMemcpyHostToDevice);
    for (int wgs : {256, 512, 1024}) {
        const int iters = 4000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, in, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 5.0 * wgs * 4 /*waves*/ * iters * 24.0 * 32 * 32 * 16 * 2;
        printf("workgroups %4d (%d per CU): %.1f ms, %.0f TFLOP/s bf16 = %.1f fp32-equivalent (x6)\n", wgs, wgs / 256, ms, flops / ms / 1e9, flops / ms / 1e9 / 6);
    }
    return 0;
}
