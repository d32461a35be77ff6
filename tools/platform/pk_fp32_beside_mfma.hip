// Victim kernel for tools/platform/pk_fp32_beside_mfma.py: a chain of packed-FP32 FMAs (v_pk_fma_f32) and, beside it, the same
// chain in plain v_fma_f32, on lane-dependent but launch-independent inputs.  4 waves per workgroup, ~100 VGPRs requested so
// that a workgroup takes the wave slot left beside two 174-VGPR convolution workgroups per CU.  Both results are written out;
// the driver compares them with a launch that ran alone.
#include <hip/hip_runtime.h>
#ifndef MODE
#define MODE 0
#endif
typedef float v2f __attribute__((ext_vector_type(2)));

extern "C" __global__ __launch_bounds__(256, 3) void pk_victim(float* __restrict__ out, int iters, int pad_regs) {
    const int t = threadIdx.x, g = blockIdx.x * 256 + t;
    v2f acc = {1.0f + 0.001f * t, 2.0f - 0.001f * t};
    float p0 = acc.x, p1 = acc.y;
    const v2f a = {0.999f, 1.001f};
    const v2f b = {0.0005f * (t & 15), -0.0003f * (t & 7)};
    for (int i = 0; i < iters; ++i) {
#if MODE == 0      // plain packed FMA
        asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(p0) : "v"(a.x), "v"(b.x));
        asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(p1) : "v"(a.y), "v"(b.y));
#elif MODE == 1    // the form the compiler emits for scalar * vector: the low half of src0 feeds both halves
        asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(p0) : "v"(a.x), "v"(b.x));
        asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(p1) : "v"(a.x), "v"(b.y));
#elif MODE == 3    // as MODE 1, the packed operand coming from a broadcast ds_read_b128 (every lane the same LDS address)
        {
            __shared__ float4 s_g[16];
            if (i == 0) { if (t < 16) s_g[t] = make_float4(0.0005f * t, -0.0003f * t, 0.0002f * t, 0.0001f * t); __syncthreads(); }
            float4 g4;
            const unsigned addr = (unsigned)(uintptr_t)(s_g + (i & 15));
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(g4) : "v"(addr) : "memory");
            const v2f g = {g4.x, g4.y};
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(g));
            asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(p0) : "v"(a.x), "v"(g4.x));
            asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(p1) : "v"(a.x), "v"(g4.y));
        }
#else              // packed multiply + packed add with op_sel, as in the skinning epilogue
        v2f m = acc;
        asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(m) : "v"(a));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc) : "v"(m), "v"(b));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(a.x));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(p0) : "v"(b.x));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(p1) : "v"(a.x));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(p1) : "v"(b.y));
#endif
    }
    out[g * 4 + 0] = acc.x; out[g * 4 + 1] = acc.y; out[g * 4 + 2] = p0; out[g * 4 + 3] = p1;
}

extern "C" int pk_victim_launch(float* out, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(pk_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 0);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
