// Shared helpers for the DSF HIP kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dsf_hip.h"

#define DSF_WAVE 64

#define DSF_CHECK_ARG(cond) \
    do { if (!(cond)) return DSF_ERR_INVALID_ARG; } while (0)

static inline int dsf_launch_status() {
    return hipGetLastError() == hipSuccess ? DSF_OK : DSF_ERR_LAUNCH;
}

// Zero fill by a kernel (api.hip).  NOT hipMemsetAsync: a memset node captured into a HIP graph does not replay
// correctly on ROCm 7.2 (tools/graph_memset.py: the second replay leaves inf/garbage for sizes between 256 B and
// ~300 KB), and every launcher here must be capturable (train_step.GraphedStep).  Returns hipSuccess or an error.
hipError_t dsf_zero_async(void* ptr, size_t bytes, hipStream_t stream);

// Deterministic mode (dsf_set_deterministic(1) / DSF_DETERMINISTIC=1): every backward accumulation that many
// lanes add into gives bit-identical results run to run.  Host launchers read the flag with dsf_deterministic().
int dsf_deterministic();

// Accumulator cells of the backward kernels.  Acc<false>: float atomics (fast; the order of the additions, hence the last
// bits of the sum, vary from run to run).  Acc<true>: 64-bit fixed point with 2^-40 resolution (range +-8.3e6): integer
// addition is associative, so the sum does not depend on the order -- the deterministic mode's "segmented reduce".
template <bool DET> struct Acc;
template <> struct Acc<false> {
    typedef float T;
    static __device__ __forceinline__ void add(T* p, float v) { atomicAdd(p, v); }
    static __device__ __forceinline__ float get(T v) { return v; }
};
template <> struct Acc<true> {
    typedef long long T;
    static __device__ __forceinline__ void add(T* p, float v) {
        atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double2ll_rn((double)v * 1099511627776.0));
    }
    static __device__ __forceinline__ float get(T v) { return (float)((double)v * (1.0 / 1099511627776.0)); }
};

// wave-wide sum via DPP-lowered shuffles (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
