// What a layer boundary costs on MI355X, two ways (round-6 answer to "one cooperative launch per hourglass level", item 4 of the
// round-5 verdict):
//   (a) a GRID BARRIER inside one persistent kernel (W workgroups of 256 threads; every workgroup writes a little, fences, bumps an
//       agent-scope counter and spins until all W have arrived) -- what a fused Residual chain would pay between dependent layers
//       (BatchNorm statistics need every pixel: two barriers per BatchNorm, one per 3 x 3 convolution for the halo);
//   (b) a DEPENDENT KERNEL LAUNCH: a chain of tiny kernels on one stream, issued eagerly and replayed from a hipGraph -- what the
//       unfused chain pays today.
// Stand-alone:  hipcc --offload-arch=gfx950 -O2 -o /tmp/grid_barrier tools/platform/grid_barrier.hip && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void barrier_loop(unsigned* counter, float* buf, int iters, int work) {
    const int W = gridDim.x;
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < work; ++k)                                      // the "layer": every thread touches a few floats
            buf[(blockIdx.x * 256 + threadIdx.x) * 4 + (k & 3)] += 1.0f;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1) * (unsigned)W;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void tiny(float* buf, int W) {
    buf[(blockIdx.x * 256 + threadIdx.x) * 4] += 1.0f;
}

int main() {
    unsigned* counter; float* buf;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&buf, 256 * 256 * 4 * 4));
    CK(hipMemset(buf, 0, 256 * 256 * 4 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    printf("(a) grid barrier inside one kernel, %d barriers per launch, 256-thread workgroups\n", iters);
    for (int W : {1, 4, 16, 64, 256}) {
        for (int work : {1, 16}) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(counter, 0, 4, st));
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(barrier_loop, dim3(W), dim3(256), 0, st, counter, buf, iters, work);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("  W = %3d workgroups, %2d stores per thread per layer: %6.2f us per barrier\n", W, work, best * 1e3f / iters);
        }
    }
    printf("(b) chain of dependent tiny kernels on one stream\n");
    for (int W : {16, 256}) {
        const int n = 1000;
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(tiny, dim3(W), dim3(256), 0, st, buf, W);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny, dim3(W), dim3(256), 0, st, buf, W);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  eager,        W = %3d: %6.2f us per launch\n", W, ms * 1e3f / n);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny, dim3(W), dim3(256), 0, st, buf, W);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("  graph replay, W = %3d: %6.2f us per node\n", W, best * 1e3f / n);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
