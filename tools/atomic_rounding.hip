// Is the memory-side global_atomic_add_f32 an IEEE round-to-nearest-even fp32 add, and does it keep denormals?
// One lane adds n floats one after the other into one address (each add waits for the previous one: returning form); the
// host repeats the same sequential sum in fp32.  Prints the two sums and how many of 256 independent trials differ.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void seq_atomic(const float* x, int n, float* out) {
    float* cell = out + blockIdx.x;
    const float* xs = x + (size_t)blockIdx.x * n;
    if (threadIdx.x == 0)
        for (int i = 0; i < n; ++i) (void)atomicAdd(cell, xs[i]);      // returning form: serialised in issue order
}
__global__ void seq_atomic_noret(const float* x, int n, float* out) {
    float* cell = out + blockIdx.x;
    const float* xs = x + (size_t)blockIdx.x * n;
    if (threadIdx.x == 0)
        for (int i = 0; i < n; ++i) { __hip_atomic_fetch_add(cell, xs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_s_waitcnt(0); }
}
int main() {
    const int trials = 256, n = 512;
    std::vector<float> h((size_t)trials * n);
    srand(1);
    for (auto& v : h) v = ((rand() / (float)RAND_MAX) - 0.5f) * (1.0f + (rand() % 1000));
    float *dx, *dout;
    hipMalloc(&dx, h.size() * 4); hipMalloc(&dout, trials * 4);
    hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int form = 0; form < 2; ++form) {
        hipMemset(dout, 0, trials * 4);
        if (form == 0) hipLaunchKernelGGL(seq_atomic, dim3(trials), dim3(64), 0, 0, dx, n, dout);
        else hipLaunchKernelGGL(seq_atomic_noret, dim3(trials), dim3(64), 0, 0, dx, n, dout);
        std::vector<float> got(trials);
        hipMemcpy(got.data(), dout, trials * 4, hipMemcpyDeviceToHost);
        int diff = 0; double worst = 0;
        for (int t = 0; t < trials; ++t) {
            volatile float s = 0.f;
            for (int i = 0; i < n; ++i) s = s + h[(size_t)t * n + i];
            if (memcmp((const void*)&s, &got[t], 4) != 0) { ++diff; double e = fabs((double)s - got[t]) / fabs((double)s); if (e > worst) worst = e; }
        }
        printf("form %d (%s): %d of %d sequential sums differ from the host's fp32 RNE sum (worst rel %.2e)\n", form,
               form ? "no-return + wait" : "returning", diff, trials, worst);
    }
    // denormals: 1e-40 added 1000 times
    float tiny = 1e-40f; std::vector<float> ht(1024, tiny);
    hipMemcpy(dx, ht.data(), 1024 * 4, hipMemcpyHostToDevice);
    hipMemset(dout, 0, 4);
    hipLaunchKernelGGL(seq_atomic, dim3(1), dim3(64), 0, 0, dx, 1000, dout);
    float g; hipMemcpy(&g, dout, 4, hipMemcpyDeviceToHost);
    printf("denormal sum: device %.6e, exact %.6e\n", g, 1000 * 1e-40);
    return 0;
}
