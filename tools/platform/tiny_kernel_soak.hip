// Does a trivial elementwise kernel with CONSTANT inputs return the same bits every launch while other processes load the GPU?
// (round 3: three small-grid kernels of libdsf_hip.so did not, under two concurrent bench.py processes; this is the same
// arithmetic as xyz_to_uvd_kernel in a stand-alone program: no torch, no library, default hipcc flags unless given)
//   hipcc --offload-arch=gfx950 -O3 tiny_kernel_soak.hip -o tiny_kernel_soak && ./tiny_kernel_soak 20000
//   (-DNO_VCC_DIVISION: the same kernel with reciprocal multiplications instead of IEEE divisions, as a control)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Cam { float fx, fy, px, py, w, h; };

__global__ void k(const float* __restrict__ xyz, const float* __restrict__ center, const float* __restrict__ M,
                  const float* __restrict__ cube, Cam cam, int64_t total, int N, float img_size, float* __restrict__ uvd) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / N);
    const float* m = M + b * 9; const float* c = center + b * 3; const float* cb = cube + b * 3;
    float wx = xyz[i * 3], wy = xyz[i * 3 + 1], wz = xyz[i * 3 + 2];
    wx = wx * cb[0] * 0.5f + c[0]; wy = wy * cb[1] * 0.5f + c[1]; wz = wz * cb[2] * 0.5f + c[2];      // (exact: a power of two)
#ifdef NO_VCC_DIVISION          // control: v_rcp_f32 * numerator instead of the div_scale / div_fmas / div_fixup sequence (not IEEE-exact)
    const float U = wx * cam.fx * __builtin_amdgcn_rcpf(wz + 1e-8f) + cam.px, V = wy * cam.fy * __builtin_amdgcn_rcpf(wz) + cam.py;
#else
    const float U = wx * cam.fx / (wz + 1e-8f) + cam.px, V = wy * cam.fy / wz + cam.py;
#endif
    const float u = (m[0] * U + m[1] * V) + m[2], v = (m[3] * U + m[4] * V) + m[5];
    #ifdef NO_VCC_DIVISION
    const float r = __builtin_amdgcn_rcpf(img_size);
    uvd[i * 3] = u * r * 2.0f - 1.0f; uvd[i * 3 + 1] = v * r * 2.0f - 1.0f; uvd[i * 3 + 2] = (wz - c[2]) * __builtin_amdgcn_rcpf(cb[2] * 0.5f);
#else
    uvd[i * 3] = u / img_size * 2.0f - 1.0f; uvd[i * 3 + 1] = v / img_size * 2.0f - 1.0f; uvd[i * 3 + 2] = (wz - c[2]) / (cb[2] / 2.0f);
#endif
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, B = 3, N = 21, n = B * N * 3;
    std::vector<float> hx(n), hc(B * 3), hM(B * 9), hcb(B * 3), ref(n), out(n);
    srand(1);
    for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 0.6f;
    for (int b = 0; b < B; ++b) {
        hc[b * 3] = 0; hc[b * 3 + 1] = 0; hc[b * 3 + 2] = 400; hcb[b * 3] = hcb[b * 3 + 1] = hcb[b * 3 + 2] = 250;
        const float m[9] = {0.5f, 0, -90, 0, 0.5f, -60, 0, 0, 1}; memcpy(&hM[b * 9], m, sizeof(m));
    }
    float *dx, *dc, *dM, *dcb, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dc, B * 12); hipMalloc(&dM, B * 36); hipMalloc(&dcb, B * 12); hipMalloc(&dout, n * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, hc.data(), B * 12, hipMemcpyHostToDevice);
    hipMemcpy(dM, hM.data(), B * 36, hipMemcpyHostToDevice); hipMemcpy(dcb, hcb.data(), B * 12, hipMemcpyHostToDevice);
    const Cam cam = {588.03f, 587.07f, 320.f, 240.f, 640.f, 480.f};
    int bad = 0;
    for (int it = 0; it <= iters; ++it) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dx, dc, dM, dcb, cam, (int64_t)B * N, N, 128.0f, dout);
        hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost);
        if (it == 0) { ref = out; continue; }
        if (memcmp(out.data(), ref.data(), n * 4) != 0) {
            if (++bad <= 3) {
                printf("launch %d differs at:", it);
                for (int e = 0; e < n; ++e) if (memcmp(&out[e], &ref[e], 4)) printf(" %d(%g vs %g)", e, out[e], ref[e]);
                printf("\n");
            }
        }
    }
    printf("%d launches, %d with different bits\n", iters, bad);
    return 0;
}
