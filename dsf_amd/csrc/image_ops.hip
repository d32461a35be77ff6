// K8/K9/K10: image-side kernels for gfx950 -- all HBM-bound elementwise / reduce work:
// uvd<->xyz transforms (data/render_loader.py:1044-1088), crop_hand fused with the point
// image (:1190-1227), Img2pcl as an ordered stream compaction plus a key-threshold sample
// (:1121-1156), and the GFM offset-map encode / soft-argmax decode with their backward
// passes (util/generateFeature.py:14-59).  Accesses are lane-contiguous along image rows.
#include "common.h"

namespace {

__device__ __forceinline__ float grid_aligned(int i, int S) { return 2.0f * (float)i / ((float)S - 1.0f) - 1.0f; }
__device__ __forceinline__ float grid_centre(int i, int S) { return 2.0f * ((float)i + 0.5f) / (float)S - 1.0f; }

// crop-normalised uvd -> xyz (mm or cube-normalised); mi = torch.inverse(M) rows 0..1
__device__ __forceinline__ void uvd2xyz(const float* mi, const float* c, const float* cube, const dsf_camera& cam,
                                        float half_img, float a, float bb, float dn, bool normalise, float* out) {
    const float uu = (a + 1.0f) * half_img, vv = (bb + 1.0f) * half_img;
    const float d = dn * (cube[2] / 2.0f) + c[2];
    const float u = (mi[0] * uu + mi[1] * vv) + mi[2];
    const float v = (mi[3] * uu + mi[4] * vv) + mi[5];
    float x = (u - cam.px) * d / cam.fx, y = (v - cam.py) * d / cam.fy, z = d;
    if (normalise) { x = (x - c[0]) / (cube[0] / 2.0f); y = (y - c[1]) / (cube[1] / 2.0f); z = (z - c[2]) / (cube[2] / 2.0f); }
    out[0] = x; out[1] = y; out[2] = z;
}

__global__ void uvd_to_xyz_kernel(const float* __restrict__ uvd, const float* __restrict__ center,
                                  const float* __restrict__ minv, const float* __restrict__ cube, dsf_camera cam,
                                  int64_t total, int N, float half_img, int normalise, float* __restrict__ xyz) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / N);
    uvd2xyz(minv + b * 9, center + b * 3, cube + b * 3, cam, half_img, uvd[i * 3], uvd[i * 3 + 1], uvd[i * 3 + 2],
            normalise != 0, xyz + i * 3);
}

__global__ void uvd_to_xyz_bwd_kernel(const float* __restrict__ uvd, const float* __restrict__ center,
                                      const float* __restrict__ minv, const float* __restrict__ cube, dsf_camera cam,
                                      const float* __restrict__ g, int64_t total, int N, float half_img, int normalise,
                                      float* __restrict__ gu) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / N);
    const float* mi = minv + b * 9;
    const float* c = center + b * 3;
    const float* cb = cube + b * 3;
    const float uu = (uvd[i * 3] + 1.0f) * half_img, vv = (uvd[i * 3 + 1] + 1.0f) * half_img;
    const float d = uvd[i * 3 + 2] * (cb[2] / 2.0f) + c[2];
    const float u = (mi[0] * uu + mi[1] * vv) + mi[2], v = (mi[3] * uu + mi[4] * vv) + mi[5];
    float gx = g[i * 3], gy = g[i * 3 + 1], gz = g[i * 3 + 2];
    if (normalise) { gx /= (cb[0] / 2.0f); gy /= (cb[1] / 2.0f); gz /= (cb[2] / 2.0f); }
    const float g_u = gx * d / cam.fx, g_v = gy * d / cam.fy;
    const float g_d = gx * (u - cam.px) / cam.fx + gy * (v - cam.py) / cam.fy + gz;
    gu[i * 3] = (mi[0] * g_u + mi[3] * g_v) * half_img;
    gu[i * 3 + 1] = (mi[1] * g_u + mi[4] * g_v) * half_img;
    gu[i * 3 + 2] = g_d * (cb[2] / 2.0f);
}

__global__ void xyz_to_uvd_kernel(const float* __restrict__ xyz, const float* __restrict__ center,
                                  const float* __restrict__ M, const float* __restrict__ cube, dsf_camera cam,
                                  int64_t total, int N, float img_size, int world, float* __restrict__ uvd) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / N);
    const float* m = M + b * 9;
    const float* c = center + b * 3;
    const float* cb = cube + b * 3;
    float wx = xyz[i * 3], wy = xyz[i * 3 + 1], wz = xyz[i * 3 + 2];
    if (!world) { wx = wx * cb[0] / 2.0f + c[0]; wy = wy * cb[1] / 2.0f + c[1]; wz = wz * cb[2] / 2.0f + c[2]; }
    const float U = wx * cam.fx / (wz + 1e-8f) + cam.px;          // eps on u only (:1321-1322)
    const float Vv = wy * cam.fy / wz + cam.py;
    const float u = (m[0] * U + m[1] * Vv) + m[2], v = (m[3] * U + m[4] * Vv) + m[5];
    uvd[i * 3] = u / img_size * 2.0f - 1.0f;
    uvd[i * 3 + 1] = v / img_size * 2.0f - 1.0f;
    uvd[i * 3 + 2] = (wz - c[2]) / (cb[2] / 2.0f);
}

__global__ void xyz_to_uvd_bwd_kernel(const float* __restrict__ xyz, const float* __restrict__ center,
                                      const float* __restrict__ M, const float* __restrict__ cube, dsf_camera cam,
                                      const float* __restrict__ g, int64_t total, int N, float img_size, int world,
                                      float* __restrict__ gx) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / N);
    const float* m = M + b * 9;
    const float* c = center + b * 3;
    const float* cb = cube + b * 3;
    float wx = xyz[i * 3], wy = xyz[i * 3 + 1], wz = xyz[i * 3 + 2];
    if (!world) { wx = wx * cb[0] / 2.0f + c[0]; wy = wy * cb[1] / 2.0f + c[1]; wz = wz * cb[2] / 2.0f + c[2]; }
    const float gu = g[i * 3] / img_size * 2.0f, gv = g[i * 3 + 1] / img_size * 2.0f;
    const float gU = m[0] * gu + m[3] * gv, gV = m[1] * gu + m[4] * gv;
    const float ze = wz + 1e-8f;
    float gwx = gU * cam.fx / ze, gwy = gV * cam.fy / wz;
    float gwz = -gU * wx * cam.fx / (ze * ze) - gV * wy * cam.fy / (wz * wz) + g[i * 3 + 2] / (cb[2] / 2.0f);
    if (!world) { gwx = gwx * cb[0] / 2.0f; gwy = gwy * cb[1] / 2.0f; gwz = gwz * cb[2] / 2.0f; }
    gx[i * 3] = gwx; gx[i * 3 + 1] = gwy; gx[i * 3 + 2] = gwz;
}

// crop_hand: bbox of the skeleton (+offsets) in mm, strict inside test on the xyz image
__global__ __launch_bounds__(256) void crop_hand_kernel(const float* __restrict__ img, const float* __restrict__ joints,
                                                        const float* __restrict__ center, const float* __restrict__ minv,
                                                        const float* __restrict__ cube, dsf_camera cam, int J, int S,
                                                        int wg_per_sample, float oxy, float oz, float thick,
                                                        float* __restrict__ out, float* __restrict__ xyz_nl,
                                                        uint8_t* __restrict__ keep_out) {
    __shared__ float s_lo[3], s_hi[3];
    const int b = blockIdx.x / wg_per_sample, part = blockIdx.x % wg_per_sample, t = threadIdx.x;
    const float* c = center + b * 3;
    const float* cb = cube + b * 3;
    if (t < 3) {
        float lo = INFINITY, hi = -INFINITY;
        for (int j = 0; j < J; ++j) {
            const float w = joints[(b * J + j) * 3 + t] * cb[t] / 2.0f + c[t];
            lo = fminf(lo, w); hi = fmaxf(hi, w);
        }
        const float off = (t < 2) ? oxy : oz;
        lo = lo - off; hi = hi + off;
        if (t == 2) lo = lo - thick;
        s_lo[t] = lo; s_hi[t] = hi;
    }
    __syncthreads();
    const int npx = S * S;
    const float half_img = (float)S / 2.0f;
    for (int q = part * 256 + t; q < npx; q += wg_per_sample * 256) {
        const int i = q / S, j = q % S;
        const int64_t o = (int64_t)b * npx + q;
        const float dval = img[o];
        float w[3];
        uvd2xyz(minv + b * 9, c, cb, cam, half_img, grid_aligned(j, S), grid_aligned(i, S), dval, false, w);
        const bool keep = w[0] > s_lo[0] && w[0] < s_hi[0] && w[1] > s_lo[1] && w[1] < s_hi[1] && w[2] > s_lo[2] && w[2] < s_hi[2];
        out[o] = keep ? dval : 1.0f;
        if (keep_out) keep_out[o] = keep ? 1 : 0;
        if (xyz_nl) {
            xyz_nl[o * 3] = (w[0] - c[0]) / (cb[0] / 2.0f);
            xyz_nl[o * 3 + 1] = (w[1] - c[1]) / (cb[1] / 2.0f);
            xyz_nl[o * 3 + 2] = (w[2] - c[2]) / (cb[2] / 2.0f);
        }
    }
}

// ---- Img2pcl ------------------------------------------------------------------------------------
// one workgroup per sample; every lane owns a contiguous run of pixels so that the compaction
// keeps scan order (the order torch.masked_select produces).
__device__ int block_exclusive_scan(int v, int* s_scan, int& total) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) s_scan[wave] = x;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_scan[w];
    total = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    __syncthreads();
    return base + x - v;
}

__global__ __launch_bounds__(256) void img2pcl_kernel(const float* __restrict__ img, const float* __restrict__ center,
                                                      const float* __restrict__ minv, const float* __restrict__ cube,
                                                      dsf_camera cam, const uint32_t* __restrict__ keys, int S,
                                                      int n_sample, float* __restrict__ pcl, int32_t* __restrict__ counts,
                                                      uint32_t* __restrict__ ws) {
    __shared__ int s_scan[4];
    __shared__ int s_hist[256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_need;
    const int b = blockIdx.x, t = threadIdx.x;
    const int npx = S * S;
    const int run = (npx + 255) / 256;
    const float* im = img + (int64_t)b * npx;
    uint32_t* list = ws + (int64_t)b * 2 * npx;           // valid pixel ids, scan order
    uint32_t* sel = list + npx;                            // sampled subset (pixel ids), scan order
    const int q0 = t * run, q1 = min(npx, q0 + run);

    int cnt = 0;
    for (int q = q0; q < q1; ++q) cnt += (im[q] <= 0.99f) ? 1 : 0;
    int total;
    int off = block_exclusive_scan(cnt, s_scan, total);
    for (int q = q0; q < q1; ++q) if (im[q] <= 0.99f) list[off++] = (uint32_t)q;
    if (t == 0) counts[b] = total;
    __syncthreads();

    float* out = pcl + (int64_t)b * n_sample * 3;
    if (total == 0) {
        for (int e = t; e < n_sample * 3; e += 256) out[e] = 0.f;
        return;
    }
    const int mult = n_sample / total;
    const int rem = n_sample - mult * total;
    const float half_img = (float)S / 2.0f;
    const float* mi = minv + b * 9;
    const float* c = center + b * 3;
    const float* cb = cube + b * 3;

    // whole copies (temp.repeat(mult, 1), :1145)
    for (int e = t; e < mult * total; e += 256) {
        const int q = (int)list[e % total];
        uvd2xyz(mi, c, cb, cam, half_img, grid_aligned(q % S, S), grid_aligned(q / S, S), im[q], true, out + (int64_t)e * 3);
    }
    if (rem == 0) return;

    // the `rem` valid pixels with the smallest keys: 4-pass radix select of the rem-th key
    const uint32_t* kb = keys + (int64_t)b * npx;
    if (t == 0) { s_prefix = 0u; s_need = rem; }
    __syncthreads();
    for (int pass = 3; pass >= 0; --pass) {
        s_hist[t] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        const uint32_t hi_mask = (pass == 3) ? 0u : (0xFFFFFFFFu << ((pass + 1) * 8));
        for (int e = t; e < total; e += 256) {
            const uint32_t k = kb[list[e]];
            if ((k & hi_mask) == prefix) atomicAdd(&s_hist[(k >> (pass * 8)) & 0xFF], 1);
        }
        __syncthreads();
        if (t == 0) {
            int need = s_need, acc = 0, d = 0;
            for (; d < 256; ++d) { if (acc + s_hist[d] >= need) break; acc += s_hist[d]; }
            s_prefix = prefix | ((uint32_t)d << (pass * 8));
            s_need = need - acc;
        }
        __syncthreads();
    }
    const uint32_t thr = s_prefix;          // the rem-th smallest key; s_need = how many ties at thr to take
    const int ties_to_take = s_need;
    // ordered compaction of {key < thr} plus the first `ties_to_take` with key == thr
    const int lrun = (total + 255) / 256;
    const int e0 = t * lrun, e1 = min(total, e0 + lrun);
    int c_lt = 0, c_eq = 0;
    for (int e = e0; e < e1; ++e) { const uint32_t k = kb[list[e]]; c_lt += (k < thr); c_eq += (k == thr); }
    int tot_eq, tot_lt;
    const int eq_before = block_exclusive_scan(c_eq, s_scan, tot_eq);
    int taken_eq_here = min(max(ties_to_take - eq_before, 0), c_eq);
    int tot_sel;
    int pos = block_exclusive_scan(c_lt + taken_eq_here, s_scan, tot_sel);
    (void)tot_lt;
    int eq_seen = eq_before;
    for (int e = e0; e < e1; ++e) {
        const uint32_t k = kb[list[e]];
        bool take = k < thr;
        if (k == thr) { take = eq_seen < ties_to_take; ++eq_seen; }
        if (take) sel[pos++] = list[e];
    }
    __syncthreads();
    for (int e = t; e < rem; e += 256) {
        const int q = (int)sel[e];
        uvd2xyz(mi, c, cb, cam, half_img, grid_aligned(q % S, S), grid_aligned(q / S, S), im[q], true,
                out + (int64_t)(mult * total + e) * 3);
    }
}

// ---- GFM ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void joint2offset_fwd_kernel(const float* __restrict__ joints,
                                                               const float* __restrict__ img, int J, int H, int S,
                                                               float ks, float* __restrict__ maps, int64_t sb,
                                                               int64_t sc, int64_t sq) {
    extern __shared__ float s_j[];
    const int b = blockIdx.y, t = threadIdx.x;
    for (int e = t; e < J * 3; e += 256) s_j[e] = joints[b * J * 3 + e];
    __syncthreads();
    const int q = blockIdx.x * 256 + t;
    if (q >= S * S) return;
    const int y = q / S, x = q % S, step = H / S;
    const float dep = img[((int64_t)b * H + y * step) * H + x * step];       // F.interpolate nearest
    const float cu = grid_centre(x, S), cv = grid_centre(y, S);
    const bool fg = dep < 0.99f;
    float* mb = maps + (int64_t)b * sb + q * sq;
    const int64_t plane = sc;
    for (int j = 0; j < J; ++j) {
        const float ox = s_j[j * 3] - cu, oy = s_j[j * 3 + 1] - cv, oz = s_j[j * 3 + 2] - dep;
        const float dist = sqrtf(ox * ox + oy * oy + oz * oz + 1e-8f);
        const float heat = (ks - dist) / ks;
        const float m = (heat >= 0.f && fg) ? 1.f : 0.f;
        mb[(j * 3 + 0) * plane] = (ox / dist) * m;
        mb[(j * 3 + 1) * plane] = (oy / dist) * m;
        mb[(j * 3 + 2) * plane] = (oz / dist) * m;
        mb[(3 * J + j) * plane] = heat * m;
    }
}

// channels-last twin: one thread per (pixel, joint) so that consecutive lanes write consecutive 12-byte groups of one
// pixel's 4J-float row (one thread per pixel with 4J strided stores touched 168 cache lines per store instruction).
// Same arithmetic per element as joint2offset_fwd_kernel.
__global__ __launch_bounds__(256) void joint2offset_fwd_cl_kernel(const float* __restrict__ joints,
                                                                  const float* __restrict__ img, int J, int H, int S,
                                                                  float ks, float* __restrict__ maps, int64_t sb,
                                                                  int64_t sq) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= S * S * J) return;
    const int q = idx / J, j = idx - q * J;
    const int y = q / S, x = q % S, step = H / S;
    const float dep = img[((int64_t)b * H + y * step) * H + x * step];       // F.interpolate nearest
    const float cu = grid_centre(x, S), cv = grid_centre(y, S);
    const bool fg = dep < 0.99f;
    const float* jp = joints + ((int64_t)b * J + j) * 3;
    const float ox = jp[0] - cu, oy = jp[1] - cv, oz = jp[2] - dep;
    const float dist = sqrtf(ox * ox + oy * oy + oz * oz + 1e-8f);
    const float heat = (ks - dist) / ks;
    const float m = (heat >= 0.f && fg) ? 1.f : 0.f;
    float* mb = maps + (int64_t)b * sb + (int64_t)q * sq;
    mb[j * 3 + 0] = (ox / dist) * m;
    mb[j * 3 + 1] = (oy / dist) * m;
    mb[j * 3 + 2] = (oz / dist) * m;
    mb[3 * J + j] = heat * m;
}

__global__ __launch_bounds__(256) void joint2offset_bwd_kernel(const float* __restrict__ joints,
                                                               const float* __restrict__ img,
                                                               const float* __restrict__ gmaps, int J, int H, int S,
                                                               float ks, float* __restrict__ gj, int64_t sb, int64_t sc,
                                                               int64_t sq) {
    __shared__ float s_red[4][3];
    const int j = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float jx = joints[(b * J + j) * 3], jy = joints[(b * J + j) * 3 + 1], jz = joints[(b * J + j) * 3 + 2];
    const int64_t plane = sc;
    const float* gb = gmaps + (int64_t)b * sb;
    const int step = H / S;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int q = t; q < S * S; q += 256) {
        const int y = q / S, x = q % S;
        const float dep = img[((int64_t)b * H + y * step) * H + x * step];
        const float ox = jx - grid_centre(x, S), oy = jy - grid_centre(y, S), oz = jz - dep;
        const float dist = sqrtf(ox * ox + oy * oy + oz * oz + 1e-8f);
        const float heat = (ks - dist) / ks;
        if (!(heat >= 0.f && dep < 0.99f)) continue;
        const float ux = ox / dist, uy = oy / dist, uz = oz / dist;
        const float* gq = gb + q * sq;
        const float g0 = gq[(j * 3) * plane], g1 = gq[(j * 3 + 1) * plane], g2 = gq[(j * 3 + 2) * plane];
        const float gh = gq[(3 * J + j) * plane];
        // d(unit)/d(off) = (I - u u^T)/dist ; d(heat)/d(off) = -u/ks ; dist = sqrt(|off|^2 + eps)
        const float gu = g0 * ux + g1 * uy + g2 * uz;
        const float k = gu + gh * dist / ks;
        a0 += (g0 - ux * k) / dist; a1 += (g1 - uy * k) / dist; a2 += (g2 - uz * k) / dist;
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) { s_red[wave][0] = a0; s_red[wave][1] = a1; s_red[wave][2] = a2; }
    __syncthreads();
    if (t < 3) gj[(b * J + j) * 3 + t] = s_red[0][t] + s_red[1][t] + s_red[2][t] + s_red[3][t];
}

__device__ __forceinline__ float block_max(float v, float* s) {
    const int t = threadIdx.x;
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((t & 63) == 0) s[t >> 6] = v;
    __syncthreads();
    v = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    __syncthreads();
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* s) {
    const int t = threadIdx.x;
    v = wave_sum(v);
    if ((t & 63) == 0) s[t >> 6] = v;
    __syncthreads();
    v = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    return v;
}

// (sb, sc, sq): batch / channel / pixel strides of the maps in floats -- NCHW (4 J S S, S S, 1) or channels-last (4 J S S, 1, 4 J):
// the network's maps are channels-last, and reading them in place saves the layout copy in front of every decode (and behind
// every backward: the gradient is written in the maps' own layout)
__global__ __launch_bounds__(256) void offset2joint_fwd_kernel(const float* __restrict__ maps,
                                                               const float* __restrict__ depth, int J, int H, int S,
                                                               float ks, float scale, float* __restrict__ joints,
                                                               float* __restrict__ stats, int64_t sb, int64_t sc, int64_t sq) {
    __shared__ float s_red[4];
    const int j = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const float* mb = maps + (int64_t)b * sb;
    const int step = H / S;
    float mx = -INFINITY;
    for (int q = t; q < S * S; q += 256) {
        const int y = q / S, x = q % S;
        const float dep = depth[((int64_t)b * H + y * step) * H + x * step];
        const float hm = (dep < 0.99f) ? mb[(3 * J + j) * sc + q * sq] : 0.f;
        mx = fmaxf(mx, hm * scale);
    }
    mx = block_max(mx, s_red);
    float den = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int q = t; q < S * S; q += 256) {
        const int y = q / S, x = q % S;
        const float dep = depth[((int64_t)b * H + y * step) * H + x * step];
        const float m = (dep < 0.99f) ? 1.f : 0.f;
        const float hm = mb[(3 * J + j) * sc + q * sq] * m;
        const float e = expf(hm * scale - mx);
        const float dist = ks - hm * ks;
        den += e;
        a0 += (mb[(j * 3) * sc + q * sq] * m * dist + grid_centre(x, S)) * e;
        a1 += (mb[(j * 3 + 1) * sc + q * sq] * m * dist + grid_centre(y, S)) * e;
        a2 += (mb[(j * 3 + 2) * sc + q * sq] * m * dist + dep) * e;
    }
    den = block_sum(den, s_red); a0 = block_sum(a0, s_red); a1 = block_sum(a1, s_red); a2 = block_sum(a2, s_red);
    if (t == 0) {
        float* o = joints + (b * J + j) * 3;
        o[0] = a0 / den; o[1] = a1 / den; o[2] = a2 / den;
        stats[(b * J + j) * 2] = mx; stats[(b * J + j) * 2 + 1] = den;
    }
}

__global__ __launch_bounds__(256) void offset2joint_bwd_kernel(const float* __restrict__ maps,
                                                               const float* __restrict__ depth,
                                                               const float* __restrict__ joints,
                                                               const float* __restrict__ stats,
                                                               const float* __restrict__ gj, int J, int H, int S,
                                                               float ks, float scale, float* __restrict__ gmaps, int64_t sb, int64_t sc,
                                                               int64_t sq) {
    const int j = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const float* mb = maps + (int64_t)b * sb;
    float* gb = gmaps + (int64_t)b * sb;
    const int step = H / S;
    const float mx = stats[(b * J + j) * 2], den = stats[(b * J + j) * 2 + 1];
    const float g0 = gj[(b * J + j) * 3], g1 = gj[(b * J + j) * 3 + 1], g2 = gj[(b * J + j) * 3 + 2];
    const float* jo = joints + (b * J + j) * 3;
    const float gdotj = g0 * jo[0] + g1 * jo[1] + g2 * jo[2];
    for (int q = t; q < S * S; q += 256) {
        const int y = q / S, x = q % S;
        const float dep = depth[((int64_t)b * H + y * step) * H + x * step];
        const float m = (dep < 0.99f) ? 1.f : 0.f;
        const float hm = mb[(3 * J + j) * sc + q * sq] * m;
        const float w = expf(hm * scale - mx) / den;
        const float dist = ks - hm * ks;
        const float u0 = mb[(j * 3) * sc + q * sq] * m, u1 = mb[(j * 3 + 1) * sc + q * sq] * m, u2 = mb[(j * 3 + 2) * sc + q * sq] * m;
        const float v0 = u0 * dist + grid_centre(x, S), v1 = u1 * dist + grid_centre(y, S), v2 = u2 * dist + dep;
        const float gw = g0 * v0 + g1 * v1 + g2 * v2;
        const float ga = w * (gw - gdotj);                      // softmax backward
        const float gdist = w * (g0 * u0 + g1 * u1 + g2 * u2);
        gb[(j * 3) * sc + q * sq] = w * g0 * dist * m;
        gb[(j * 3 + 1) * sc + q * sq] = w * g1 * dist * m;
        gb[(j * 3 + 2) * sc + q * sq] = w * g2 * dist * m;
        gb[(3 * J + j) * sc + q * sq] = (ga * scale - gdist * ks) * m;
    }
}

// ---- soft-argmax decode of a CHANNELS-LAST map (round 6) --------------------------------------------------------------------------
// offset2joint_fwd / bwd_kernel give a workgroup one (sample, joint) and walk its pixels: on a channels-last map (the layout the
// network's heads write: pixel stride 4 J floats) every lane then touches its own 336-byte record -- 84 us forward, 125 us backward
// at B = 32 against 21 / 23 us on an NCHW map (plus the layout copies the strided entry points had removed).  Here a workgroup
// owns O2J_PIX consecutive pixels of a sample and ALL joints: thread = (joint t & 31, pixel lane t >> 5), so that the lanes of a
// pixel read its record contiguously (21 of 32 lanes active).  The softmax needs the per-joint maximum over all pixels first:
//   pass 1  per-chunk maxima                       -> ws[b][chunk][32]
//   pass 2  per-chunk sums with the GLOBAL maximum  -> ws2[b][chunk][32][4]   (the same expf(hm * scale - mx) per element as the
//           (sample, joint) kernel: only the order of the fp32 sums differs)
//   pass 3  fold the chunks in order (deterministic) -> joints, stats
// The backward is elementwise given (mx, den, joints, grad): one launch, the same expressions as offset2joint_bwd_kernel.
constexpr int O2J_PIX = 128;
__global__ __launch_bounds__(256) void o2j_cl_max_kernel(const float* __restrict__ maps, const float* __restrict__ depth, int J, int H,
                                                         int S, float scale, float* __restrict__ pmax) {
    __shared__ float s_m[8][32];
    const int t = threadIdx.x, j = t & 31, pl = t >> 5, chunk = blockIdx.x, b = blockIdx.y, C = 4 * J, SS = S * S, step = H / S;
    const float* mb = maps + (int64_t)b * SS * C;
    const int q1 = min(SS, (chunk + 1) * O2J_PIX);
    float mx = -INFINITY;
    if (j < J)
        for (int q = chunk * O2J_PIX + pl; q < q1; q += 8) {
            const float dep = depth[((int64_t)b * H + (q / S) * step) * H + (q % S) * step];
            const float hm = (dep < 0.99f) ? mb[(int64_t)q * C + 3 * J + j] : 0.f;
            mx = fmaxf(mx, hm * scale);
        }
    s_m[pl][j] = mx;
    __syncthreads();
    if (t < 32) {
        float v = s_m[0][t];
#pragma unroll
        for (int k = 1; k < 8; ++k) v = fmaxf(v, s_m[k][t]);
        pmax[((int64_t)b * gridDim.x + chunk) * 32 + t] = v;
    }
}
__global__ __launch_bounds__(256) void o2j_cl_sum_kernel(const float* __restrict__ maps, const float* __restrict__ depth, int J, int H,
                                                         int S, float ks, float scale, const float* __restrict__ pmax,
                                                         float* __restrict__ part) {
    __shared__ float s_a[4][8][32];
    const int t = threadIdx.x, j = t & 31, pl = t >> 5, chunk = blockIdx.x, b = blockIdx.y, C = 4 * J, SS = S * S, step = H / S;
    const int n_ch = gridDim.x;
    const float* mb = maps + (int64_t)b * SS * C;
    float mx = -INFINITY;
    for (int c = 0; c < n_ch; ++c) mx = fmaxf(mx, pmax[((int64_t)b * n_ch + c) * 32 + j]);
    const int q1 = min(SS, (chunk + 1) * O2J_PIX);
    float den = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (j < J)
        for (int q = chunk * O2J_PIX + pl; q < q1; q += 8) {
            const int y = q / S, x = q % S;
            const float dep = depth[((int64_t)b * H + y * step) * H + x * step];
            const float m = (dep < 0.99f) ? 1.f : 0.f;
            const float* rec = mb + (int64_t)q * C;
            const float hm = rec[3 * J + j] * m;
            const float e = expf(hm * scale - mx);
            const float dist = ks - hm * ks;
            den += e;
            a0 += (rec[j * 3] * m * dist + grid_centre(x, S)) * e;
            a1 += (rec[j * 3 + 1] * m * dist + grid_centre(y, S)) * e;
            a2 += (rec[j * 3 + 2] * m * dist + dep) * e;
        }
    s_a[0][pl][j] = den; s_a[1][pl][j] = a0; s_a[2][pl][j] = a1; s_a[3][pl][j] = a2;
    __syncthreads();
    if (t < 128) {
        const int w = t >> 5, jj = t & 31;
        float v = s_a[w][0][jj];
#pragma unroll
        for (int k = 1; k < 8; ++k) v += s_a[w][k][jj];
        part[(((int64_t)b * n_ch + chunk) * 32 + jj) * 4 + w] = v;
    }
}
__global__ __launch_bounds__(64) void o2j_cl_final_kernel(const float* __restrict__ pmax, const float* __restrict__ part, int J, int n_ch,
                                                          float* __restrict__ joints, float* __restrict__ stats) {
    const int b = blockIdx.x, j = threadIdx.x;
    if (j >= J) return;
    float mx = -INFINITY, den = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int c = 0; c < n_ch; ++c) {
        mx = fmaxf(mx, pmax[((int64_t)b * n_ch + c) * 32 + j]);
        const float* pp = part + (((int64_t)b * n_ch + c) * 32 + j) * 4;
        den += pp[0]; a0 += pp[1]; a1 += pp[2]; a2 += pp[3];
    }
    float* o = joints + ((int64_t)b * J + j) * 3;
    o[0] = a0 / den; o[1] = a1 / den; o[2] = a2 / den;
    stats[((int64_t)b * J + j) * 2] = mx; stats[((int64_t)b * J + j) * 2 + 1] = den;
}
__global__ __launch_bounds__(256) void o2j_cl_bwd_kernel(const float* __restrict__ maps, const float* __restrict__ depth,
                                                         const float* __restrict__ joints, const float* __restrict__ stats,
                                                         const float* __restrict__ gj, int J, int H, int S, float ks, float scale,
                                                         float* __restrict__ gmaps) {
    const int t = threadIdx.x, j = t & 31, pl = t >> 5, chunk = blockIdx.x, b = blockIdx.y, C = 4 * J, SS = S * S, step = H / S;
    if (j >= J) return;
    const float* mb = maps + (int64_t)b * SS * C;
    float* gb = gmaps + (int64_t)b * SS * C;
    const float mx = stats[((int64_t)b * J + j) * 2], den = stats[((int64_t)b * J + j) * 2 + 1];
    const float g0 = gj[((int64_t)b * J + j) * 3], g1 = gj[((int64_t)b * J + j) * 3 + 1], g2 = gj[((int64_t)b * J + j) * 3 + 2];
    const float* jo = joints + ((int64_t)b * J + j) * 3;
    const float gdotj = g0 * jo[0] + g1 * jo[1] + g2 * jo[2];
    const int q1 = min(SS, (chunk + 1) * O2J_PIX);
    for (int q = chunk * O2J_PIX + pl; q < q1; q += 8) {
        const int y = q / S, x = q % S;
        const float dep = depth[((int64_t)b * H + y * step) * H + x * step];
        const float m = (dep < 0.99f) ? 1.f : 0.f;
        const float* rec = mb + (int64_t)q * C;
        float* grec = gb + (int64_t)q * C;
        const float hm = rec[3 * J + j] * m;
        const float w = expf(hm * scale - mx) / den;
        const float dist = ks - hm * ks;
        const float u0 = rec[j * 3] * m, u1 = rec[j * 3 + 1] * m, u2 = rec[j * 3 + 2] * m;
        const float v0 = u0 * dist + grid_centre(x, S), v1 = u1 * dist + grid_centre(y, S), v2 = u2 * dist + dep;
        const float gw = g0 * v0 + g1 * v1 + g2 * v2;
        const float ga = w * (gw - gdotj);                      // softmax backward
        const float gdist = w * (g0 * u0 + g1 * u1 + g2 * u2);
        grec[j * 3] = w * g0 * dist * m;
        grec[j * 3 + 1] = w * g1 * dist * m;
        grec[j * 3 + 2] = w * g2 * dist * m;
        grec[3 * J + j] = (ga * scale - gdist * ks) * m;
    }
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int dsf_uvd_to_xyz(const float* uvd, const float* center, const float* minv, const float* cube,
                              const dsf_camera* cam, int B, int N, int img_size, int normalise, float* xyz,
                              dsf_stream_t stream) {
    DSF_CHECK_ARG(uvd && center && minv && cube && cam && xyz && B >= 0 && N >= 0 && img_size > 0);
    const int64_t total = (int64_t)B * N;
    if (total == 0) return DSF_OK;
    hipLaunchKernelGGL(uvd_to_xyz_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, uvd, center, minv,
                       cube, *cam, total, N, (float)img_size / 2.0f, normalise, xyz);
    return dsf_launch_status();
}

extern "C" int dsf_uvd_to_xyz_backward(const float* uvd, const float* center, const float* minv, const float* cube,
                                       const dsf_camera* cam, const float* grad_xyz, int B, int N, int img_size,
                                       int normalise, float* grad_uvd, dsf_stream_t stream) {
    DSF_CHECK_ARG(uvd && center && minv && cube && cam && grad_xyz && grad_uvd && B >= 0 && N >= 0 && img_size > 0);
    const int64_t total = (int64_t)B * N;
    if (total == 0) return DSF_OK;
    hipLaunchKernelGGL(uvd_to_xyz_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, uvd, center,
                       minv, cube, *cam, grad_xyz, total, N, (float)img_size / 2.0f, normalise, grad_uvd);
    return dsf_launch_status();
}

extern "C" int dsf_xyz_to_uvd(const float* xyz, const float* center, const float* M, const float* cube,
                              const dsf_camera* cam, int B, int N, int img_size, int world, float* uvd,
                              dsf_stream_t stream) {
    DSF_CHECK_ARG(xyz && center && M && cube && cam && uvd && B >= 0 && N >= 0 && img_size > 0);
    const int64_t total = (int64_t)B * N;
    if (total == 0) return DSF_OK;
    hipLaunchKernelGGL(xyz_to_uvd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, xyz, center, M,
                       cube, *cam, total, N, (float)img_size, world, uvd);
    return dsf_launch_status();
}

extern "C" int dsf_xyz_to_uvd_backward(const float* xyz, const float* center, const float* M, const float* cube,
                                       const dsf_camera* cam, const float* grad_uvd, int B, int N, int img_size,
                                       int world, float* grad_xyz, dsf_stream_t stream) {
    DSF_CHECK_ARG(xyz && center && M && cube && cam && grad_uvd && grad_xyz && B >= 0 && N >= 0 && img_size > 0);
    const int64_t total = (int64_t)B * N;
    if (total == 0) return DSF_OK;
    hipLaunchKernelGGL(xyz_to_uvd_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, xyz, center, M,
                       cube, *cam, grad_uvd, total, N, (float)img_size, world, grad_xyz);
    return dsf_launch_status();
}

extern "C" int dsf_crop_hand(const float* img, const float* joints_nl, const float* center, const float* minv,
                             const float* cube, const dsf_camera* cam, int B, int J, int S, float offset_xy,
                             float offset_z, float thickness, float* img_hand, float* xyz_nl, uint8_t* keep,
                             dsf_stream_t stream) {
    DSF_CHECK_ARG(img && joints_nl && center && minv && cube && cam && img_hand && B >= 0 && J > 0 && S > 1);
    if (B == 0) return DSF_OK;
    const int g = 8;
    hipLaunchKernelGGL(crop_hand_kernel, dim3(B * g), dim3(256), 0, (hipStream_t)stream, img, joints_nl, center, minv,
                       cube, *cam, J, S, g, offset_xy, offset_z, thickness, img_hand, xyz_nl, keep);
    return dsf_launch_status();
}

extern "C" int dsf_img2pcl(const float* img, const float* center, const float* minv, const float* cube,
                           const dsf_camera* cam, const uint32_t* rand_keys, int B, int S, int n_sample, float* pcl,
                           int32_t* counts, uint32_t* workspace, dsf_stream_t stream) {
    DSF_CHECK_ARG(img && center && minv && cube && cam && rand_keys && pcl && counts && workspace);
    DSF_CHECK_ARG(B >= 0 && S > 1 && n_sample > 0);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(img2pcl_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, img, center, minv, cube, *cam,
                       rand_keys, S, n_sample, pcl, counts, workspace);
    return dsf_launch_status();
}

extern "C" int dsf_joint2offset_forward(const float* joints, const float* img, int B, int J, int H, int S,
                                        float kernel_size, float* maps, const int64_t* map_strides,
                                        dsf_stream_t stream) {
    DSF_CHECK_ARG(joints && img && maps && B >= 0 && J > 0 && J <= 64 && S > 0 && H >= S && H % S == 0);
    if (B == 0) return DSF_OK;
    const int64_t sb = map_strides ? map_strides[0] : (int64_t)4 * J * S * S, sc = map_strides ? map_strides[1] : (int64_t)S * S,
                  sq = map_strides ? map_strides[2] : 1;
    if (sc == 1) {                                          // channels-last rows
        hipLaunchKernelGGL(joint2offset_fwd_cl_kernel, dim3((S * S * J + 255) / 256, B), dim3(256), 0, (hipStream_t)stream,
                           joints, img, J, H, S, kernel_size, maps, sb, sq);
        return dsf_launch_status();
    }
    hipLaunchKernelGGL(joint2offset_fwd_kernel, dim3((S * S + 255) / 256, B), dim3(256), J * 3 * sizeof(float),
                       (hipStream_t)stream, joints, img, J, H, S, kernel_size, maps, sb, sc, sq);
    return dsf_launch_status();
}

extern "C" int dsf_joint2offset_backward(const float* joints, const float* img, const float* grad_maps, int B, int J,
                                         int H, int S, float kernel_size, float* grad_joints, const int64_t* map_strides,
                                         dsf_stream_t stream) {
    DSF_CHECK_ARG(joints && img && grad_maps && grad_joints && B >= 0 && J > 0 && S > 0 && H >= S && H % S == 0);
    if (B == 0) return DSF_OK;
    const int64_t sb = map_strides ? map_strides[0] : (int64_t)4 * J * S * S, sc = map_strides ? map_strides[1] : (int64_t)S * S,
                  sq = map_strides ? map_strides[2] : 1;
    hipLaunchKernelGGL(joint2offset_bwd_kernel, dim3(J, B), dim3(256), 0, (hipStream_t)stream, joints, img, grad_maps, J,
                       H, S, kernel_size, grad_joints, sb, sc, sq);
    return dsf_launch_status();
}

extern "C" int dsf_offset2joint_forward(const float* maps, const float* depth, int B, int J, int H, int S,
                                        float kernel_size, float scale, float* joints, float* stats,
                                        dsf_stream_t stream) {
    DSF_CHECK_ARG(maps && depth && joints && stats && B >= 0 && J > 0 && S > 0 && H >= S && H % S == 0);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(offset2joint_fwd_kernel, dim3(J, B), dim3(256), 0, (hipStream_t)stream, maps, depth, J, H, S,
                       kernel_size, scale, joints, stats, (int64_t)4 * J * S * S, (int64_t)S * S, (int64_t)1);
    return dsf_launch_status();
}

extern "C" int dsf_offset2joint_forward_strided(const float* maps, const int64_t* map_strides, const float* depth, int B, int J, int H, int S,
                                                float kernel_size, float scale, float* joints, float* stats, dsf_stream_t stream) {
    DSF_CHECK_ARG(maps && map_strides && depth && joints && stats && B >= 0 && J > 0 && S > 0 && H >= S && H % S == 0);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(offset2joint_fwd_kernel, dim3(J, B), dim3(256), 0, (hipStream_t)stream, maps, depth, J, H, S,
                       kernel_size, scale, joints, stats, map_strides[0], map_strides[1], map_strides[2]);
    return dsf_launch_status();
}

extern "C" int dsf_offset2joint_backward(const float* maps, const float* depth, const float* joints,
                                         const float* stats, const float* grad_joints, int B, int J, int H, int S,
                                         float kernel_size, float scale, float* grad_maps, dsf_stream_t stream) {
    DSF_CHECK_ARG(maps && depth && joints && stats && grad_joints && grad_maps && B >= 0 && J > 0 && S > 0 && H >= S &&
                  H % S == 0);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(offset2joint_bwd_kernel, dim3(J, B), dim3(256), 0, (hipStream_t)stream, maps, depth, joints,
                       stats, grad_joints, J, H, S, kernel_size, scale, grad_maps, (int64_t)4 * J * S * S, (int64_t)S * S, (int64_t)1);
    return dsf_launch_status();
}

extern "C" int dsf_offset2joint_backward_strided(const float* maps, const int64_t* map_strides, const float* depth, const float* joints,
                                                 const float* stats, const float* grad_joints, int B, int J, int H, int S, float kernel_size,
                                                 float scale, float* grad_maps, dsf_stream_t stream) {
    DSF_CHECK_ARG(maps && map_strides && depth && joints && stats && grad_joints && grad_maps && B >= 0 && J > 0 && S > 0 && H >= S &&
                  H % S == 0);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(offset2joint_bwd_kernel, dim3(J, B), dim3(256), 0, (hipStream_t)stream, maps, depth, joints,
                       stats, grad_joints, J, H, S, kernel_size, scale, grad_maps, map_strides[0], map_strides[1], map_strides[2]);
    return dsf_launch_status();
}

// the decode of a dense channels-last map (B, S S, 4 J), J <= 32: pixel-chunk workgroups with contiguous reads (see o2j_cl_*).
// workspace: dsf_offset2joint_cl_workspace_floats(B, S) floats (per-chunk maxima and sums; no initialisation needed).
extern "C" int64_t dsf_offset2joint_cl_workspace_floats(int B, int S) {
    const int64_t n_ch = ((int64_t)S * S + O2J_PIX - 1) / O2J_PIX;
    return (int64_t)(B > 0 ? B : 0) * n_ch * 32 * 5;
}
extern "C" int dsf_offset2joint_forward_cl(const float* maps, const float* depth, int B, int J, int H, int S, float kernel_size,
                                           float scale, float* joints, float* stats, float* workspace, dsf_stream_t stream) {
    DSF_CHECK_ARG(maps && depth && joints && stats && B >= 0 && J > 0 && S > 0 && H >= S && H % S == 0);
    if (J > 32) return DSF_ERR_UNSUPPORTED;
    if (B == 0) return DSF_OK;
    DSF_CHECK_ARG(workspace);
    const int n_ch = (S * S + O2J_PIX - 1) / O2J_PIX;
    float* pmax = workspace;
    float* part = workspace + (int64_t)B * n_ch * 32;
    hipLaunchKernelGGL(o2j_cl_max_kernel, dim3(n_ch, B), dim3(256), 0, (hipStream_t)stream, maps, depth, J, H, S, scale, pmax);
    hipLaunchKernelGGL(o2j_cl_sum_kernel, dim3(n_ch, B), dim3(256), 0, (hipStream_t)stream, maps, depth, J, H, S, kernel_size, scale,
                       (const float*)pmax, part);
    hipLaunchKernelGGL(o2j_cl_final_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const float*)pmax, (const float*)part, J, n_ch,
                       joints, stats);
    return dsf_launch_status();
}
extern "C" int dsf_offset2joint_backward_cl(const float* maps, const float* depth, const float* joints, const float* stats,
                                            const float* grad_joints, int B, int J, int H, int S, float kernel_size, float scale,
                                            float* grad_maps, dsf_stream_t stream) {
    DSF_CHECK_ARG(maps && depth && joints && stats && grad_joints && grad_maps && B >= 0 && J > 0 && S > 0 && H >= S && H % S == 0);
    if (J > 32) return DSF_ERR_UNSUPPORTED;
    if (B == 0) return DSF_OK;
    const int n_ch = (S * S + O2J_PIX - 1) / O2J_PIX;
    hipLaunchKernelGGL(o2j_cl_bwd_kernel, dim3(n_ch, B), dim3(256), 0, (hipStream_t)stream, maps, depth, joints, stats, grad_joints, J, H,
                       S, kernel_size, scale, grad_maps);
    return dsf_launch_status();
}
