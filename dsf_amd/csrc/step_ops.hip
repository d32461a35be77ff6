// Loss-side glue of the trainer steps as a handful of fused launches (round 6; reference train_render.py:444-466, 645-808).
// The reference writes these terms as chains of elementwise / reduce torch operators over small tensors -- 10-25 launches each,
// forward and backward, 939 torch launches per FinetuneStage step in round 5 -- which on one stream cost their launch latency,
// not their arithmetic.  Every kernel here restates ONE such chain:
//   S1 m2d        model-to-data depth term and the agreement sums of the M2P gate (train_render.py:728-732, 786-789)
//   S2 cube points  MANO points -> camera space and back to cube-normalised (render_model/mano_layer.py:1078-1092: `* cube / 2 + center`,
//                   `(p - center) / cube * 2`) for the vertex and the joint tensor in one launch each way
//   S3 view rotation  RotationPoints (mano_layer.py:874-884) with batch_rodrigues / quat2mat (:773-805): inference only
//   S4 masked part mean  the per-part masked means of JointICPLoss / FingerICPLoss (metric/meshLoss.py:389-394)
//   S5 MANO regularisers  mean(beta^2) and mean(|min(scale, 0)|) of the Pretrain losses (train_render.py:463-464)
//   S7 pooled head  AdaptiveAvgPool2d(1) + Linear of the MANO regression head (model/backbone.py:225-226), forward and backward
//   S6 M2P        the masked Huber term that lets a trusted MANO fit teach the pixel branch (train_render.py:590-603, 787-801)
// Arithmetic per element is the reference's, operation for operation (-ffp-contract=off); reductions run in a fixed order
// (deterministic), which differs from torch's reduction order in the last bits only.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* s) {             // 256 threads, fixed order; s: 4 floats
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s[0] + s[1]) + (s[2] + s[3]);
}

// ---- S1: per sample b over P pixels: u = (r < t) | (s < t), a = (r < t) & (s < t), d = |r - s|
//      sums[b] = { sum d u, sum u, sum d a, sum a };  per[b] = sums[b][0] / (sums[b][1] + 1e-8)
__global__ __launch_bounds__(256) void m2d_sums_kernel(const float* __restrict__ real, const float* __restrict__ synth, int P, float thresh,
                                                       float* __restrict__ sums, float* __restrict__ per) {
    __shared__ float s[4];
    const int b = blockIdx.x;
    const float* r = real + (int64_t)b * P;
    const float* q = synth + (int64_t)b * P;
    float du = 0.f, cu = 0.f, da = 0.f, ca = 0.f;
    for (int i = threadIdx.x; i < P; i += 256) {
        const float rv = r[i], sv = q[i];
        const bool fr = rv < thresh, fs = sv < thresh;
        const float d = fabsf(rv - sv);
        if (fr || fs) { du += d; cu += 1.f; }
        if (fr && fs) { da += d; ca += 1.f; }
    }
    du = block_sum(du, s); cu = block_sum(cu, s); da = block_sum(da, s); ca = block_sum(ca, s);
    if (threadIdx.x == 0) {
        sums[b * 4 + 0] = du; sums[b * 4 + 1] = cu; sums[b * 4 + 2] = da; sums[b * 4 + 3] = ca;
        per[b] = du / (cu + 1e-8f);
    }
}

// loss = mean_b(per[b]) * scale  (one workgroup)
__global__ __launch_bounds__(256) void mean_scale_kernel(const float* __restrict__ per, int B, float scale, float* __restrict__ out) {
    __shared__ float s[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) acc += per[i];
    acc = block_sum(acc, s);
    if (threadIdx.x == 0) out[0] = acc / (float)B * scale;
}

// d loss / d synth[b][p] = g * scale / B * (-sign(r - s)) u / (sum u + 1e-8)
__global__ __launch_bounds__(256) void m2d_bwd_kernel(const float* __restrict__ real, const float* __restrict__ synth, const float* __restrict__ sums,
                                                      const float* __restrict__ g, int B, int P, float thresh, float scale,
                                                      float* __restrict__ gs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * P) return;
    const int b = (int)(i / P);
    const float rv = real[i], sv = synth[i];
    const float k = g[0] * scale / (float)B / (sums[b * 4 + 1] + 1e-8f);
    const float z = rv - sv;
    const float sg = z > 0.f ? 1.f : (z < 0.f ? -1.f : 0.f);
    gs[i] = (rv < thresh || sv < thresh) ? -(sg * k) : 0.f;
}

// ---- S2: w = p * cube / 2 + center;  n = (w - center) / cube * 2   (p: verts (B,NV,3) and joints (B,NJ,3) in one launch)
__global__ __launch_bounds__(256) void cube_points_fwd_kernel(const float* __restrict__ v, const float* __restrict__ j, const float* __restrict__ center,
                                                              const float* __restrict__ cube, int B, int NV, int NJ, float* __restrict__ vw,
                                                              float* __restrict__ jw, float* __restrict__ vn, float* __restrict__ jn) {
    const int per = (NV + NJ) * 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * per) return;
    const int b = (int)(i / per), e = (int)(i % per);
    const bool is_v = e < NV * 3;
    const int64_t o = is_v ? (int64_t)b * NV * 3 + e : (int64_t)b * NJ * 3 + (e - NV * 3);
    const int c = (is_v ? e : e - NV * 3) % 3;
    const float cu = cube[b * 3 + c], ce = center[b * 3 + c];
    const float p = is_v ? v[o] : j[o];
    const float w = p * cu / 2.f + ce;
    const float n = (w - ce) / cu * 2.f;
    if (is_v) { vw[o] = w; vn[o] = n; } else { jw[o] = w; jn[o] = n; }
}

// g_p = ((g_w + g_n * 2 / cube) / 2) * cube   (either incoming gradient may be absent)
__global__ __launch_bounds__(256) void cube_points_bwd_kernel(const float* __restrict__ gvw, const float* __restrict__ gjw, const float* __restrict__ gvn,
                                                              const float* __restrict__ gjn, const float* __restrict__ cube, int B, int NV, int NJ,
                                                              float* __restrict__ gv, float* __restrict__ gj) {
    const int per = (NV + NJ) * 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * per) return;
    const int b = (int)(i / per), e = (int)(i % per);
    const bool is_v = e < NV * 3;
    const int64_t o = is_v ? (int64_t)b * NV * 3 + e : (int64_t)b * NJ * 3 + (e - NV * 3);
    const int c = (is_v ? e : e - NV * 3) % 3;
    const float cu = cube[b * 3 + c];
    const float* gw = is_v ? gvw : gjw;
    const float* gn = is_v ? gvn : gjn;
    float t = gw ? gw[o] : 0.f;
    if (gn) t = t + gn[o] * 2.f / cu;
    const float r = t / 2.f * cu;
    if (is_v) gv[o] = r; else gj[o] = r;
}

// ---- S3: rotation about `center` by an axis-angle (rot_dim 3) or a quaternion (rot_dim 4) per sample.  recentre: the points are first
//      moved so that the mean of the joints sits at `center` (Render.forward, mano_layer.py:995-1003: p - mean(joints) + center);
//      rot == nullptr: no rotation
__global__ __launch_bounds__(256) void view_rotate_kernel(const float* __restrict__ v, const float* __restrict__ j, const float* __restrict__ center,
                                                          const float* __restrict__ rot, int rot_dim, int recentre, int NV, int NJ,
                                                          float* __restrict__ ov, float* __restrict__ oj) {
    __shared__ float R[9], SC[3];
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        float q[4] = {1.f, 0.f, 0.f, 0.f};
        if (rot && rot_dim == 3) {                                        // batch_rodrigues: angle = |theta + 1e-8|, quat = [cos(a/2), sin(a/2) theta / a]
            const float t0 = rot[b * 3 + 0], t1 = rot[b * 3 + 1], t2 = rot[b * 3 + 2];
            const float a0 = t0 + 1e-8f, a1 = t1 + 1e-8f, a2 = t2 + 1e-8f;
            const float angle = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
            const float half = angle * 0.5f, sn = sinf(half);
            q[0] = cosf(half); q[1] = sn * (t0 / angle); q[2] = sn * (t1 / angle); q[3] = sn * (t2 / angle);
        } else if (rot) {
            q[0] = rot[b * 4 + 0]; q[1] = rot[b * 4 + 1]; q[2] = rot[b * 4 + 2]; q[3] = rot[b * 4 + 3];
        }
        const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const float w = q[0] / nrm, x = q[1] / nrm, y = q[2] / nrm, z = q[3] / nrm;
        R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * x * y - 2 * w * z;         R[2] = 2 * w * y + 2 * x * z;
        R[3] = 2 * w * z + 2 * x * y;         R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * y * z - 2 * w * x;
        R[6] = 2 * x * z - 2 * w * y;         R[7] = 2 * w * x + 2 * y * z;         R[8] = w * w - x * x - y * y + z * z;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 67) {                          // mean of the joints (another wave than the matrix)
        const int a = threadIdx.x - 64;
        float sum = 0.f;
        if (recentre)
            for (int k = 0; k < NJ; ++k) sum += j[((int64_t)b * NJ + k) * 3 + a];
        SC[a] = recentre ? sum / (float)NJ : 0.f;
    }
    __syncthreads();
    const float c0 = center[b * 3 + 0], c1 = center[b * 3 + 1], c2 = center[b * 3 + 2];
    for (int p = threadIdx.x; p < NV + NJ; p += 256) {
        const bool is_v = p < NV;
        const int64_t o = is_v ? ((int64_t)b * NV + p) * 3 : ((int64_t)b * NJ + (p - NV)) * 3;
        const float* src = is_v ? v : j;
        float* dst = is_v ? ov : oj;
        float p0 = src[o], p1 = src[o + 1], p2 = src[o + 2];
        if (recentre) { p0 = (p0 - SC[0]) + c0; p1 = (p1 - SC[1]) + c1; p2 = (p2 - SC[2]) + c2; }
        if (rot) {
            const float d0 = p0 - c0, d1 = p1 - c1, d2 = p2 - c2;
            p0 = (d0 * R[0] + d1 * R[1] + d2 * R[2]) + c0;                // row i of (p - c) R^T = sum_k (p - c)_k R[i][k]
            p1 = (d0 * R[3] + d1 * R[4] + d2 * R[5]) + c1;
            p2 = (d0 * R[6] + d1 * R[7] + d2 * R[8]) + c2;
        }
        dst[o] = p0; dst[o + 1] = p1; dst[o + 2] = p2;
    }
}

// n = (p - center) / cube * 2 for the vertex and the joint tensor (Render.forward, mano_layer.py:1033-1034)
__global__ __launch_bounds__(256) void cube_normalise_kernel(const float* __restrict__ v, const float* __restrict__ j, const float* __restrict__ center,
                                                             const float* __restrict__ cube, int B, int NV, int NJ, float* __restrict__ vn,
                                                             float* __restrict__ jn) {
    const int per = (NV + NJ) * 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * per) return;
    const int b = (int)(i / per), e = (int)(i % per);
    const bool is_v = e < NV * 3;
    const int64_t o = is_v ? (int64_t)b * NV * 3 + e : (int64_t)b * NJ * 3 + (e - NV * 3);
    const int c = (is_v ? e : e - NV * 3) % 3;
    const float p = is_v ? v[o] : j[o];
    const float n = (p - center[b * 3 + c]) / cube[b * 3 + c] * 2.f;
    if (is_v) vn[o] = n; else jn[o] = n;
}

// ---- S6: the M2P term (train_render.py:590-603 / 787-801): rows (b, joint) with ok[b] && jm[b][joint], jm = [1, pd2m < t (15), pd2m[2,5,8,11,14] < t];
//      val = sum_rows mean_k h(a - b) / max(#rows, 1) * weight, 0 when no row with index > 0 is selected (the reference tests the SUM OF THE
//      SELECTED INDICES == 0 on the host); aux = {#rows, 1 if any selected row has index > 0}.  One workgroup; fixed-order sums.
__device__ __forceinline__ bool m2p_row(const unsigned char* ok, const float* pd2m, float t, int b, int jn) {
    if (!ok[b]) return false;
    if (jn == 0) return true;
    const int col = jn <= 15 ? jn - 1 : (jn - 16) * 3 + 2;               // 16..20 -> parts 2, 5, 8, 11, 14
    return pd2m[b * 15 + col] < t;
}
__global__ __launch_bounds__(256) void m2p_fwd_kernel(const float* __restrict__ a, const float* __restrict__ bb, const unsigned char* __restrict__ ok,
                                                      const float* __restrict__ pd2m, int B, float t, float delta, float weight,
                                                      float* __restrict__ out, float* __restrict__ aux) {
    __shared__ float s[4];
    float acc = 0.f, cnt = 0.f, hi = 0.f;
    for (int r = threadIdx.x; r < B * 21; r += 256) {
        if (!m2p_row(ok, pd2m, t, r / 21, r % 21)) continue;
        float h = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float z = a[r * 3 + k] - bb[r * 3 + k], az = fabsf(z);
            h += az < delta ? (0.5f * z) * z : delta * (az - 0.5f * delta);
        }
        acc += h / 3.f; cnt += 1.f; if (r > 0) hi = 1.f;
    }
    acc = block_sum(acc, s); cnt = block_sum(cnt, s); hi = block_sum(hi, s);
    if (threadIdx.x == 0) {
        const float val = acc / fmaxf(cnt, 1.f);
        out[0] = hi == 0.f ? 0.f : val * weight;
        aux[0] = cnt; aux[1] = hi == 0.f ? 0.f : 1.f;
    }
}
__global__ __launch_bounds__(256) void m2p_bwd_kernel(const float* __restrict__ a, const float* __restrict__ bb, const unsigned char* __restrict__ ok,
                                                      const float* __restrict__ pd2m, const float* __restrict__ aux, const float* __restrict__ g, int B,
                                                      float t, float delta, float weight, float* __restrict__ ga) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * 63) return;
    const int r = i / 3;
    float res = 0.f;
    if (aux[1] != 0.f && m2p_row(ok, pd2m, t, r / 21, r % 21)) {
        const float z = a[i] - bb[i];
        const float hg = fabsf(z) < delta ? z : copysignf(delta, z);
        res = g[0] * weight / fmaxf(aux[0], 1.f) / 3.f * hg;
    }
    ga[i] = res;
}

// ---- S4: out[b][k] = sum_{p: seg = k + 1} dis[b][p] / (#{p: seg = k + 1, dis > 0} + 1e-8), 0 when that count is 0;  valid[b][k] = the count
constexpr int PM_MAX_PARTS = 16;
__global__ __launch_bounds__(256) void part_mean_fwd_kernel(const float* __restrict__ dis, const int64_t* __restrict__ seg, int P, int n_parts,
                                                            float* __restrict__ out, float* __restrict__ valid) {
    __shared__ float s[4];
    const int b = blockIdx.x;
    float sum[PM_MAX_PARTS], cnt[PM_MAX_PARTS];
#pragma unroll
    for (int k = 0; k < PM_MAX_PARTS; ++k) { sum[k] = 0.f; cnt[k] = 0.f; }
    for (int p = threadIdx.x; p < P; p += 256) {
        const int lab = (int)seg[(int64_t)b * P + p];
        const float d = dis[(int64_t)b * P + p];
#pragma unroll
        for (int k = 0; k < PM_MAX_PARTS; ++k)
            if (lab == k + 1) { sum[k] += d; cnt[k] += (d > 0.f) ? 1.f : 0.f; }
    }
#pragma unroll
    for (int k = 0; k < PM_MAX_PARTS; ++k) {                              // (all 16 folded: constant register indices; the extra ones are zeros)
        const float su = block_sum(sum[k], s), cn = block_sum(cnt[k], s);
        if (threadIdx.x == 0 && k < n_parts) {
            out[b * n_parts + k] = cn == 0.f ? 0.f : su / (cn + 1e-8f);
            valid[b * n_parts + k] = cn;
        }
    }
}

__global__ __launch_bounds__(256) void part_mean_bwd_kernel(const float* __restrict__ g, const int64_t* __restrict__ seg, const float* __restrict__ valid,
                                                            int B, int P, int n_parts, float* __restrict__ gd) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * P) return;
    const int b = (int)(i / P), lab = (int)seg[i];
    float r = 0.f;
    if (lab >= 1 && lab <= n_parts) {
        const float cn = valid[b * n_parts + lab - 1];
        if (cn != 0.f) r = g[b * n_parts + lab - 1] / (cn + 1e-8f);
    }
    gd[i] = r;
}

// ---- S5: out[0] = mean_{b, c in [c0, c0 + 10)} p[b][c]^2 * w_beta;  out[1] = mean_b |min(p[b][cs], 0)| * w_scale   (p: (B, W) rows)
__global__ __launch_bounds__(256) void mano_reg_fwd_kernel(const float* __restrict__ p, int B, int W, int c0, int cs, float w_beta, float w_scale,
                                                           float* __restrict__ out) {
    __shared__ float s[4];
    float a = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < B * 10; i += 256) { const float v = p[(int64_t)(i / 10) * W + c0 + i % 10]; a += v * v; }
    for (int i = threadIdx.x; i < B; i += 256) c += fabsf(fminf(p[(int64_t)i * W + cs], 0.f));
    a = block_sum(a, s); c = block_sum(c, s);
    if (threadIdx.x == 0) { out[0] = a / (float)(B * 10) * w_beta; out[1] = c / (float)B * w_scale; }
}

// gp (B, W), every column written: 2 p g0 w_beta / (10 B) on the shape columns, -g1 w_scale / B where the scale is negative, 0 elsewhere
__global__ __launch_bounds__(256) void mano_reg_bwd_kernel(const float* __restrict__ p, const float* __restrict__ g, int B, int W, int c0, int cs,
                                                           float w_beta, float w_scale, float* __restrict__ gp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * W) return;
    const int c = (int)(i % W);
    float r = 0.f;
    if (c >= c0 && c < c0 + 10) r = g[0] * w_beta / (float)(B * 10) * (2.f * p[i]);
    else if (c == cs) r = p[i] < 0.f ? -(g[1] * w_scale / (float)B) : 0.f;
    gp[i] = r;
}

// ---- S7: the MANO regression head (model/backbone.py:225-226: AdaptiveAvgPool2d(1) -> Flatten -> Linear(C, 62)) on a channels-last
//      feature map x (B, HW, C): pooled[b][c] = mean_p x[b][p][c];  out[b][o] = bias[o] + sum_c pooled[b][c] W[o][c].  One workgroup
//      per sample.  torch runs it as a mean reduction + a hipBLASLt GEMM (a 46 us launch for a 32 x 512 x 62 product) + their
//      five backward launches; fixed-order sums here.
constexpr int PL_MAX_C = 2048;
// The pooling walks HW pixels of C channels: thread = (float4 column, pixel lane), four pixels in flight per thread (round 6: one
// thread per channel walking all pixels one dependent load after the other took 97 us at B = 32, 8 x 8 x 512 -- at the END of the
// forward pass, where nothing hides it), pixel lanes folded through LDS in a fixed order.
__global__ __launch_bounds__(256) void pool_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                              int HW, int C, int O, float* __restrict__ pooled, float* __restrict__ out) {
    __shared__ float s_p[PL_MAX_C];
    __shared__ __attribute__((aligned(16))) float s_part[4][PL_MAX_C];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* xb = x + (int64_t)b * HW * C;
    if ((C & 3) == 0) {
        const int C4 = C >> 2;
        const int cols = C4 < 256 ? C4 : 256;                  // float4 columns a pass covers
        const int npl = (256 / cols) < 4 ? (256 / cols) : 4;   // pixel lanes (threads beyond cols * npl idle in the pooling)
        const int cq = t % cols, pl = t / cols;
        for (int c0 = 0; c0 < C4; c0 += cols) {
            const int col = c0 + cq;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col < C4 && pl < npl) {
                const float4* xq = reinterpret_cast<const float4*>(xb) + col;
                int p = pl;
                for (; p + 3 * npl < HW; p += 4 * npl) {
                    const float4 v0 = xq[(int64_t)p * C4], v1 = xq[(int64_t)(p + npl) * C4], v2 = xq[(int64_t)(p + 2 * npl) * C4],
                                 v3 = xq[(int64_t)(p + 3 * npl) * C4];
                    a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
                    a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
                }
                for (; p < HW; p += npl) { const float4 v = xq[(int64_t)p * C4]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
                *reinterpret_cast<float4*>(&s_part[pl][col * 4]) = a;
            }
        }
        __syncthreads();
        for (int c = t; c < C; c += 256) {
            float acc = s_part[0][c];
            for (int k = 1; k < npl; ++k) acc += s_part[k][c];
            const float m = acc / (float)HW;
            s_p[c] = m;
            pooled[(int64_t)b * C + c] = m;
        }
    } else {
        for (int c = t; c < C; c += 256) {
            float acc = 0.f;
            for (int p = 0; p < HW; ++p) acc += xb[(int64_t)p * C + c];
            const float m = acc / (float)HW;
            s_p[c] = m;
            pooled[(int64_t)b * C + c] = m;
        }
    }
    __syncthreads();
    for (int o = wave; o < O; o += 4) {
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc += s_p[c] * W[(int64_t)o * C + c];
        acc = wave_sum(acc);
        if (lane == 0) out[(int64_t)b * O + o] = acc + (bias ? bias[o] : 0.f);
    }
}

// gx[b][p][c] = (sum_o g[b][o] W[o][c]) / HW   (one workgroup per sample)
__global__ __launch_bounds__(256) void pool_linear_bwd_x_kernel(const float* __restrict__ g, const float* __restrict__ W, int HW, int C, int O,
                                                                float* __restrict__ gx) {
    __shared__ float s_g[64];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < O) s_g[t] = g[(int64_t)b * O + t];
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        float acc = 0.f;
        for (int o = 0; o < O; ++o) acc += s_g[o] * W[(int64_t)o * C + c];
        const float v = acc / (float)HW;
        for (int p = 0; p < HW; ++p) gx[((int64_t)b * HW + p) * C + c] = v;
    }
}

// gW[o][c] = sum_b g[b][o] pooled[b][c];  gb[o] = sum_b g[b][o]   (one workgroup per output row o)
__global__ __launch_bounds__(256) void pool_linear_bwd_w_kernel(const float* __restrict__ g, const float* __restrict__ pooled, int B, int C, int O,
                                                                float* __restrict__ gW, float* __restrict__ gb) {
    const int o = blockIdx.x, t = threadIdx.x;
    for (int c = t; c < C; c += 256) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += g[(int64_t)b * O + o] * pooled[(int64_t)b * C + c];
        gW[(int64_t)o * C + c] = acc;
    }
    if (t == 0 && gb) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += g[(int64_t)b * O + o];
        gb[o] = acc;
    }
}

}  // namespace

extern "C" int dsf_m2d_forward(const float* real, const float* synth, int B, int P, float thresh, float scale, float* sums, float* per,
                               float* loss, dsf_stream_t stream) {
    DSF_CHECK_ARG(real && synth && sums && per && loss && B >= 0 && P > 0);
    if (B == 0) return DSF_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(m2d_sums_kernel, dim3(B), dim3(256), 0, st, real, synth, P, thresh, sums, per);
    hipLaunchKernelGGL(mean_scale_kernel, dim3(1), dim3(256), 0, st, per, B, scale, loss);
    return dsf_launch_status();
}

extern "C" int dsf_m2d_backward(const float* real, const float* synth, const float* sums, const float* grad_loss, int B, int P, float thresh,
                                float scale, float* grad_synth, dsf_stream_t stream) {
    DSF_CHECK_ARG(real && synth && sums && grad_loss && grad_synth && B >= 0 && P > 0);
    if (B == 0) return DSF_OK;
    const int64_t n = (int64_t)B * P;
    hipLaunchKernelGGL(m2d_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, real, synth, sums, grad_loss, B, P, thresh,
                       scale, grad_synth);
    return dsf_launch_status();
}

extern "C" int dsf_cube_points_forward(const float* verts, const float* joints, const float* center, const float* cube, int B, int NV, int NJ,
                                       float* verts_world, float* joints_world, float* verts_norm, float* joints_norm, dsf_stream_t stream) {
    DSF_CHECK_ARG(verts && joints && center && cube && verts_world && joints_world && verts_norm && joints_norm && B >= 0 && NV >= 0 && NJ >= 0);
    const int64_t n = (int64_t)B * (NV + NJ) * 3;
    if (n == 0) return DSF_OK;
    hipLaunchKernelGGL(cube_points_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, verts, joints, center, cube, B, NV,
                       NJ, verts_world, joints_world, verts_norm, joints_norm);
    return dsf_launch_status();
}

extern "C" int dsf_cube_points_backward(const float* g_verts_world, const float* g_joints_world, const float* g_verts_norm,
                                        const float* g_joints_norm, const float* cube, int B, int NV, int NJ, float* g_verts, float* g_joints,
                                        dsf_stream_t stream) {
    DSF_CHECK_ARG(cube && g_verts && g_joints && B >= 0 && NV >= 0 && NJ >= 0);
    const int64_t n = (int64_t)B * (NV + NJ) * 3;
    if (n == 0) return DSF_OK;
    hipLaunchKernelGGL(cube_points_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_verts_world, g_joints_world,
                       g_verts_norm, g_joints_norm, cube, B, NV, NJ, g_verts, g_joints);
    return dsf_launch_status();
}

extern "C" int dsf_view_rotate(const float* verts, const float* joints, const float* center, const float* rot, int rot_dim, int recentre, int B,
                               int NV, int NJ, float* verts_out, float* joints_out, dsf_stream_t stream) {
    DSF_CHECK_ARG(verts && joints && center && verts_out && joints_out && (!rot || rot_dim == 3 || rot_dim == 4) && B >= 0 && NJ > 0);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(view_rotate_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, verts, joints, center, rot, rot_dim, recentre, NV, NJ,
                       verts_out, joints_out);
    return dsf_launch_status();
}

extern "C" int dsf_cube_normalise(const float* verts, const float* joints, const float* center, const float* cube, int B, int NV, int NJ,
                                  float* verts_norm, float* joints_norm, dsf_stream_t stream) {
    DSF_CHECK_ARG(verts && joints && center && cube && verts_norm && joints_norm && B >= 0 && NV >= 0 && NJ >= 0);
    const int64_t n = (int64_t)B * (NV + NJ) * 3;
    if (n == 0) return DSF_OK;
    hipLaunchKernelGGL(cube_normalise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, verts, joints, center, cube, B, NV,
                       NJ, verts_norm, joints_norm);
    return dsf_launch_status();
}

extern "C" int dsf_m2p_forward(const float* juvd_pix, const float* juvd_mano, const unsigned char* sample_ok, const float* part_dist, int B,
                               float part_thresh, float delta, float weight, float* out, float* aux, dsf_stream_t stream) {
    DSF_CHECK_ARG(juvd_pix && juvd_mano && sample_ok && part_dist && out && aux && B > 0);
    hipLaunchKernelGGL(m2p_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, juvd_pix, juvd_mano, sample_ok, part_dist, B, part_thresh, delta,
                       weight, out, aux);
    return dsf_launch_status();
}

extern "C" int dsf_m2p_backward(const float* juvd_pix, const float* juvd_mano, const unsigned char* sample_ok, const float* part_dist,
                                const float* aux, const float* grad_out, int B, float part_thresh, float delta, float weight, float* grad_pix,
                                dsf_stream_t stream) {
    DSF_CHECK_ARG(juvd_pix && juvd_mano && sample_ok && part_dist && aux && grad_out && grad_pix && B > 0);
    hipLaunchKernelGGL(m2p_bwd_kernel, dim3((B * 63 + 255) / 256), dim3(256), 0, (hipStream_t)stream, juvd_pix, juvd_mano, sample_ok, part_dist, aux,
                       grad_out, B, part_thresh, delta, weight, grad_pix);
    return dsf_launch_status();
}

extern "C" int dsf_part_mean_forward(const float* dis, const int64_t* seg, int B, int P, int n_parts, float* out, float* valid,
                                     dsf_stream_t stream) {
    DSF_CHECK_ARG(dis && seg && out && valid && B >= 0 && P > 0);
    if (n_parts < 1 || n_parts > PM_MAX_PARTS) return DSF_ERR_UNSUPPORTED;
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(part_mean_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dis, seg, P, n_parts, out, valid);
    return dsf_launch_status();
}

extern "C" int dsf_part_mean_backward(const float* grad_out, const int64_t* seg, const float* valid, int B, int P, int n_parts, float* grad_dis,
                                      dsf_stream_t stream) {
    DSF_CHECK_ARG(grad_out && seg && valid && grad_dis && B >= 0 && P > 0 && n_parts >= 1 && n_parts <= PM_MAX_PARTS);
    const int64_t n = (int64_t)B * P;
    if (n == 0) return DSF_OK;
    hipLaunchKernelGGL(part_mean_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_out, seg, valid, B, P, n_parts,
                       grad_dis);
    return dsf_launch_status();
}

extern "C" int dsf_mano_reg_forward(const float* paras, int B, int W, int beta_col, int scale_col, float w_beta, float w_scale, float* out,
                                    dsf_stream_t stream) {
    DSF_CHECK_ARG(paras && out && B > 0 && W > 0 && beta_col >= 0 && beta_col + 10 <= W && scale_col >= 0 && scale_col < W);
    hipLaunchKernelGGL(mano_reg_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, paras, B, W, beta_col, scale_col, w_beta, w_scale, out);
    return dsf_launch_status();
}

extern "C" int dsf_mano_reg_backward(const float* paras, const float* grad_out, int B, int W, int beta_col, int scale_col, float w_beta,
                                     float w_scale, float* grad_paras, dsf_stream_t stream) {
    DSF_CHECK_ARG(paras && grad_out && grad_paras && B > 0 && W > 0 && beta_col >= 0 && beta_col + 10 <= W && scale_col >= 0 && scale_col < W);
    const int64_t n = (int64_t)B * W;
    hipLaunchKernelGGL(mano_reg_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, paras, grad_out, B, W, beta_col,
                       scale_col, w_beta, w_scale, grad_paras);
    return dsf_launch_status();
}

extern "C" int dsf_pool_linear_forward(const float* x, const float* weight, const float* bias, int B, int HW, int C, int O, float* pooled,
                                       float* out, dsf_stream_t stream) {
    DSF_CHECK_ARG(x && weight && pooled && out && B >= 0 && HW > 0 && C > 0 && O > 0);
    if (C > PL_MAX_C) return DSF_ERR_UNSUPPORTED;
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(pool_linear_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, weight, bias, HW, C, O, pooled, out);
    return dsf_launch_status();
}

extern "C" int dsf_pool_linear_backward(const float* grad_out, const float* pooled, const float* weight, int B, int HW, int C, int O,
                                        float* grad_x, float* grad_weight, float* grad_bias, dsf_stream_t stream) {
    DSF_CHECK_ARG(grad_out && pooled && weight && B >= 0 && HW > 0 && C > 0 && O > 0 && O <= 64);
    if (B == 0) return DSF_OK;
    if (grad_x) hipLaunchKernelGGL(pool_linear_bwd_x_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, grad_out, weight, HW, C, O, grad_x);
    if (grad_weight) hipLaunchKernelGGL(pool_linear_bwd_w_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, grad_out, pooled, B, C, O, grad_weight,
                                        grad_bias);
    return dsf_launch_status();
}

// ---- channel concatenation of channels-last maps, one launch ---------------------------------------------------------------------
// torch.cat((c0, feat, pix, remap), dim=1) in front of the stage-2 fusion convolution (reference model/backbone.py:256): torch copies
// the four NHWC maps into the 488-channel one with three launches at 3.7 TB/s (139 us for 256 MB at B = 32, on the critical path:
// every input must be there, the fusion convolution waits).  Here a thread moves one float4 of the OUTPUT: non-temporal loads from
// the source its channel quad falls into, ordinary stores (the convolution reads the result next); four float4 in flight per thread.
namespace {
struct Cat4 { const float* src[4]; int q[4]; };          // channel quads per source (0: absent)
typedef float cat_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void cat_channels_kernel(Cat4 c, float* __restrict__ out, uint32_t qt, uint32_t n4) {
    constexpr int U = 4;
    const uint32_t S = gridDim.x * 256u;
    for (uint32_t i0 = blockIdx.x * 256u + threadIdx.x; i0 < n4; i0 += U * S) {
        cat_v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = i0 + u * S;
            if (i >= n4) break;
            const uint32_t pix = i / qt;
            uint32_t q = i - pix * qt;
            int s = 0;
            if (q >= (uint32_t)c.q[0]) { q -= c.q[0]; s = 1; if (q >= (uint32_t)c.q[1]) { q -= c.q[1]; s = 2; if (q >= (uint32_t)c.q[2]) { q -= c.q[2]; s = 3; } } }
            const float* p = s == 0 ? c.src[0] : (s == 1 ? c.src[1] : (s == 2 ? c.src[2] : c.src[3]));
            const int ld = s == 0 ? c.q[0] : (s == 1 ? c.q[1] : (s == 2 ? c.q[2] : c.q[3]));
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const cat_v4f*>(p) + ((int64_t)pix * ld + q));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = i0 + u * S;
            if (i >= n4) break;
            reinterpret_cast<cat_v4f*>(out)[i] = v[u];
        }
    }
}
}  // namespace

extern "C" int dsf_cat_channels_nhwc(const float* a, int ca, const float* b, int cb, const float* c, int cc, const float* d, int cd, float* out,
                                     int64_t pixels, dsf_stream_t stream) {
    DSF_CHECK_ARG(a && out && ca > 0 && cb >= 0 && cc >= 0 && cd >= 0 && pixels >= 0 && (cb == 0 || b) && (cc == 0 || c) && (cd == 0 || d));
    DSF_CHECK_ARG((cb > 0 || (cc == 0 && cd == 0)) && (cc > 0 || cd == 0));            // sources are given in order, without holes
    if (((ca | cb | cc | cd) & 3) != 0) return DSF_ERR_UNSUPPORTED;
    const int64_t qt = (ca + cb + cc + cd) >> 2, n4 = pixels * qt;
    if (n4 >= ((int64_t)1 << 32)) return DSF_ERR_UNSUPPORTED;
    if (n4 == 0) return DSF_OK;
    Cat4 k = {{a, b, c, d}, {ca >> 2, cb >> 2, cc >> 2, cd >> 2}};
    int64_t g = (n4 + 1023) / 1024;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(cat_channels_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, k, out, (uint32_t)qt, (uint32_t)n4);
    return dsf_launch_status();
}
