// K5: fused MANO layer forward / backward for gfx950.
//
// One 1024-thread workgroup per sample.  The 1.4 MB of model constants
// (posedirs 135x2334, shapedirs 10x2334, weights, regressor) are read with
// lane-contiguous dword loads and stay L2/Infinity-Cache resident across the
// batch; everything per-sample (v_posed, posed verts, the 16 joint transforms)
// lives in LDS.  The kinematic chain is walked with 12 lanes per joint.
//
// Reference: MANO_SMPL.forward / get_mano_vertices / batch_rodrigues / quat2mat /
// batch_global_rigid_transformation, render_model/mano_layer.py:573-770.
#include "../../dsf_amd/csrc/common.h"

namespace {

constexpr int NV = 778;
constexpr int NE = 2334;            // 778*3
// (779 output vertices: 778 + the wrist vertex) *
constexpr int NEO = 2337;           // 779*3
constexpr int SV_VPOSED = 0, SV_VERTS = 2334, SV_JOINTS = 4671, SV_G = 4734, SV_RS = 4926, SV_J = 5070,
              SV_TH = 5118;
static_assert(SV_TH + 45 <= DSF_MANO_SAVE_FLOATS, "save layout");

__device__ __forceinline__ void quat_to_rot(float w, float x, float y, float z, float* R) {
    const float ww = w * w, xx = x * x, yy = y * y, zz = z * z;
    const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    R[0] = ww + xx - yy - zz; R[1] = 2 * xy - 2 * wz;   R[2] = 2 * wy + 2 * xz;
    R[3] = 2 * wz + 2 * xy;   R[4] = ww - xx + yy - zz; R[5] = 2 * yz - 2 * wx;
    R[6] = 2 * xz - 2 * wy;   R[7] = 2 * wx + 2 * yz;   R[8] = ww - xx - yy + zz;
}

// batch_rodrigues (mano_layer.py:720-728): 1e-8 inside the norm, axis = theta/angle,
// quaternion re-normalised inside quat2mat.
__device__ __forceinline__ void rodrigues(const float* th, float* R) {
    const float t0 = th[0] + 1e-8f, t1 = th[1] + 1e-8f, t2 = th[2] + 1e-8f;
    const float a = sqrtf(t0 * t0 + t1 * t1 + t2 * t2);
    const float h = a * 0.5f;
    const float c = cosf(h), s = sinf(h);
    float q0 = c, q1 = s * (th[0] / a), q2 = s * (th[1] / a), q3 = s * (th[2] / a);
    const float nq = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    quat_to_rot(q0 / nq, q1 / nq, q2 / nq, q3 / nq, R);
}

// dL/dq_normalised from dL/dR
__device__ __forceinline__ void quat_rot_bwd(float w, float x, float y, float z, const float* G, float* g) {
    g[0] = 2 * w * (G[0] + G[4] + G[8]) + 2 * (-z * G[1] + y * G[2] + z * G[3] - x * G[5] - y * G[6] + x * G[7]);
    g[1] = 2 * x * (G[0] - G[4] - G[8]) + 2 * (y * G[1] + z * G[2] + y * G[3] - w * G[5] + z * G[6] + w * G[7]);
    g[2] = 2 * y * (-G[0] + G[4] - G[8]) + 2 * (x * G[1] + w * G[2] + x * G[3] + z * G[5] - w * G[6] + z * G[7]);
    g[3] = 2 * z * (-G[0] - G[4] + G[8]) + 2 * (-w * G[1] + x * G[2] + w * G[3] + y * G[5] + x * G[6] + y * G[7]);
}

__device__ __forceinline__ void quat_bwd(const float* q, const float* G, float* gq) {
    const float nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float w = q[0] / nq, x = q[1] / nq, y = q[2] / nq, z = q[3] / nq;
    float g[4];
    quat_rot_bwd(w, x, y, z, G, g);
    const float d = w * g[0] + x * g[1] + y * g[2] + z * g[3];
    gq[0] = (g[0] - w * d) / nq; gq[1] = (g[1] - x * d) / nq;
    gq[2] = (g[2] - y * d) / nq; gq[3] = (g[3] - z * d) / nq;
}

__device__ __forceinline__ void rodrigues_bwd(const float* th, const float* G, float* gth) {
    const float t0 = th[0] + 1e-8f, t1 = th[1] + 1e-8f, t2 = th[2] + 1e-8f;
    const float a = sqrtf(t0 * t0 + t1 * t1 + t2 * t2);
    const float h = a * 0.5f;
    const float c = cosf(h), s = sinf(h);
    const float n0 = th[0] / a, n1 = th[1] / a, n2 = th[2] / a;
    float q[4] = {c, s * n0, s * n1, s * n2};
    float gq[4];
    quat_bwd(q, G, gq);
    const float gn0 = s * gq[1], gn1 = s * gq[2], gn2 = s * gq[3];
    const float gs = n0 * gq[1] + n1 * gq[2] + n2 * gq[3];
    const float gh = -s * gq[0] + c * gs;
    const float ga = gh * 0.5f - (gn0 * th[0] + gn1 * th[1] + gn2 * th[2]) / (a * a);
    gth[0] = gn0 / a + ga * (t0 / a);
    gth[1] = gn1 / a + ga * (t1 / a);
    gth[2] = gn2 / a + ga * (t2 / a);
}

// 1024 threads per sample: every element / vertex loop is thread-parallel, and with one workgroup per sample (32 per
// GPU at the benchmark batch) the launch is latency-bound, so a CU's worth of waves per sample is what hides the 145
// dependent posedirs loads per element (88 -> see profiles).  Each output is still produced by one thread with an
// unchanged operation order: results are bit-identical to the 256-thread version.
constexpr int FWD_NT = 1024;
__global__ __launch_bounds__(FWD_NT) void mano_fwd_kernel(dsf_mano_model m, const float* __restrict__ beta,
                                                       const float* __restrict__ theta,
                                                       const float* __restrict__ rot,
                                                       const float* __restrict__ cam, int ncomp, int rot_dim, int ps,
                                                       float k1, float k2, float* __restrict__ verts,
                                                       float* __restrict__ joints, float* __restrict__ Rs_out,
                                                       float* __restrict__ save) {
    __shared__ float s_vp[NE];
    __shared__ float s_v[NEO];
    __shared__ float s_beta[10], s_theta[45], s_rot[4], s_thf[45], s_pf[135];
    __shared__ float s_R[16 * 9], s_J[48], s_G[16 * 12], s_A[16 * 12], s_jnt[63];
    const int b = blockIdx.x, t = threadIdx.x;

    // ps: floats between consecutive samples of beta / theta / rot / cam (0 = each array tightly packed); the four
    // pointers may be column offsets into one (B, 62) parameter matrix
    const int sb = ps ? ps : 10, st = ps ? ps : ncomp, sr = ps ? ps : rot_dim, scam = ps ? ps : 4;
    if (t < 10) s_beta[t] = beta[b * sb + t];
    if (t >= 64 && t < 64 + ncomp) s_theta[t - 64] = theta[b * st + t - 64];
    if (t >= 128 && t < 128 + rot_dim) s_rot[t - 128] = rot[b * sr + t - 128];
    __syncthreads();

    // full pose = theta . comp[:ncomp] + mean (:601); rest joints J = J_template + beta . J_shapedirs
    if (t < 45) {
        float acc = 0.f;
        for (int c = 0; c < ncomp; ++c) acc = fmaf(s_theta[c], m.hands_comp[c * 45 + t], acc);
        s_thf[t] = acc + m.hands_mean[t];
    } else if (t >= 64 && t < 112) {
        const int k = t - 64;
        float acc = m.j_template[k];
        for (int s = 0; s < 10; ++s) acc = fmaf(s_beta[s], m.j_shapedirs[s * 48 + k], acc);
        s_J[k] = acc;
    }
    __syncthreads();

    if (t < 16) {
        float R[9];
        if (t == 0) {
            if (rot_dim == 3) {
                rodrigues(s_rot, R);
            } else {
                const float nq = sqrtf(s_rot[0] * s_rot[0] + s_rot[1] * s_rot[1] + s_rot[2] * s_rot[2] + s_rot[3] * s_rot[3]);
                quat_to_rot(s_rot[0] / nq, s_rot[1] / nq, s_rot[2] / nq, s_rot[3] / nq, R);
            }
        } else {
            rodrigues(s_thf + (t - 1) * 3, R);
#pragma unroll
            for (int k = 0; k < 9; ++k) s_pf[(t - 1) * 9 + k] = R[k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) s_R[t * 9 + k] = R[k];
    }
    __syncthreads();

    // v_posed = v_template + beta.shapedirs + pose_feature.posedirs   (:586, :613)
    for (int e = t; e < NE; e += FWD_NT) {
        float acc = m.v_template[e];
#pragma unroll
        for (int s = 0; s < 10; ++s) acc = fmaf(s_beta[s], m.shapedirs[s * NE + e], acc);
#pragma unroll 27
        for (int j = 0; j < 135; ++j) acc = fmaf(s_pf[j], m.posedirs[j * NE + e], acc);     // 27 loads in flight per L2 round trip
        s_vp[e] = acc;
    }

    // kinematic chain (:730-770): G_i = G_parent . [R_i | J_i - J_parent]
    if (t < 12) {
        const int r = t >> 2, c = t & 3;
        s_G[t] = (c < 3) ? s_R[r * 3 + c] : s_J[r];
    }
    for (int i = 1; i < 16; ++i) {
        __syncthreads();
        if (t < 12) {
            const int r = t >> 2, c = t & 3;
            const int p = m.parents[i];
            const float* Gp = s_G + p * 12 + r * 4;
            float val;
            if (c < 3) {
                val = Gp[0] * s_R[i * 9 + c] + Gp[1] * s_R[i * 9 + 3 + c] + Gp[2] * s_R[i * 9 + 6 + c];
            } else {
                val = Gp[0] * (s_J[i * 3] - s_J[p * 3]) + Gp[1] * (s_J[i * 3 + 1] - s_J[p * 3 + 1]) +
                      Gp[2] * (s_J[i * 3 + 2] - s_J[p * 3 + 2]) + Gp[3];
            }
            s_G[i * 12 + t] = val;
        }
    }
    __syncthreads();
    if (t < 192) {
        const int i = t / 12, k = t % 12, r = k >> 2, c = k & 3;
        const float* G = s_G + i * 12 + r * 4;
        s_A[t] = (c < 3) ? G[c] : G[3] - (G[0] * s_J[i * 3] + G[1] * s_J[i * 3 + 1] + G[2] * s_J[i * 3 + 2]);
    }
    __syncthreads();

    // linear blend skinning (:619-629)
    for (int v = t; v < NV; v += FWD_NT) {
        float T[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = 0.f;
        const float4* wrow = reinterpret_cast<const float4*>(m.weights + v * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w4 = wrow[q];
            const float w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (w[u] != 0.f) {
                    const float* A = s_A + (q * 4 + u) * 12;
#pragma unroll
                    for (int k = 0; k < 12; ++k) T[k] = fmaf(w[u], A[k], T[k]);
                }
            }
        }
        const float x = s_vp[v * 3], y = s_vp[v * 3 + 1], z = s_vp[v * 3 + 2];
        s_v[v * 3 + 0] = T[0] * x + T[1] * y + T[2] * z + T[3];
        s_v[v * 3 + 1] = T[4] * x + T[5] * y + T[6] * z + T[7];
        s_v[v * 3 + 2] = T[8] * x + T[9] * y + T[10] * z + T[11];
    }
    __syncthreads();

    // joint regression from posed verts (:630-633, CSR of J_regressor^T) + wrist cap vertex (:636)
    if (t < 63) {
        const int j = t / 3, c = t % 3;
        float acc = 0.f;
        for (int k = m.jreg_rowptr[j]; k < m.jreg_rowptr[j + 1]; ++k) acc = fmaf(m.jreg_val[k], s_v[m.jreg_col[k] * 3 + c], acc);
        s_jnt[t] = acc;
    } else if (t >= 64 && t < 67) {
        const int c = t - 64;
        float acc = 0.f;
        for (int r = 0; r < 16; ++r) acc += s_v[m.wrist_ring[r] * 3 + c];
        s_v[NE + c] = acc / 16.f;
    }
    __syncthreads();

    float sc = 1.f, tr[3] = {0.f, 0.f, 0.f};
    if (cam) { sc = cam[b * scam]; tr[0] = cam[b * scam + 1]; tr[1] = cam[b * scam + 2]; tr[2] = cam[b * scam + 3]; }
    for (int e = t; e < NEO; e += FWD_NT) verts[(size_t)b * NEO + e] = ((s_v[e] * k1) * k2) * sc + tr[e % 3];
    if (t < 63) joints[b * 63 + t] = ((s_jnt[t] * k1) * k2) * sc + tr[t % 3];
    if (Rs_out && t < 135) Rs_out[b * 135 + t] = s_R[9 + t];
    if (save) {
        float* sv = save + (size_t)b * DSF_MANO_SAVE_FLOATS;
        for (int e = t; e < NE; e += FWD_NT) sv[SV_VPOSED + e] = s_vp[e];
        for (int e = t; e < NEO; e += FWD_NT) sv[SV_VERTS + e] = s_v[e];
        if (t < 63) sv[SV_JOINTS + t] = s_jnt[t];
        if (t < 192) sv[SV_G + t] = s_G[t];
        if (t < 144) sv[SV_RS + t] = s_R[t];
        if (t < 48) sv[SV_J + t] = s_J[t];
        if (t < 45) sv[SV_TH + t] = s_thf[t];
    }
}

// 1024 threads per sample as in the forward; cross-thread reductions keep a fixed order (per-wave partials summed
// wave 0..15, five vertex lanes per d/dA entry summed 0..4), so the gradients are deterministic.
constexpr int BWD_NT = 1024, BWD_NW = BWD_NT / 64, BWD_VL = 5, BWD_KPT = (NE + BWD_NT - 1) / BWD_NT;
__global__ __launch_bounds__(BWD_NT) void mano_bwd_kernel(dsf_mano_model m, const float* __restrict__ theta,
                                                       const float* __restrict__ rot,
                                                       const float* __restrict__ cam,
                                                       const float* __restrict__ save,
                                                       const float* __restrict__ gV, const float* __restrict__ gJ,
                                                       int ncomp, int rot_dim, int ps, float k1, float k2,
                                                       float* __restrict__ g_beta, float* __restrict__ g_theta,
                                                       float* __restrict__ g_rot, float* __restrict__ g_cam) {
    const int sb = ps ? ps : 10, st = ps ? ps : ncomp, sr = ps ? ps : rot_dim, scam = ps ? ps : 4;    // as in mano_fwd_kernel
    __shared__ float s_gv[NEO];
    __shared__ float s_gvp[NE];
    __shared__ float s_vp[NE];
    __shared__ float s_gj[63];
    __shared__ float s_G[192], s_R[144], s_J[48], s_thf[45];
    __shared__ float s_gA[192], s_gRg[144], s_gt[48], s_gJ16[48], s_gR[144], s_add[9], s_gd[3];
    __shared__ float s_part[135 * BWD_NW], s_partb[10 * BWD_NW], s_red[BWD_NW * 4], s_gthf[45], s_gAp[BWD_VL * 192];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* sv = save + (size_t)b * DSF_MANO_SAVE_FLOATS;
    const float sc = cam ? cam[b * scam] : 1.f;
    const float kk = k1 * k2;

    // ---- upstream grads, d/d(cam) ----
    float acc_s = 0.f, acc_t[3] = {0.f, 0.f, 0.f};
    for (int e = t; e < NEO; e += BWD_NT) {
        const float g = gV ? gV[(size_t)b * NEO + e] : 0.f;
        s_gv[e] = g * (kk * sc);
        acc_s += g * ((sv[SV_VERTS + e] * k1) * k2);
        const int c = e % 3;
        acc_t[0] += (c == 0) ? g : 0.f; acc_t[1] += (c == 1) ? g : 0.f; acc_t[2] += (c == 2) ? g : 0.f;
    }
    if (t < 63) {
        const float g = gJ ? gJ[b * 63 + t] : 0.f;
        s_gj[t] = g * (kk * sc);
        acc_s += g * ((sv[SV_JOINTS + t] * k1) * k2);
        const int c = t % 3;
        acc_t[0] += (c == 0) ? g : 0.f; acc_t[1] += (c == 1) ? g : 0.f; acc_t[2] += (c == 2) ? g : 0.f;
    }
    for (int e = t; e < NE; e += BWD_NT) s_vp[e] = sv[SV_VPOSED + e];
    if (t < 192) s_G[t] = sv[SV_G + t];
    if (t < 144) s_R[t] = sv[SV_RS + t];
    if (t < 48) s_J[t] = sv[SV_J + t];
    if (t < 45) s_thf[t] = sv[SV_TH + t];
    if (g_cam) {
        acc_s = wave_sum(acc_s);
        acc_t[0] = wave_sum(acc_t[0]); acc_t[1] = wave_sum(acc_t[1]); acc_t[2] = wave_sum(acc_t[2]);
        if (lane == 0) { s_red[wave * 4] = acc_s; s_red[wave * 4 + 1] = acc_t[0]; s_red[wave * 4 + 2] = acc_t[1]; s_red[wave * 4 + 3] = acc_t[2]; }
    }
    __syncthreads();
    if (g_cam && t < 4) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < BWD_NW; ++w) acc += s_red[w * 4 + t];
        g_cam[b * scam + t] = acc;
    }

    // ---- wrist cap (:636) and joint regression (:630-633) ----
    if (t < 3) {
        const float g = s_gv[NE + t] / 16.f;
        for (int r = 0; r < 16; ++r) s_gv[m.wrist_ring[r] * 3 + t] += g;
    }
    __syncthreads();
    for (int v = t; v < NV; v += BWD_NT) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int j = 0; j < 21; ++j) {
            const float w = m.j_regressor[v * 21 + j];
            if (w != 0.f) { a0 = fmaf(w, s_gj[j * 3], a0); a1 = fmaf(w, s_gj[j * 3 + 1], a1); a2 = fmaf(w, s_gj[j * 3 + 2], a2); }
        }
        s_gv[v * 3] += a0; s_gv[v * 3 + 1] += a1; s_gv[v * 3 + 2] += a2;
    }
    __syncthreads();

    // ---- skinning: d/d(v_posed) = Trot^T g ; d/dA_i = sum_v W[v,i] g_v (x) [vp_v;1] ----
    for (int v = t; v < NV; v += BWD_NT) {
        float T[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) T[k] = 0.f;
        for (int i = 0; i < 16; ++i) {
            const float w = m.weights[v * 16 + i];
            if (w != 0.f) {
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) T[r * 3 + c] = fmaf(w, s_G[i * 12 + r * 4 + c], T[r * 3 + c]);
            }
        }
        const float g0 = s_gv[v * 3], g1 = s_gv[v * 3 + 1], g2 = s_gv[v * 3 + 2];
        s_gvp[v * 3 + 0] = T[0] * g0 + T[3] * g1 + T[6] * g2;
        s_gvp[v * 3 + 1] = T[1] * g0 + T[4] * g1 + T[7] * g2;
        s_gvp[v * 3 + 2] = T[2] * g0 + T[5] * g1 + T[8] * g2;
    }
    if (t < 192 * BWD_VL) {                                      // 192 entries of d/dA x 5 vertex lanes
        const int o = t % 192, vl = t / 192;
        const int i = o & 15, k = o >> 4, r = k >> 2, c = k & 3;
        float acc = 0.f;
        for (int v = vl; v < NV; v += BWD_VL) {
            const float w = m.weights[v * 16 + i];
            const float x = (c < 3) ? s_vp[v * 3 + c] : 1.f;
            acc = fmaf(w * s_gv[v * 3 + r], x, acc);
        }
        s_gAp[vl * 192 + o] = acc;
    }
    __syncthreads();
    if (t < 192) {
        const int i = t & 15, k = t >> 4;
        float acc = s_gAp[t];
#pragma unroll
        for (int vl = 1; vl < BWD_VL; ++vl) acc += s_gAp[vl * 192 + t];
        s_gA[i * 12 + k] = acc;
    }
    __syncthreads();

    // ---- chain backward.  A_i = [Rg_i | t_i - Rg_i J_i] ----
    if (t < 144) {
        const int i = t / 9, r = (t % 9) / 3, c = t % 3;
        s_gRg[t] = s_gA[i * 12 + r * 4 + c] - s_gA[i * 12 + r * 4 + 3] * s_J[i * 3 + c];
        s_gR[t] = 0.f;
    } else if (t >= 160 && t < 208) {
        const int k = t - 160, i = k / 3, c = k % 3;
        s_gt[k] = s_gA[i * 12 + c * 4 + 3];
        s_gJ16[k] = -(s_G[i * 12 + c] * s_gA[i * 12 + 3] + s_G[i * 12 + 4 + c] * s_gA[i * 12 + 7] +
                      s_G[i * 12 + 8 + c] * s_gA[i * 12 + 11]);
    }
    __syncthreads();
    for (int i = 15; i >= 1; --i) {
        const int p = m.parents[i];
        if (t < 9) {
            const int r = t / 3, c = t % 3;
            s_gR[i * 9 + t] = s_G[p * 12 + r] * s_gRg[i * 9 + c] + s_G[p * 12 + 4 + r] * s_gRg[i * 9 + 3 + c] +
                              s_G[p * 12 + 8 + r] * s_gRg[i * 9 + 6 + c];
            s_add[t] = s_gRg[i * 9 + r * 3] * s_R[i * 9 + c * 3] + s_gRg[i * 9 + r * 3 + 1] * s_R[i * 9 + c * 3 + 1] +
                       s_gRg[i * 9 + r * 3 + 2] * s_R[i * 9 + c * 3 + 2] +
                       s_gt[i * 3 + r] * (s_J[i * 3 + c] - s_J[p * 3 + c]);
        } else if (t >= 64 && t < 67) {
            const int c = t - 64;
            s_gd[c] = s_G[p * 12 + c] * s_gt[i * 3] + s_G[p * 12 + 4 + c] * s_gt[i * 3 + 1] + s_G[p * 12 + 8 + c] * s_gt[i * 3 + 2];
        }
        __syncthreads();
        if (t < 9) {
            s_gRg[p * 9 + t] += s_add[t];
        } else if (t >= 64 && t < 67) {
            const int c = t - 64;
            s_gt[p * 3 + c] += s_gt[i * 3 + c];
            s_gJ16[i * 3 + c] += s_gd[c];
            s_gJ16[p * 3 + c] -= s_gd[c];
        }
        __syncthreads();
    }
    if (t < 9) s_gR[t] = s_gRg[t];
    else if (t >= 64 && t < 67) s_gJ16[t - 64] += s_gt[t - 64];

    // ---- blendshape reductions: g_pf[j] = <posedirs_j, g_vp>, g_beta[k] = <shapedirs_k, g_vp> ----
    float gk[BWD_KPT];
#pragma unroll
    for (int k = 0; k < BWD_KPT; ++k) { const int e = t + BWD_NT * k; gk[k] = (e < NE) ? s_gvp[e] : 0.f; }
    // rows are taken JB at a time with all their loads issued before the first reduction: the loop was one L2 round trip
    // (~0.6 us) per row, 145 rows deep (87 of the kernel's 123 us at B = 32); same arithmetic, same order
    constexpr int JB = 9, NBLK = 135 / JB;
    static_assert(135 % JB == 0 && NBLK % 2 == 1, "row blocking");
    float va[JB][BWD_KPT], vb[JB][BWD_KPT];              // two row blocks: the next one is in flight while this one is reduced
    auto load_block = [&](int j0, float (&v)[JB][BWD_KPT]) {
#pragma unroll
        for (int u = 0; u < JB; ++u)
#pragma unroll
            for (int k = 0; k < BWD_KPT; ++k) { const int e = t + BWD_NT * k; v[u][k] = (e < NE) ? m.posedirs[(j0 + u) * NE + e] : 0.f; }
    };
    auto reduce_block = [&](int j0, const float (&v)[JB][BWD_KPT]) {
#pragma unroll
        for (int u = 0; u < JB; ++u) {
            float p = 0.f;
#pragma unroll
            for (int k = 0; k < BWD_KPT; ++k) p = fmaf(v[u][k], gk[k], p);
            p = wave_sum(p);
            if (lane == 0) s_part[(j0 + u) * BWD_NW + wave] = p;
        }
    };
    load_block(0, va);
    for (int blk = 0; blk + 2 < NBLK; blk += 2) {
        load_block((blk + 1) * JB, vb);
        reduce_block(blk * JB, va);
        load_block((blk + 2) * JB, va);
        reduce_block((blk + 1) * JB, vb);
    }
    reduce_block((NBLK - 1) * JB, va);
    {
        float v[10][BWD_KPT];
#pragma unroll
        for (int s = 0; s < 10; ++s)
#pragma unroll
            for (int k = 0; k < BWD_KPT; ++k) { const int e = t + BWD_NT * k; v[s][k] = (e < NE) ? m.shapedirs[s * NE + e] : 0.f; }
#pragma unroll
        for (int s = 0; s < 10; ++s) {
            float p = 0.f;
#pragma unroll
            for (int k = 0; k < BWD_KPT; ++k) p = fmaf(v[s][k], gk[k], p);
            p = wave_sum(p);
            if (lane == 0) s_partb[s * BWD_NW + wave] = p;
        }
    }
    __syncthreads();
    if (t < 135) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < BWD_NW; ++w) acc += s_part[t * BWD_NW + w];
        s_gR[9 + t] += acc;
    } else if (t >= 192 && t < 202) {
        const int s = t - 192;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < BWD_NW; ++w) acc += s_partb[s * BWD_NW + w];
        for (int k = 0; k < 48; ++k) acc = fmaf(m.j_shapedirs[s * 48 + k], s_gJ16[k], acc);
        g_beta[b * sb + s] = acc;
    }
    __syncthreads();

    // ---- Rodrigues / quaternion backward, PCA projection ----
    if (t < 16) {
        if (t == 0) {
            float g[4];
            if (rot_dim == 3) {
                rodrigues_bwd(rot + b * sr, s_gR, g);
                g_rot[b * sr] = g[0]; g_rot[b * sr + 1] = g[1]; g_rot[b * sr + 2] = g[2];
            } else {
                quat_bwd(rot + b * sr, s_gR, g);
                g_rot[b * sr] = g[0]; g_rot[b * sr + 1] = g[1]; g_rot[b * sr + 2] = g[2]; g_rot[b * sr + 3] = g[3];
            }
        } else {
            rodrigues_bwd(s_thf + (t - 1) * 3, s_gR + t * 9, s_gthf + (t - 1) * 3);
        }
    }
    __syncthreads();
    if (t < ncomp) {
        float acc = 0.f;
        for (int k = 0; k < 45; ++k) acc = fmaf(m.hands_comp[t * 45 + k], s_gthf[k], acc);
        g_theta[b * st + t] = acc;
    }
}

}  // namespace

extern "C" int dsf_mano_forward_v1(const dsf_mano_model* m, const float* beta, const float* theta, const float* rot,
                                const float* cam, int B, int ncomp, int rot_dim, int param_stride, float k1, float k2,
                                float* verts, float* joints, float* Rs, float* save, dsf_stream_t stream) {
    DSF_CHECK_ARG(m && beta && theta && rot && verts && joints);
    DSF_CHECK_ARG(B >= 0 && ncomp >= 0 && ncomp <= 45 && (rot_dim == 3 || rot_dim == 4) && param_stride >= 0);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(mano_fwd_kernel, dim3(B), dim3(FWD_NT), 0, (hipStream_t)stream, *m, beta, theta, rot, cam, ncomp,
                       rot_dim, param_stride, k1, k2, verts, joints, Rs, save);
    return dsf_launch_status();
}

extern "C" int dsf_mano_backward_v1(const dsf_mano_model* m, const float* theta, const float* rot, const float* cam,
                                 const float* save, const float* grad_verts, const float* grad_joints, int B,
                                 int ncomp, int rot_dim, int param_stride, float k1, float k2, float* grad_beta,
                                 float* grad_theta, float* grad_rot, float* grad_cam, dsf_stream_t stream) {
    DSF_CHECK_ARG(m && rot && save && grad_beta && grad_theta && grad_rot);
    DSF_CHECK_ARG(B >= 0 && ncomp >= 0 && ncomp <= 45 && (rot_dim == 3 || rot_dim == 4) && param_stride >= 0);
    DSF_CHECK_ARG(!(grad_cam && !cam));
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(mano_bwd_kernel, dim3(B), dim3(BWD_NT), 0, (hipStream_t)stream, *m, theta, rot, cam, save,
                       grad_verts, grad_joints, ncomp, rot_dim, param_stride, k1, k2, grad_beta, grad_theta, grad_rot, grad_cam);
    return dsf_launch_status();
}
