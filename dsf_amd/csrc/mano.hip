// K5: MANO layer forward / backward for gfx950.
//
// Every kernel is a 256-thread workgroup (4 waves, <= 128 VGPRs, <= 36 KB of LDS) so that it can take the free wave slot
// beside the two 174-VGPR convolution workgroups a CU holds during the backward pass; rounds 1-3 ran one 1024-thread
// workgroup per sample, which needs a whole drained CU: 84 us alone became 700 us inside the step.
//
// The blendshapes are BATCHED: v_posed = v_template + beta.shapedirs + pose_feature.posedirs is a (B x 145) . (145 x 2334)
// product, so a workgroup owns 256 columns for 8 samples and reads each 148 KB column block of the 1.4 MB constants
// once per 8 samples (one workgroup per sample read all of them per sample); its transpose in the backward pass,
// g_pose_feature = g_v_posed . posedirs^T, walks the rows once per sample (per pair of samples above 512).  Skinning, the kinematic chain
// (12 lanes per joint) and joint regression stay per sample with everything in LDS.  Per output the order of the float
// operations is the one of the 1024-thread kernels: forward results are bit-identical to theirs.
//
// Reference: MANO_SMPL.forward / get_mano_vertices / batch_rodrigues / quat2mat /
// batch_global_rigid_transformation, render_model/mano_layer.py:573-770.
#include "common.h"

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


constexpr int NT = 256;             // threads of every workgroup here
constexpr int BL_SB = 8;            // samples per blendshape workgroup (forward)
// backward scratch per sample: d/d(v_posed) | d/dR of the 16 joints from the chain (blendshape part not yet added) | d/dJ
constexpr int SC_GVP = 0, SC_GR = 2334, SC_GJ = 2478;
static_assert(SC_GJ + 48 <= DSF_MANO_BWD_SCRATCH_FLOATS, "scratch layout");

// ---- forward 1/2: pose features + blendshapes for BL_SB samples x 256 columns -----------------------------------------
// grid (ceil(2334 / 256), ceil(B / BL_SB)).  Column block 0 also leaves the per-sample pose state (full pose, the 16
// rotations, rest joints) in `save` for the skinning kernel and the backward pass.
__global__ __launch_bounds__(NT, 3) void mano_blend_kernel(dsf_mano_model m, const float* __restrict__ beta,
                                                        const float* __restrict__ theta, const float* __restrict__ rot,
                                                        int B, int ncomp, int rot_dim, int ps, float* __restrict__ save) {
    __shared__ float s_bs[10 * BL_SB];              // [shape][sample]
    __shared__ float s_theta[BL_SB * 45], s_thf[BL_SB * 45];
    __shared__ __attribute__((aligned(16))) float s_pf[135 * BL_SB];      // [pose feature][sample]
    const int t = threadIdx.x, b0 = blockIdx.y * BL_SB;
    const int nb = (B - b0 < BL_SB) ? B - b0 : BL_SB;
    const bool first = blockIdx.x == 0;
    // ps: floats between consecutive samples of beta / theta / rot / cam (0 = each array tightly packed); the four
    // pointers may be column offsets into one (B, 62) parameter matrix
    const int sb_ = ps ? ps : 10, st = ps ? ps : ncomp, sr = ps ? ps : rot_dim;
    for (int i = t; i < BL_SB * 10; i += NT) {
        const int q = i / 10, s = i % 10;
        s_bs[s * BL_SB + q] = (q < nb) ? beta[(size_t)(b0 + q) * sb_ + s] : 0.f;
    }
    for (int i = t; i < BL_SB * 45; i += NT) {
        const int q = i / 45, c = i % 45;
        s_theta[i] = (q < nb && c < ncomp) ? theta[(size_t)(b0 + q) * st + c] : 0.f;
    }
    __syncthreads();
    // full pose = theta . comp[:ncomp] + mean (:601)
    for (int i = t; i < BL_SB * 45; i += NT) {
        const int q = i / 45, k = i % 45;
        float acc = 0.f;
        for (int c = 0; c < ncomp; ++c) acc = fmaf(s_theta[q * 45 + c], m.hands_comp[c * 45 + k], acc);
        acc += m.hands_mean[k];
        s_thf[i] = acc;
        if (first && q < nb) save[(size_t)(b0 + q) * DSF_MANO_SAVE_FLOATS + SV_TH + k] = acc;
    }
    __syncthreads();
    if (t < 15 * BL_SB) {                           // Rodrigues of the 15 articulated joints
        const int q = t / 15, j = t % 15;
        float R[9];
        rodrigues(s_thf + q * 45 + j * 3, R);
#pragma unroll
        for (int k = 0; k < 9; ++k) s_pf[(j * 9 + k) * BL_SB + q] = R[k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f);
        if (first && q < nb) {
#pragma unroll
            for (int k = 0; k < 9; ++k) save[(size_t)(b0 + q) * DSF_MANO_SAVE_FLOATS + SV_RS + 9 + j * 9 + k] = R[k];
        }
    } else if (first && t >= 128 && t < 128 + nb) { // root rotation: axis-angle or quaternion
        const int q = t - 128;
        const float* r = rot + (size_t)(b0 + q) * sr;
        float R[9];
        if (rot_dim == 3) {
            const float th[3] = {r[0], r[1], r[2]};
            rodrigues(th, R);
        } else {
            const float nq = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
            quat_to_rot(r[0] / nq, r[1] / nq, r[2] / nq, r[3] / nq, R);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) save[(size_t)(b0 + q) * DSF_MANO_SAVE_FLOATS + SV_RS + k] = R[k];
    } else if (first && t >= 192 && t < 192 + 48) { // rest joints J = J_template + beta . J_shapedirs
        const int k = t - 192;
        for (int q = 0; q < nb; ++q) {
            float acc = m.j_template[k];
            for (int s = 0; s < 10; ++s) acc = fmaf(s_bs[s * BL_SB + q], m.j_shapedirs[s * 48 + k], acc);
            save[(size_t)(b0 + q) * DSF_MANO_SAVE_FLOATS + SV_J + k] = acc;
        }
    }
    __syncthreads();

    // v_posed = v_template + beta.shapedirs + pose_feature.posedirs   (:586, :613); one column, BL_SB samples per thread
    const int e = blockIdx.x * NT + t;
    if (e >= NE) return;
    float acc[BL_SB];
    const float vt = m.v_template[e];
#pragma unroll
    for (int q = 0; q < BL_SB; ++q) acc[q] = vt;
#pragma unroll
    for (int s = 0; s < 10; ++s) {
        const float d = m.shapedirs[s * NE + e];
#pragma unroll
        for (int q = 0; q < BL_SB; ++q) acc[q] = fmaf(s_bs[s * BL_SB + q], d, acc[q]);
    }
#pragma unroll 27
    for (int j = 0; j < 135; ++j) {                 // 27 loads in flight per L2 round trip
        const float d = m.posedirs[j * NE + e];
        const float4 p0 = *reinterpret_cast<const float4*>(s_pf + j * BL_SB), p1 = *reinterpret_cast<const float4*>(s_pf + j * BL_SB + 4);
        acc[0] = fmaf(p0.x, d, acc[0]); acc[1] = fmaf(p0.y, d, acc[1]); acc[2] = fmaf(p0.z, d, acc[2]); acc[3] = fmaf(p0.w, d, acc[3]);
        acc[4] = fmaf(p1.x, d, acc[4]); acc[5] = fmaf(p1.y, d, acc[5]); acc[6] = fmaf(p1.z, d, acc[6]); acc[7] = fmaf(p1.w, d, acc[7]);
    }
#pragma unroll
    for (int q = 0; q < BL_SB; ++q)
        if (q < nb) save[(size_t)(b0 + q) * DSF_MANO_SAVE_FLOATS + SV_VPOSED + e] = acc[q];
}
static_assert(BL_SB == 8, "the pose-feature reads above are two float4");

// ---- forward 2/2: kinematic chain, skinning, joint regression, wrist vertex, output affine; one workgroup per sample ----
__global__ __launch_bounds__(NT, 3) void mano_skin_kernel(dsf_mano_model m, const float* __restrict__ cam, int ps, float k1,
                                                       float k2, float* __restrict__ verts, float* __restrict__ joints,
                                                       float* __restrict__ Rs_out, float* __restrict__ save) {
    __shared__ float s_vp[NE];
    __shared__ float s_v[NEO];
    __shared__ float s_R[16 * 9], s_J[48], s_G[16 * 12], s_A[16 * 12], s_jnt[63];
    const int b = blockIdx.x, t = threadIdx.x;
    const int scam = ps ? ps : 4;
    float* sv = save + (size_t)b * DSF_MANO_SAVE_FLOATS;
#pragma unroll 2
    for (int e = t; e < NE; e += NT) s_vp[e] = sv[SV_VPOSED + e];
    if (t < 144) s_R[t] = sv[SV_RS + t];
    else if (t >= 192 && t < 240) s_J[t - 192] = sv[SV_J + t - 192];
    __syncthreads();

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
#pragma unroll 1
    for (int v = t; v < NV; v += NT) {
        asm volatile("" ::: "memory");              // keeps the 192 floats of s_A in LDS (hoisted out of this loop they are 192 VGPRs)
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
    if (cam) { sc = cam[(size_t)b * scam]; tr[0] = cam[(size_t)b * scam + 1]; tr[1] = cam[(size_t)b * scam + 2]; tr[2] = cam[(size_t)b * scam + 3]; }
#pragma unroll 2
    for (int e = t; e < NEO; e += NT) verts[(size_t)b * NEO + e] = ((s_v[e] * k1) * k2) * sc + tr[e % 3];
    if (t < 63) joints[b * 63 + t] = ((s_jnt[t] * k1) * k2) * sc + tr[t % 3];
    if (Rs_out && t < 135) Rs_out[b * 135 + t] = s_R[9 + t];
#pragma unroll 2
    for (int e = t; e < NEO; e += NT) sv[SV_VERTS + e] = s_v[e];
    if (t < 63) sv[SV_JOINTS + t] = s_jnt[t];
    if (t < 192) sv[SV_G + t] = s_G[t];
}

// ---- backward 1/2, one workgroup per sample: output affine, joint regression, skinning and the kinematic chain --------
// Leaves d/d(v_posed), the chain's d/dR and d/dJ in `scratch`; writes g_cam.  Cross-thread reductions keep a fixed order
// (per-wave partials summed wave 0..3, sixteen vertex lanes per d/dA entry summed 0..15), so the gradients are deterministic.
constexpr int SK_VL = 16;
__global__ __launch_bounds__(NT, 3) void mano_skin_bwd_kernel(dsf_mano_model m, const float* __restrict__ cam,
                                                           const float* __restrict__ save, const float* __restrict__ gV,
                                                           const float* __restrict__ gJ, int ps, float k1, float k2,
                                                           float* __restrict__ g_cam, float* __restrict__ scratch) {
    const int scam = ps ? ps : 4;
    __shared__ float s_gv[NEO];
    __shared__ float s_vp[NE];
    __shared__ float s_gj[63];
    __shared__ float s_G[192], s_R[144], s_J[48];
    __shared__ float s_gA[192], s_gRg[144], s_gt[48], s_gJ16[48], s_gR[144], s_add[9], s_gd[3];
    __shared__ float s_red[4 * 4], s_gAp[SK_VL * 192];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* sv = save + (size_t)b * DSF_MANO_SAVE_FLOATS;
    float* sc_out = scratch + (size_t)b * DSF_MANO_BWD_SCRATCH_FLOATS;
    const float sc = cam ? cam[(size_t)b * scam] : 1.f;
    const float kk = k1 * k2;

    // ---- upstream grads, d/d(cam) ----
    float acc_s = 0.f, acc_t[3] = {0.f, 0.f, 0.f};
#pragma unroll 2
    for (int e = t; e < NEO; e += NT) {
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
#pragma unroll 2
    for (int e = t; e < NE; e += NT) s_vp[e] = sv[SV_VPOSED + e];
    if (t < 192) s_G[t] = sv[SV_G + t];
    if (t < 144) s_R[t] = sv[SV_RS + t];
    if (t < 48) s_J[t] = sv[SV_J + t];
    if (g_cam) {
        acc_s = wave_sum(acc_s);
        acc_t[0] = wave_sum(acc_t[0]); acc_t[1] = wave_sum(acc_t[1]); acc_t[2] = wave_sum(acc_t[2]);
        if (lane == 0) { s_red[wave * 4] = acc_s; s_red[wave * 4 + 1] = acc_t[0]; s_red[wave * 4 + 2] = acc_t[1]; s_red[wave * 4 + 3] = acc_t[2]; }
    }
    __syncthreads();
    if (g_cam && t < 4) g_cam[(size_t)b * scam + t] = ((s_red[t] + s_red[4 + t]) + s_red[8 + t]) + s_red[12 + t];

    // ---- wrist cap (:636) and joint regression (:630-633) ----
    if (t < 3) {
        const float g = s_gv[NE + t] / 16.f;
        for (int r = 0; r < 16; ++r) s_gv[m.wrist_ring[r] * 3 + t] += g;
    }
    __syncthreads();
#pragma unroll 1
    for (int v = t; v < NV; v += NT) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int j = 0; j < 21; ++j) {
            const float w = m.j_regressor[v * 21 + j];
            if (w != 0.f) { a0 = fmaf(w, s_gj[j * 3], a0); a1 = fmaf(w, s_gj[j * 3 + 1], a1); a2 = fmaf(w, s_gj[j * 3 + 2], a2); }
        }
        s_gv[v * 3] += a0; s_gv[v * 3 + 1] += a1; s_gv[v * 3 + 2] += a2;
    }
    __syncthreads();

    // ---- skinning: d/d(v_posed) = Trot^T g ; d/dA_i = sum_v W[v,i] g_v (x) [vp_v;1] ----
#pragma unroll 1
    for (int v = t; v < NV; v += NT) {
        asm volatile("" ::: "memory");              // as in mano_skin_kernel: s_G stays in LDS
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
        sc_out[SC_GVP + v * 3 + 0] = T[0] * g0 + T[3] * g1 + T[6] * g2;
        sc_out[SC_GVP + v * 3 + 1] = T[1] * g0 + T[4] * g1 + T[7] * g2;
        sc_out[SC_GVP + v * 3 + 2] = T[2] * g0 + T[5] * g1 + T[8] * g2;
    }
    {                                                           // joint i = t & 15, vertex lane t >> 4: 12 entries of d/dA_i each
        const int i = t & 15, vl = t >> 4;
        float a[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) a[k] = 0.f;
#pragma unroll 2
        for (int v = vl; v < NV; v += SK_VL) {
            const float w = m.weights[v * 16 + i];
            if (w != 0.f) {
                const float x[4] = {s_vp[v * 3], s_vp[v * 3 + 1], s_vp[v * 3 + 2], 1.f};
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float wg = w * s_gv[v * 3 + r];
#pragma unroll
                    for (int c = 0; c < 4; ++c) a[r * 4 + c] = fmaf(wg, x[c], a[r * 4 + c]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) s_gAp[vl * 192 + k * 16 + i] = a[k];
    }
    __syncthreads();
    if (t < 192) {
        const int i = t & 15, k = t >> 4;
        float acc = s_gAp[t];
#pragma unroll
        for (int vl = 1; vl < SK_VL; ++vl) acc += s_gAp[vl * 192 + t];
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
    __syncthreads();
    if (t < 144) sc_out[SC_GR + t] = s_gR[t];
    else if (t >= 192 && t < 240) sc_out[SC_GJ + t - 192] = s_gJ16[t - 192];
}

// ---- backward 2/2: blendshape reductions g_pf[j] = <posedirs_j, g_vp>, g_beta[k] = <shapedirs_k, g_vp> for BB_SB samples
// per workgroup (every row of the constants is read once per BB_SB samples; the per-sample arithmetic does not depend on BB_SB), then the Rodrigues / quaternion backward and the
// PCA projection of those samples.  Per row: KPT products per thread, a wave sum, the four waves added in order.
constexpr int BB_KPT = (NE + NT - 1) / NT, BB_ROWS = 145, BB_JB = 5, BB_NBLK = BB_ROWS / BB_JB;
static_assert(BB_ROWS % BB_JB == 0, "row blocking");
template <int BB_SB>      // samples per workgroup: 1 (a workgroup per sample: the shortest launch while B workgroups fit the chip), or 2
__global__ __launch_bounds__(NT, 3) void mano_blend_bwd_kernel(dsf_mano_model m, const float* __restrict__ rot,
                                                            const float* __restrict__ save, const float* __restrict__ scratch,
                                                            int B, int ncomp, int rot_dim, int ps,
                                                            float* __restrict__ g_beta, float* __restrict__ g_theta,
                                                            float* __restrict__ g_rot) {
    const int sb_ = ps ? ps : 10, st = ps ? ps : ncomp, sr = ps ? ps : rot_dim;
    __shared__ float s_part[BB_ROWS * BB_SB * 4];
    __shared__ float s_gR[BB_SB * 144], s_gJ16[BB_SB * 48], s_thf[BB_SB * 45], s_gthf[BB_SB * 45];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b0 = blockIdx.x * BB_SB;
    const int nb = (B - b0 < BB_SB) ? B - b0 : BB_SB;

    float gk[BB_SB][BB_KPT];
#pragma unroll
    for (int q = 0; q < BB_SB; ++q)
#pragma unroll
        for (int k = 0; k < BB_KPT; ++k) {
            const int e = t + NT * k;
            gk[q][k] = (e < NE && q < nb) ? scratch[(size_t)(b0 + q) * DSF_MANO_BWD_SCRATCH_FLOATS + SC_GVP + e] : 0.f;
        }
    for (int i = t; i < BB_SB * 144; i += NT) { const int q = i / 144; s_gR[i] = (q < nb) ? scratch[(size_t)(b0 + q) * DSF_MANO_BWD_SCRATCH_FLOATS + SC_GR + i % 144] : 0.f; }
    for (int i = t; i < BB_SB * 48; i += NT) { const int q = i / 48; s_gJ16[i] = (q < nb) ? scratch[(size_t)(b0 + q) * DSF_MANO_BWD_SCRATCH_FLOATS + SC_GJ + i % 48] : 0.f; }
    for (int i = t; i < BB_SB * 45; i += NT) { const int q = i / 45; s_thf[i] = (q < nb) ? save[(size_t)(b0 + q) * DSF_MANO_SAVE_FLOATS + SV_TH + i % 45] : 0.f; }

    // rows 0..134 = posedirs, 135..144 = shapedirs; BB_JB rows per block, the next block in flight while this one is reduced
    float va[BB_JB][BB_KPT], vb[BB_JB][BB_KPT];
    auto load_block = [&](int blk, float (&v)[BB_JB][BB_KPT]) {
#pragma unroll
        for (int u = 0; u < BB_JB; ++u) {
            const int r = blk * BB_JB + u;
            const float* row = (r < 135) ? m.posedirs + (size_t)r * NE : m.shapedirs + (size_t)(r - 135) * NE;
#pragma unroll
            for (int k = 0; k < BB_KPT; ++k) { const int e = t + NT * k; v[u][k] = (e < NE) ? row[e] : 0.f; }
        }
    };
    auto reduce_block = [&](int blk, const float (&v)[BB_JB][BB_KPT]) {
#pragma unroll
        for (int u = 0; u < BB_JB; ++u)
#pragma unroll
            for (int q = 0; q < BB_SB; ++q) {
                float p = 0.f;
#pragma unroll
                for (int k = 0; k < BB_KPT; ++k) p = fmaf(v[u][k], gk[q][k], p);
                p = wave_sum(p);
                if (lane == 0) s_part[((blk * BB_JB + u) * BB_SB + q) * 4 + wave] = p;
            }
    };
    load_block(0, va);
    for (int blk = 0; blk < BB_NBLK; blk += 2) {
        if (blk + 1 < BB_NBLK) load_block(blk + 1, vb);
        reduce_block(blk, va);
        if (blk + 2 < BB_NBLK) load_block(blk + 2, va);
        if (blk + 1 < BB_NBLK) reduce_block(blk + 1, vb);
    }
    __syncthreads();
    for (int i = t; i < BB_SB * 145; i += NT) {
        const int q = i / 145, r = i % 145;
        const float* pp = s_part + (r * BB_SB + q) * 4;
        float acc = ((pp[0] + pp[1]) + pp[2]) + pp[3];
        if (r < 135) {
            s_gR[q * 144 + 9 + r] += acc;
        } else if (q < nb) {
            const int s = r - 135;
            for (int k = 0; k < 48; ++k) acc = fmaf(m.j_shapedirs[s * 48 + k], s_gJ16[q * 48 + k], acc);
            g_beta[(size_t)(b0 + q) * sb_ + s] = acc;
        }
    }
    __syncthreads();

    // ---- Rodrigues / quaternion backward, PCA projection ----
    if (t < 16 * BB_SB) {
        const int q = t >> 4, j = t & 15;
        if (q < nb) {
            if (j == 0) {
                const float* r = rot + (size_t)(b0 + q) * sr;
                float* go = g_rot + (size_t)(b0 + q) * sr;
                float g[4];
                if (rot_dim == 3) {
                    rodrigues_bwd(r, s_gR + q * 144, g);
                    go[0] = g[0]; go[1] = g[1]; go[2] = g[2];
                } else {
                    quat_bwd(r, s_gR + q * 144, g);
                    go[0] = g[0]; go[1] = g[1]; go[2] = g[2]; go[3] = g[3];
                }
            } else {
                rodrigues_bwd(s_thf + q * 45 + (j - 1) * 3, s_gR + q * 144 + j * 9, s_gthf + q * 45 + (j - 1) * 3);
            }
        }
    }
    __syncthreads();
    if (t < 64 * BB_SB) {
        const int q = t >> 6, c = t & 63;
        if (q < nb && c < ncomp) {
            float acc = 0.f;
            for (int k = 0; k < 45; ++k) acc = fmaf(m.hands_comp[c * 45 + k], s_gthf[q * 45 + k], acc);
            g_theta[(size_t)(b0 + q) * st + c] = acc;
        }
    }
}

}  // namespace

extern "C" int dsf_mano_forward(const dsf_mano_model* m, const float* beta, const float* theta, const float* rot,
                                const float* cam, int B, int ncomp, int rot_dim, int param_stride, float k1, float k2,
                                float* verts, float* joints, float* Rs, float* save, dsf_stream_t stream) {
    DSF_CHECK_ARG(m && beta && theta && rot && verts && joints && save);
    DSF_CHECK_ARG(B >= 0 && ncomp >= 0 && ncomp <= 45 && (rot_dim == 3 || rot_dim == 4) && param_stride >= 0);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(mano_blend_kernel, dim3((NE + NT - 1) / NT, (B + BL_SB - 1) / BL_SB), dim3(NT), 0, (hipStream_t)stream,
                       *m, beta, theta, rot, B, ncomp, rot_dim, param_stride, save);
    hipLaunchKernelGGL(mano_skin_kernel, dim3(B), dim3(NT), 0, (hipStream_t)stream, *m, cam, param_stride, k1, k2, verts,
                       joints, Rs, save);
    return dsf_launch_status();
}

extern "C" int dsf_mano_backward(const dsf_mano_model* m, const float* theta, const float* rot, const float* cam,
                                 const float* save, const float* grad_verts, const float* grad_joints, int B,
                                 int ncomp, int rot_dim, int param_stride, float k1, float k2, float* grad_beta,
                                 float* grad_theta, float* grad_rot, float* grad_cam, float* scratch, dsf_stream_t stream) {
    DSF_CHECK_ARG(m && rot && save && grad_beta && grad_theta && grad_rot && scratch);
    DSF_CHECK_ARG(B >= 0 && ncomp >= 0 && ncomp <= 45 && (rot_dim == 3 || rot_dim == 4) && param_stride >= 0);
    DSF_CHECK_ARG(!(grad_cam && !cam));
    (void)theta;                                                        // the full pose is part of `save`
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(mano_skin_bwd_kernel, dim3(B), dim3(NT), 0, (hipStream_t)stream, *m, cam, save, grad_verts,
                       grad_joints, param_stride, k1, k2, grad_cam, scratch);
    // one sample per workgroup up to 512 samples (the backward pair alone at B = 32: 123 -> 90 us; each workgroup streams the
    // L2-resident constants once), pairs beyond
    if (B <= 512)
        hipLaunchKernelGGL(mano_blend_bwd_kernel<1>, dim3(B), dim3(NT), 0, (hipStream_t)stream, *m, rot, save, scratch, B, ncomp,
                           rot_dim, param_stride, grad_beta, grad_theta, grad_rot);
    else
        hipLaunchKernelGGL(mano_blend_bwd_kernel<2>, dim3((B + 1) / 2), dim3(NT), 0, (hipStream_t)stream, *m, rot, save, scratch, B,
                           ncomp, rot_dim, param_stride, grad_beta, grad_theta, grad_rot);
    return dsf_launch_status();
}
