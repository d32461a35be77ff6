// K6/K7: sphere hand model for gfx950: joint radii (mean of the 10 nearest owned vertices),
// 66 sphere centres/radii, pairwise collision hinge with the per-row 0.1 gate, and the
// point-cloud part labels.  One 256-thread workgroup per sample; the top-10 selection is a
// wave-level iterative argmin (one wave per joint, shuffles instead of torch.topk's sort).
//
// Reference: MANO_SMPL.get_sphere_radius / calculate_coll / seg_pcl,
// render_model/mano_layer.py:271-317, 373-386, 404-426.
#include "common.h"

namespace {

constexpr int NS = 66, NPALM = 21, TOPK = 10;      // (21 joints)
__constant__ int c_child[15] = {2, 3, 16, 5, 6, 17, 8, 9, 18, 11, 12, 19, 14, 15, 20};
__constant__ int c_knuckle[5] = {1, 4, 7, 10, 13};

// joints (21x3) in LDS, joint radii (21) in LDS -> sphere s centre/radius
__device__ __forceinline__ void sphere_from_joints(const dsf_sphere_model& sm, const float* J, const float* jr,
                                                   float r_root, int s, float* c, float& r) {
    if (s == 0) {
        c[0] = J[0]; c[1] = J[1]; c[2] = J[2]; r = r_root;
    } else if (s < NPALM) {
        const int k = (s - 1) >> 2, m = (s - 1) & 3, kn = c_knuckle[k];
        const float tt = sm.t_palm[m];
        r = (jr[kn] - r_root) * tt + r_root;
#pragma unroll
        for (int a = 0; a < 3; ++a) c[a] = (J[kn * 3 + a] - J[a]) * tt + J[a];
    } else {
        const int i = (s - NPALM) / 3, m = (s - NPALM) % 3, ch = c_child[i], pa = i + 1;
        const float tt = sm.t_finger[m];
        r = (jr[ch] - jr[pa]) * tt + jr[pa];
#pragma unroll
        for (int a = 0; a < 3; ++a) c[a] = (J[ch * 3 + a] - J[pa * 3 + a]) * tt + J[pa * 3 + a];
    }
}

// per-sample: s_jr[21] joint radii (tips = parent/1.5), topk ids (optional)
__device__ void joint_radii(const dsf_sphere_model& sm, const float* __restrict__ joints_b,
                            const float* __restrict__ mesh_b, float* s_J, float* s_jr, int32_t* topk_b) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t < 63) s_J[t] = joints_b[t];
    __syncthreads();
    for (int j = wave; j < 16; j += 4) {
        // each lane: candidates v = lane + 64k; non-owned vertices count as distance 100 (:279)
        float cand[13];
        const float jx = s_J[j * 3], jy = s_J[j * 3 + 1], jz = s_J[j * 3 + 2];
#pragma unroll
        for (int k = 0; k < 13; ++k) {
            const int v = lane + 64 * k;
            float d = INFINITY;
            if (v < 778) {
                const float dx = jx - mesh_b[v * 3], dy = jy - mesh_b[v * 3 + 1], dz = jz - mesh_b[v * 3 + 2];
                d = sm.jreg_mask[j * 778 + v] ? sqrtf(dx * dx + dy * dy + dz * dz + 1e-8f) : 100.0f;
            }
            cand[k] = d;
        }
        float sum = 0.f;
        for (int r = 0; r < TOPK; ++r) {
            float best = cand[0];
            int bk = 0;
#pragma unroll
            for (int k = 1; k < 13; ++k) if (cand[k] < best) { best = cand[k]; bk = k; }
            int bv = lane + 64 * bk;
            float wb = best;
            int wv = bv;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(wb, o, 64);
                const int ov = __shfl_xor(wv, o, 64);
                if (ob < wb || (ob == wb && ov < wv)) { wb = ob; wv = ov; }
            }
            sum += wb;
            if (wv == bv) {
#pragma unroll
                for (int k = 0; k < 13; ++k) if (k == bk) cand[k] = INFINITY;
            }
            if (lane == 0 && topk_b) topk_b[j * TOPK + r] = (wb == 100.0f && !sm.jreg_mask[j * 778 + wv]) ? -1 : wv;
        }
        if (lane == 0) s_jr[j] = sum / (float)TOPK;
    }
    __syncthreads();
    if (t < 5) s_jr[16 + t] = s_jr[3 + 3 * t] / 1.5f;
    __syncthreads();
}

__global__ __launch_bounds__(256) void sphere_set_kernel(dsf_sphere_model sm, const float* __restrict__ joints,
                                                         const float* __restrict__ mesh, int V,
                                                         float* __restrict__ centres, float* __restrict__ radii,
                                                         int32_t* __restrict__ topk) {
    __shared__ float s_J[63], s_jr[21];
    const int b = blockIdx.x, t = threadIdx.x;
    joint_radii(sm, joints + b * 63, mesh + (int64_t)b * V * 3, s_J, s_jr, topk ? topk + b * 16 * TOPK : nullptr);
    if (t < NS) {
        const float r_root = fminf(fmaxf(s_jr[0] - 0.05f, 0.01f), 0.4f);
        float c[3], r;
        sphere_from_joints(sm, s_J, s_jr, r_root, t, c, r);
        radii[b * NS + t] = r;
        centres[(b * NS + t) * 3] = c[0]; centres[(b * NS + t) * 3 + 1] = c[1]; centres[(b * NS + t) * 3 + 2] = c[2];
    }
}

// seg_pcl's sphere set (mano_layer.py:404-413): CENTRES from one skeleton (the pixel branch's joints), RADII from another (the MANO
// branch's joints, whose distances to the mesh define them).  The reference evaluates get_sphere_radius twice and keeps half of
// each result; centres need no radii, so one top-10 selection serves: one launch instead of two.
__global__ __launch_bounds__(256) void sphere_mixed_kernel(dsf_sphere_model sm, const float* __restrict__ joints_c,
                                                           const float* __restrict__ joints_r, const float* __restrict__ mesh, int V,
                                                           float* __restrict__ centres, float* __restrict__ radii) {
    __shared__ float s_J[63], s_jr[21], s_Jc[63];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t >= 64 && t < 127) s_Jc[t - 64] = joints_c[b * 63 + t - 64];
    joint_radii(sm, joints_r + b * 63, mesh + (int64_t)b * V * 3, s_J, s_jr, nullptr);       // (ends with a barrier)
    if (t < NS) {
        const float r_root = fminf(fmaxf(s_jr[0] - 0.05f, 0.01f), 0.4f);
        float c[3], cr[3], r, r_unused;
        sphere_from_joints(sm, s_J, s_jr, r_root, t, cr, r);                 // radius: the MANO skeleton's
        sphere_from_joints(sm, s_Jc, s_jr, r_root, t, c, r_unused);          // centre: the other skeleton's
        radii[b * NS + t] = r;
        centres[(b * NS + t) * 3] = c[0]; centres[(b * NS + t) * 3 + 1] = c[1]; centres[(b * NS + t) * 3 + 2] = c[2];
    }
}

__global__ __launch_bounds__(256) void collision_fwd_kernel(dsf_sphere_model sm, const float* __restrict__ joints,
                                                            const float* __restrict__ mesh, int V,
                                                            float* __restrict__ loss_rows, float* __restrict__ centres,
                                                            float* __restrict__ radii, int32_t* __restrict__ topk) {
    __shared__ float s_J[63], s_jr[21], s_c[NS * 3], s_r[NS];
    const int b = blockIdx.x, t = threadIdx.x;
    joint_radii(sm, joints + b * 63, mesh + (int64_t)b * V * 3, s_J, s_jr, topk ? topk + b * 16 * TOPK : nullptr);
    if (t < NS) {
        const float r_root = fminf(fmaxf(s_jr[0] - 0.05f, 0.01f), 0.4f);
        float c[3], r;
        sphere_from_joints(sm, s_J, s_jr, r_root, t, c, r);
        s_r[t] = r; s_c[t * 3] = c[0]; s_c[t * 3 + 1] = c[1]; s_c[t * 3 + 2] = c[2];
        if (radii) radii[b * NS + t] = r;
        if (centres) { centres[(b * NS + t) * 3] = c[0]; centres[(b * NS + t) * 3 + 1] = c[1]; centres[(b * NS + t) * 3 + 2] = c[2]; }
    }
    __syncthreads();
    if (t < NS) {
        float row = 0.f;
        for (int j = 0; j < NS; ++j) {
            const float m = sm.coll_mask[t * NS + j];
            const float dx = s_c[t * 3] - s_c[j * 3], dy = s_c[t * 3 + 1] - s_c[j * 3 + 1], dz = s_c[t * 3 + 2] - s_c[j * 3 + 2];
            const float d = sqrtf(dx * dx + dy * dy + dz * dz + 1e-8f);
            row += fmaxf(s_r[t] + s_r[j] - d, 0.f) * m;
        }
        // :383 sums the last axis twice with keepdim -> the 0.1 gate is per sphere row
        loss_rows[b * NS + t] = (row < 0.1f) ? row : 0.f;
    }
}

template <bool DET>
__global__ __launch_bounds__(256) void collision_bwd_kernel(dsf_sphere_model sm, const float* __restrict__ joints,
                                                            const float* __restrict__ mesh,
                                                            const float* __restrict__ centres,
                                                            const float* __restrict__ radii,
                                                            const int32_t* __restrict__ topk,
                                                            const float* __restrict__ grad_rows, int V,
                                                            float* __restrict__ g_joints, float* __restrict__ g_mesh) {
    typedef Acc<DET> A;                               // float atomics, or order-independent fixed point (deterministic mode)
    typedef typename A::T AT;
    __shared__ float s_c[NS * 3], s_r[NS], s_gate[NS], s_J[63], s_jr0;
    __shared__ AT s_gc[NS * 3], s_gr[NS], s_gJ[63], s_gjr[21];
    __shared__ AT s_gm[DET ? 16 * TOPK * 3 : 1];      // deterministic mode: the (joint, rank) mesh contributions, summed per vertex below
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < NS) {
        s_r[t] = radii[b * NS + t];
        s_c[t * 3] = centres[(b * NS + t) * 3]; s_c[t * 3 + 1] = centres[(b * NS + t) * 3 + 1]; s_c[t * 3 + 2] = centres[(b * NS + t) * 3 + 2];
        s_gr[t] = 0; s_gc[t * 3] = 0; s_gc[t * 3 + 1] = 0; s_gc[t * 3 + 2] = 0;
    }
    if (t >= 64 && t < 127) { s_J[t - 64] = joints[b * 63 + t - 64]; s_gJ[t - 64] = 0; }
    if (t >= 128 && t < 149) s_gjr[t - 128] = 0;
    __syncthreads();
    // row gates (recomputed) ------------------------------------------------------------
    if (t < NS) {
        float row = 0.f;
        for (int j = 0; j < NS; ++j) {
            const float dx = s_c[t * 3] - s_c[j * 3], dy = s_c[t * 3 + 1] - s_c[j * 3 + 1], dz = s_c[t * 3 + 2] - s_c[j * 3 + 2];
            const float d = sqrtf(dx * dx + dy * dy + dz * dz + 1e-8f);
            row += fmaxf(s_r[t] + s_r[j] - d, 0.f) * sm.coll_mask[t * NS + j];
        }
        s_gate[t] = (row < 0.1f) ? grad_rows[b * NS + t] : 0.f;
    }
    __syncthreads();
    for (int q = t; q < NS * NS; q += 256) {
        const int i = q / NS, j = q % NS;
        const float g = s_gate[i] * sm.coll_mask[q];
        if (g == 0.f) continue;
        const float dx = s_c[i * 3] - s_c[j * 3], dy = s_c[i * 3 + 1] - s_c[j * 3 + 1], dz = s_c[i * 3 + 2] - s_c[j * 3 + 2];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz + 1e-8f);
        if (s_r[i] + s_r[j] - d <= 0.f) continue;
        A::add(&s_gr[i], g); A::add(&s_gr[j], g);
        const float k = g / d;
        A::add(&s_gc[i * 3], -k * dx); A::add(&s_gc[i * 3 + 1], -k * dy); A::add(&s_gc[i * 3 + 2], -k * dz);
        A::add(&s_gc[j * 3], k * dx); A::add(&s_gc[j * 3 + 1], k * dy); A::add(&s_gc[j * 3 + 2], k * dz);
    }
    __syncthreads();
    // spheres -> joint radii / joint positions ----------------------------------------------
    if (t < NS) {
        const float gr = A::get(s_gr[t]);
        const float gc[3] = {A::get(s_gc[t * 3]), A::get(s_gc[t * 3 + 1]), A::get(s_gc[t * 3 + 2])};
        if (t == 0) {
            A::add(&s_gjr[0], gr);                          // routed through the clamp below (as r_root)
            for (int a = 0; a < 3; ++a) A::add(&s_gJ[a], gc[a]);
        } else if (t < NPALM) {
            const int k = (t - 1) >> 2, m = (t - 1) & 3, kn = c_knuckle[k];
            const float tt = sm.t_palm[m];
            A::add(&s_gjr[kn], gr * tt);
            A::add(&s_gjr[0], gr * (1.f - tt));
            for (int a = 0; a < 3; ++a) { A::add(&s_gJ[kn * 3 + a], gc[a] * tt); A::add(&s_gJ[a], gc[a] * (1.f - tt)); }
        } else {
            const int i = (t - NPALM) / 3, m = (t - NPALM) % 3, ch = c_child[i], pa = i + 1;
            const float tt = sm.t_finger[m];
            // radii of bones are not routed through r_root: joint_r[1:16] / children
            A::add(&s_gjr[ch], gr * tt);
            A::add(&s_gjr[pa], gr * (1.f - tt));
            for (int a = 0; a < 3; ++a) { A::add(&s_gJ[ch * 3 + a], gc[a] * tt); A::add(&s_gJ[pa * 3 + a], gc[a] * (1.f - tt)); }
        }
    }
    __syncthreads();
    // NOTE: s_gjr[0] currently holds d/d(r_root) (palm spheres only use r_root, never joint_r[0]).
    // recompute joint_r[0] from its top-10 list to evaluate the clamp derivative.
    if (t == 0) {
        float sum = 0.f;
        for (int r = 0; r < TOPK; ++r) {
            const int v = topk[(b * 16 + 0) * TOPK + r];
            if (v < 0) { sum += 100.f; continue; }
            const float dx = s_J[0] - mesh[((int64_t)b * V + v) * 3], dy = s_J[1] - mesh[((int64_t)b * V + v) * 3 + 1],
                        dz = s_J[2] - mesh[((int64_t)b * V + v) * 3 + 2];
            sum += sqrtf(dx * dx + dy * dy + dz * dz + 1e-8f);
        }
        const float x = sum / (float)TOPK - 0.05f;
        s_jr0 = (x >= 0.01f && x <= 0.4f) ? 1.f : 0.f;
    }
    __syncthreads();
    if (t < 5) A::add(&s_gjr[3 + 3 * t], A::get(s_gjr[16 + t]) / 1.5f);     // tip radii = parent / 1.5 (:281)
    __syncthreads();
    // joint radii -> joints / mesh through the 10 selected distances ------------------------------
    if (t < 16 * TOPK) {
        const int j = t / TOPK;
        const int v = topk[(b * 16 + j) * TOPK + (t % TOPK)];
        const float g = A::get(s_gjr[j]) * (j == 0 ? s_jr0 : 1.f) / (float)TOPK;     // (joint 0: through the clamp)
        float c[3] = {0.f, 0.f, 0.f};
        if (v >= 0 && g != 0.f) {
            const float* mv = mesh + ((int64_t)b * V + v) * 3;
            const float dx = s_J[j * 3] - mv[0], dy = s_J[j * 3 + 1] - mv[1], dz = s_J[j * 3 + 2] - mv[2];
            const float d = sqrtf(dx * dx + dy * dy + dz * dz + 1e-8f);
            const float k = g / d;
            A::add(&s_gJ[j * 3], k * dx); A::add(&s_gJ[j * 3 + 1], k * dy); A::add(&s_gJ[j * 3 + 2], k * dz);
            c[0] = -k * dx; c[1] = -k * dy; c[2] = -k * dz;
            if (!DET) {
                float* gm = g_mesh + ((int64_t)b * V + v) * 3;
                atomicAdd(gm, c[0]); atomicAdd(gm + 1, c[1]); atomicAdd(gm + 2, c[2]);
            }
        }
        if (DET) { s_gm[t * 3] = 0; A::add(&s_gm[t * 3], c[0]); s_gm[t * 3 + 1] = 0; A::add(&s_gm[t * 3 + 1], c[1]);
                   s_gm[t * 3 + 2] = 0; A::add(&s_gm[t * 3 + 2], c[2]); }
    }
    __syncthreads();
    if (DET && t < 16 * TOPK) {
        // a vertex can sit in the top-10 lists of several joints: the FIRST (joint, rank) slot that names it sums them all
        const int v = topk[(b * 16 + t / TOPK) * TOPK + (t % TOPK)];
        bool first = v >= 0;
        for (int u = 0; u < t && first; ++u) first = topk[(b * 16 + u / TOPK) * TOPK + (u % TOPK)] != v;
        if (first) {
            AT s0 = 0, s1 = 0, s2 = 0;
            for (int u = t; u < 16 * TOPK; ++u)
                if (topk[(b * 16 + u / TOPK) * TOPK + (u % TOPK)] == v) { s0 += s_gm[u * 3]; s1 += s_gm[u * 3 + 1]; s2 += s_gm[u * 3 + 2]; }
            float* gm = g_mesh + ((int64_t)b * V + v) * 3;
            gm[0] = A::get(s0); gm[1] = A::get(s1); gm[2] = A::get(s2);
        }
    }
    if (t < 63) g_joints[b * 63 + t] = A::get(s_gJ[t]);
}

__global__ __launch_bounds__(256) void seg_pcl_kernel(const float* __restrict__ centres, const float* __restrict__ radii,
                                                      const float* __restrict__ pcl, int P, int wg_per_sample,
                                                      int64_t* __restrict__ labels) {
    __shared__ float s_c[NS * 3], s_r[NS];
    const int b = blockIdx.x / wg_per_sample, part = blockIdx.x % wg_per_sample, t = threadIdx.x;
    if (t < NS) {
        s_r[t] = radii[b * NS + t];
        s_c[t * 3] = centres[(b * NS + t) * 3]; s_c[t * 3 + 1] = centres[(b * NS + t) * 3 + 1]; s_c[t * 3 + 2] = centres[(b * NS + t) * 3 + 2];
    }
    __syncthreads();
    for (int p = part * 256 + t; p < P; p += wg_per_sample * 256) {
        const float* q = pcl + ((int64_t)b * P + p) * 3;
        const float x = q[0], y = q[1], z = q[2];
        float fbest = INFINITY, pbest = INFINITY;
        int fi = 0;
        for (int s = NPALM; s < NS; ++s) {
            const float dx = x - s_c[s * 3], dy = y - s_c[s * 3 + 1], dz = z - s_c[s * 3 + 2];
            const float d = fabsf(sqrtf(dx * dx + dy * dy + dz * dz + 1e-8f) - s_r[s]);
            if (d < fbest) { fbest = d; fi = s - NPALM; }
        }
        for (int s = 0; s < NPALM; ++s) {
            const float dx = x - s_c[s * 3], dy = y - s_c[s * 3 + 1], dz = z - s_c[s * 3 + 2];
            const float d = fabsf(sqrtf(dx * dx + dy * dy + dz * dz + 1e-8f) - s_r[s]);
            if (d < pbest) pbest = d;
        }
        labels[(int64_t)b * P + p] = (pbest < fbest) ? 0 : (int64_t)(fi / 3 + 1);
    }
}

}  // namespace

extern "C" int dsf_sphere_set(const dsf_sphere_model* sm, const float* joints, const float* mesh, int B, int V,
                              float* centres, float* radii, int32_t* topk_idx, dsf_stream_t stream) {
    DSF_CHECK_ARG(sm && joints && mesh && centres && radii && B >= 0 && V >= 778);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(sphere_set_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, *sm, joints, mesh, V, centres,
                       radii, topk_idx);
    return dsf_launch_status();
}

extern "C" int dsf_sphere_mixed(const dsf_sphere_model* sm, const float* joints_centres, const float* joints_radii, const float* mesh, int B,
                                int V, float* centres, float* radii, dsf_stream_t stream) {
    DSF_CHECK_ARG(sm && joints_centres && joints_radii && mesh && centres && radii && B >= 0 && V >= 778);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(sphere_mixed_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, *sm, joints_centres, joints_radii, mesh, V, centres,
                       radii);
    return dsf_launch_status();
}

extern "C" int dsf_collision_forward(const dsf_sphere_model* sm, const float* joints, const float* mesh, int B, int V,
                                     float* loss_rows, float* centres, float* radii, int32_t* topk_idx,
                                     dsf_stream_t stream) {
    DSF_CHECK_ARG(sm && joints && mesh && loss_rows && B >= 0 && V >= 778);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(collision_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, *sm, joints, mesh, V, loss_rows,
                       centres, radii, topk_idx);
    return dsf_launch_status();
}

extern "C" int dsf_collision_backward(const dsf_sphere_model* sm, const float* joints, const float* mesh,
                                      const float* centres, const float* radii, const int32_t* topk_idx,
                                      const float* grad_rows, int B, int V, float* grad_joints, float* grad_mesh,
                                      dsf_stream_t stream) {
    DSF_CHECK_ARG(sm && joints && mesh && centres && radii && topk_idx && grad_rows && grad_joints && grad_mesh);
    DSF_CHECK_ARG(B >= 0 && V >= 778);
    if (dsf_zero_async(grad_mesh, sizeof(float) * 3 * (size_t)B * V, (hipStream_t)stream) != hipSuccess)
        return DSF_ERR_LAUNCH;
    if (B == 0) return DSF_OK;
    if (dsf_deterministic())
        hipLaunchKernelGGL(collision_bwd_kernel<true>, dim3(B), dim3(256), 0, (hipStream_t)stream, *sm, joints, mesh, centres,
                           radii, topk_idx, grad_rows, V, grad_joints, grad_mesh);
    else
        hipLaunchKernelGGL(collision_bwd_kernel<false>, dim3(B), dim3(256), 0, (hipStream_t)stream, *sm, joints, mesh, centres,
                           radii, topk_idx, grad_rows, V, grad_joints, grad_mesh);
    return dsf_launch_status();
}

extern "C" int dsf_seg_pcl(const float* centres, const float* radii, const float* pcl, int B, int P, int64_t* labels,
                           dsf_stream_t stream) {
    DSF_CHECK_ARG(centres && radii && pcl && labels && B >= 0 && P >= 0);
    if (B == 0 || P == 0) return DSF_OK;
    // one point per lane: 66 sphere tests with an IEEE square root each are a ~2600-instruction chain per point; four points per
    // lane (rounds 1-4) left the chip at one wave per SIMD on half of its SIMDs (B = 64, P = 2048: 42 us)
    int g = (P + 255) / 256;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(seg_pcl_kernel, dim3(B * g), dim3(256), 0, (hipStream_t)stream, centres, radii, pcl, P, g, labels);
    return dsf_launch_status();
}
