// K3/K4: point -> triangle squared distance (pytorch3d==0.4.0 semantics, SURVEY.md
// Appendix A.4) for gfx950.
//
// One lane owns one point; triangle records (vertices, unit normal and the Gram-matrix
// invariants, computed once per workgroup instead of once per pair) are staged through LDS
// and read back as wave-uniform broadcasts.  The kernels are VALU-bound (about 120 flops and
// five IEEE divisions per pair), not HBM-bound.
//
// Argmin ties (ubiquitous: a point nearest to a shared edge sees the same distance from
// both triangles) resolve to the lowest triangle index, as in the oracle; for that the
// per-pair arithmetic below is the oracle's, operation for operation (-ffp-contract=off).
//
// Two forward forms: the packed one (pytorch3d._C's argument lists: every point of a mesh
// against every triangle of it, 256-record chunks, exhaustive) and the fused batched one
// behind ICPLoss / JointICPLoss (mesh_point_fwd_kernel), which tests each point only against
// the triangles of its own part -- the point list of a (sample, part) workgroup is compacted
// in LDS first (the reference replicates the cloud 15x and throws 14/15 of the results away,
// metric/meshLoss.py:377-395) -- and, since round 4, only against triangles whose bounding
// sphere can hold a minimiser (the cull described above that kernel).
#include "common.h"
#include <cstdlib>

namespace {

constexpr float kEps = 1e-8f;
constexpr int CHUNK = 256;

struct TriRec {                 // 80 B
    float v0x, v0y, v0z, v1x, v1y, v1z, v2x, v2y, v2z;
    float nx, ny, nz;           // normal / (|normal| + eps)
    float nn;                   // |normal|
    float d00, d01, d11, denom; // Gram invariants of (v1-v0, v2-v0)
    float l12;                  // |v2-v1|^2
    int id;
    float m1;                   // max |v_k|_1 over the three vertices: scales the rounding margin of plane_skip()
};

__device__ __forceinline__ TriRec make_tri(f3 v0, f3 v1, f3 v2, int id) {
    TriRec r;
    r.v0x = v0.x; r.v0y = v0.y; r.v0z = v0.z; r.v1x = v1.x; r.v1y = v1.y; r.v1z = v1.z;
    r.v2x = v2.x; r.v2y = v2.y; r.v2z = v2.z;
    const f3 q0 = v1 - v0, q1 = v2 - v0;
    const f3 n = cross(q1, q0);
    r.nn = sqrtf(dot(n, n));
    const float den = r.nn + kEps;
    r.nx = n.x / den; r.ny = n.y / den; r.nz = n.z / den;
    r.d00 = dot(q0, q0); r.d01 = dot(q0, q1); r.d11 = dot(q1, q1);
    r.denom = r.d00 * r.d11 - r.d01 * r.d01 + kEps;
    const f3 e12 = v2 - v1;
    r.l12 = dot(e12, e12);
    r.id = id;
    r.m1 = fmaxf(fabsf(v0.x) + fabsf(v0.y) + fabsf(v0.z), fmaxf(fabsf(v1.x) + fabsf(v1.y) + fabsf(v1.z), fabsf(v2.x) + fabsf(v2.y) + fabsf(v2.z)));
    return r;
}

__device__ __forceinline__ float seg_dist2(f3 p, f3 a, f3 b, f3 ba, float l2) {
    if (l2 <= kEps) { const f3 d = p - b; return dot(d, d); }
    const float t = dot(ba, p - a) / l2;
    const float tt = fminf(fmaxf(t, 0.0f), 1.0f);
    const f3 d = p - (a + tt * ba);
    return dot(d, d);
}

__device__ __forceinline__ float point_tri_dist2(f3 p, const TriRec& r) {
    const f3 v0 = mk3(r.v0x, r.v0y, r.v0z), v1 = mk3(r.v1x, r.v1y, r.v1z), v2 = mk3(r.v2x, r.v2y, r.v2z);
    const f3 n = mk3(r.nx, r.ny, r.nz);
    const float t = dot(v0 - p, n);
    const f3 p0 = p + t * n;
    const f3 q0 = v1 - v0, q1 = v2 - v0, q2 = p0 - v0;
    const float d20 = dot(q2, q0), d21 = dot(q2, q1);
    // (round 5: skipping these two divisions where the numerators alone decide `inside`, and the edge divisions where the
    //  clamp is known -- exact rewrites, bit-identical in every suite -- measured 6 % SLOWER: 327 / 1741 us against 309 / 1610
    //  for the cloud on the mesh / the collapsed mesh at B = 64.  The lanes of a wave disagree, so nothing is skipped wave-wide.)
    const float w1 = (r.d11 * d20 - r.d01 * d21) / r.denom;
    const float w2 = (r.d00 * d21 - r.d01 * d20) / r.denom;
    const float w0 = 1.0f - w1 - w2;
    const bool inside = (0.0f <= w0 && w0 <= 1.0f) && (0.0f <= w1 && w1 <= 1.0f) && (0.0f <= w2 && w2 <= 1.0f);
    if (inside && r.nn > kEps) return t * t;
    const float e01 = seg_dist2(p, v0, v1, q0, r.d00);
    const float e02 = seg_dist2(p, v0, v2, q1, r.d11);
    const float e12 = seg_dist2(p, v1, v2, v2 - v1, r.l12);
    float d = (e01 > e02) ? e02 : e01;
    d = (d > e12) ? e12 : d;
    return d;
}

// A lower bound of point_tri_dist2(p, r) for the price of the plane term (round 6).  Both branches of the reference are bounded by
// it: the plane branch returns t^2 with exactly this t; the three segment distances are distances to points OF the triangle, hence
// >= the true distance to its plane >= |t| (t is computed with the normal divided by |n| + 1e-8: shorter than the unit normal).
// E covers the fp32 rounding of t and of the segment distances (a few ulps of the coordinates' magnitudes): the pair is skipped only
// when (|t| - E)^2 > best STRICTLY, so a skipped pair's value is larger than the current minimum and ties are still decided among
// all minimisers; NaNs compare false and are never skipped.  What it buys: on a mesh the sphere cull cannot work on -- collapsed to
// a blob, as a freshly initialised network predicts it: every triangle is tiny, its `inside` test (barycentrics over denom + 1e-8)
// holds far away and its reported distance is the PLANE distance, so each sphere is unbounded -- most triangles' planes still pass
// far from a given point, and the points of a wave (Morton-ordered) agree on which do not.
__device__ __forceinline__ bool plane_skip(f3 p, float p1, const TriRec& r, float best) {
    const f3 v0 = mk3(r.v0x, r.v0y, r.v0z), n = mk3(r.nx, r.ny, r.nz);
    const float t = dot(v0 - p, n);
    const float a = fabsf(t) - 4e-6f * (p1 + r.m1);
    return a > 0.0f && a * a > best;
}

__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// A second bound for triangles whose cull sphere is inflated (round 6).  The reference's `inside` test divides by denom + 1e-8, so
// it holds on the triangle scaled about v0 by k = denom / (denom - 1e-8) (tri_sphere): 3-4x for the triangles of a SMALL mesh -- the
// estimate of a freshly initialised network, Gram determinants ~1e-9 -- and the sphere that bounds the scaled triangle then reaches a
// quarter of the mesh for every point (profiles/r06_pfd_pairs.txt).  But the plane branch needs the point's PROJECTION inside that
// scaled triangle.  With d2 = |p - c|^2 to the inflated sphere's centre (a point of the plane) and t the plane term, the projection
// lies rho^2 = d2 - t_true^2 from c, t_true^2 <= 1.0021 t^2 (bounded spheres have |n| / (|n| + 1e-8) >= 1 / 1.001); if rho exceeds
// the inflated radius, `inside` fails and the value is a distance to the TRUE triangle's edges, >= |p - centroid| - circumradius-like
// bound of the unscaled triangle.  Margins as in tri_sphere / plane_skip; strict comparison; NaN / inf never skip.
__device__ __forceinline__ bool tight_skip(f3 p, float p1, const TriRec& r, float4 sph, float d2, float best) {
    // only where the sphere is inflated by more than half (k = denom / (denom - 1e-8) > 1.5 <=> denom < 3e-8): on an ordinary mesh the
    // test is pure cost (+6 .. +17 % per launch, measured)
    if (!(sph.w < INFINITY) || !(r.denom < 3e-8f)) return false;
    const f3 v0 = mk3(r.v0x, r.v0y, r.v0z), v1 = mk3(r.v1x, r.v1y, r.v1z), v2 = mk3(r.v2x, r.v2y, r.v2z), n = mk3(r.nx, r.ny, r.nz);
    const float E = 4e-6f * (p1 + r.m1);
    const float ta = fabsf(dot(v0 - p, n)) + E;
    const float rho2 = d2 * (1.0f - 4.0f * 1e-4f) - 1.0021f * ta * ta;
    const float Rf = sph.w * (1.0f + 1e-4f);
    if (!(rho2 > Rf * Rf)) return false;                // the projection may lie inside the scaled triangle: the plane branch is possible
    const f3 ct = (1.0f / 3.0f) * ((v0 + v1) + v2);
    const f3 e0 = v0 - ct, e1 = v1 - ct, e2 = v2 - ct, dp = p - ct;
    const float Rt = sqrtf(fmaxf(dot(e0, e0), fmaxf(dot(e1, e1), dot(e2, e2))));
    const float a = sqrtf(dot(dp, dp)) * (1.0f - 1e-4f) - Rt * (1.0f + 1e-4f) - E;
    return a > 0.0f && a * a > best;
}

// ---- packed (pytorch3d._C) form ---------------------------------------------------------------
__global__ __launch_bounds__(256) void pfd_packed_fwd_kernel(const float* __restrict__ points,
                                                             const int64_t* __restrict__ pfirst,
                                                             const float* __restrict__ tris,
                                                             const int64_t* __restrict__ tfirst, int N, int64_t P,
                                                             int64_t T, float* __restrict__ dists,
                                                             int64_t* __restrict__ idxs) {
    __shared__ TriRec s_tri[CHUNK];
    const int n = blockIdx.y, t = threadIdx.x;
    const int64_t p0 = pfirst[n], p1 = (n + 1 < N) ? pfirst[n + 1] : P;
    const int64_t t0 = tfirst[n], t1 = (n + 1 < N) ? tfirst[n + 1] : T;
    const int64_t p = p0 + (int64_t)blockIdx.x * 256 + t;
    if (p0 + (int64_t)blockIdx.x * 256 >= p1) return;          // whole workgroup past the cloud
    const bool live = p < p1;
    const f3 pt = live ? ld3(points + p * 3) : mk3(0.f, 0.f, 0.f);
    float best = INFINITY;
    int64_t bi = -1;
    for (int64_t base = t0; base < t1; base += CHUNK) {
        __syncthreads();
        if (base + t < t1) {
            const float* v = tris + (base + t) * 9;
            s_tri[t] = make_tri(ld3(v), ld3(v + 3), ld3(v + 6), t);
        }
        __syncthreads();
        const int cnt = (int)((t1 - base < CHUNK) ? (t1 - base) : CHUNK);
        if (live) {
            for (int q = 0; q < cnt; ++q) {
                const float d = point_tri_dist2(pt, s_tri[q]);
                if (d < best || bi < 0) { best = d; bi = base + q; }
            }
        }
    }
    if (live) { dists[p] = (bi < 0) ? 0.f : best; idxs[p] = bi; }
}

// gradient of the selected branch (Appendix A.4 / oracle p3d_ref.c:orc_point_face_dist_backward)
__device__ __forceinline__ void seg_backward(f3 p, f3 a, f3 b, float g, f3& gp, f3& ga, f3& gb) {
    const f3 ba = b - a;
    const float l2 = dot(ba, ba);
    if (l2 <= kEps) { const f3 d = (2.0f * g) * (p - b); gp = gp + d; gb = gb - d; return; }
    const float t = dot(ba, p - a) / l2;
    if (t < 0.0f) { const f3 d = (2.0f * g) * (p - a); gp = gp + d; ga = ga - d; }
    else if (t > 1.0f) { const f3 d = (2.0f * g) * (p - b); gp = gp + d; gb = gb - d; }
    else {
        const f3 d = (2.0f * g) * (p - (a + t * ba));
        gp = gp + d; ga = ga - (1.0f - t) * d; gb = gb - t * d;
    }
}

__device__ __forceinline__ void point_tri_backward(f3 p, f3 v0, f3 v1, f3 v2, float g, f3& gp, f3& g0, f3& g1, f3& g2) {
    gp = mk3(0.f, 0.f, 0.f); g0 = gp; g1 = gp; g2 = gp;
    const TriRec r = make_tri(v0, v1, v2, 0);
    const f3 n = mk3(r.nx, r.ny, r.nz);
    const float t = dot(v0 - p, n);
    const f3 p0 = p + t * n;
    const f3 q0 = v1 - v0, q1 = v2 - v0, q2 = p0 - v0;
    const float d20 = dot(q2, q0), d21 = dot(q2, q1);
    const float w1 = (r.d11 * d20 - r.d01 * d21) / r.denom;
    const float w2 = (r.d00 * d21 - r.d01 * d20) / r.denom;
    const float w0 = 1.0f - w1 - w2;
    const bool inside = (0.0f <= w0 && w0 <= 1.0f) && (0.0f <= w1 && w1 <= 1.0f) && (0.0f <= w2 && w2 <= 1.0f);
    if (inside && r.nn > kEps) {
        const float gt = 2.0f * g * t;
        gp = (-gt) * n;
        g0 = gt * n;
        const f3 gn = gt * (v0 - p);
        const f3 raw = cross(q1, q0);
        const float den = r.nn + kEps;
        const float s = dot(gn, raw) / (den * den * r.nn);
        const f3 graw = (1.0f / den) * gn - s * raw;
        const f3 ge2 = cross(q0, graw), ge1 = cross(graw, q1);     // raw = q1 x q0
        g2 = g2 + ge2; g1 = g1 + ge1; g0 = (g0 - ge2) - ge1;
    } else {
        const float e01 = seg_dist2(p, v0, v1, q0, r.d00);
        const float e02 = seg_dist2(p, v0, v2, q1, r.d11);
        const float e12 = seg_dist2(p, v1, v2, v2 - v1, r.l12);
        if (e01 <= e02 && e01 <= e12) seg_backward(p, v0, v1, g, gp, g0, g1);
        else if (e02 <= e01 && e02 <= e12) seg_backward(p, v0, v2, g, gp, g0, g2);
        else seg_backward(p, v1, v2, g, gp, g1, g2);
    }
}

__global__ void pfd_packed_bwd_kernel(const float* __restrict__ points, const float* __restrict__ tris,
                                      const int64_t* __restrict__ idxs, const float* __restrict__ gd, int64_t P,
                                      float* __restrict__ gpoints, float* __restrict__ gtris) {
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int64_t ti = idxs[p];
    if (ti < 0) return;
    f3 gp, g0, g1, g2;
    const float* v = tris + ti * 9;
    point_tri_backward(ld3(points + p * 3), ld3(v), ld3(v + 3), ld3(v + 6), gd[p], gp, g0, g1, g2);
    gpoints[p * 3] = gp.x; gpoints[p * 3 + 1] = gp.y; gpoints[p * 3 + 2] = gp.z;
    float* o = gtris + ti * 9;
    atomicAdd(o, g0.x); atomicAdd(o + 1, g0.y); atomicAdd(o + 2, g0.z);
    atomicAdd(o + 3, g1.x); atomicAdd(o + 4, g1.y); atomicAdd(o + 5, g1.z);
    atomicAdd(o + 6, g2.x); atomicAdd(o + 7, g2.y); atomicAdd(o + 8, g2.z);
}

// ---- fused batched (sample, part) form --------------------------------------------------------
// Round 4: a conservative cull in front of the per-pair arithmetic (which stays the oracle's).
//
// Every triangle gets a bounding sphere (centre c, radius R) of the region whose points the reference can report at
// distance t^2 (plane distance): its `inside` test divides by denom + 1e-8, so the barycentrics it sees are the true
// ones times s = denom_true / (denom_true + 1e-8), i.e. `inside` holds on the triangle SCALED about v0 by k = 1 / s
// (>= 1; 1.01 for cube-normalised hands); the plane distance it reports is the true one times |n| = nn / (nn + 1e-8).
// The three segment distances of the other branch are distances to points of the triangle itself.  Hence, for every
// point p,  computed_dist2(p, tri) >= (max(0, |p - c| - R) * nn / (nn + 1e-8))^2  up to float rounding, and a lane whose
// current minimum is `best` may skip a triangle when  |p - c| * (1 - m) > R * (1 + m) + sqrt(best) * (1 + m) * 1.0011
// (m = 1e-4 covers the rounding of every quantity involved; triangles with nn / (nn + 1e-8) < 1 / 1.001 or an unbounded k
// get R = inf and are never skipped).  The inequality is STRICT, so a skipped triangle's distance is larger than the
// minimum: ties (ubiquitous on shared edges) are still decided among all minimisers, by the lowest index -- the update
// rule compares indices, so the order in which triangles are visited no longer matters.  That allows (1) a seed: each
// lane first finds the sphere centre nearest to its point (9 operations per pair) and evaluates that triangle, which puts
// `best` within a small factor of the minimum before the scan starts; (2) points taken in a spatially coherent order (a
// counting sort of the sample's cloud by 9-bit Morton cell, rebuilt by every workgroup of the sample in LDS; a workgroup
// takes the WHOLE cells whose first slot falls into its 256-slot window, so the partition does not depend on the order the
// LDS atomics give inside a cell), so that the 64 lanes of a wave skip the same triangles -- a triangle is evaluated when
// ANY live lane needs it.
constexpr int LIST_CAP = 2560;      // points per workgroup range (uint16 list entries; 5 KB: the kernel stays under 32 KB of LDS = 5 workgroups per CU)
constexpr int CH = 256;             // triangle records + spheres staged per pass: 64 per wave, each wave its own (7 passes over MANO's 1554 faces)
constexpr int GP = 64;              // points per group: one per lane, the same 64 in each of the four waves
constexpr int SEG_RANGE = LIST_CAP;  // labelled clouds: points one (sample, part) workgroup compacts its part's members from (512-point
                                     // ranges = 4 x the workgroups, each staging the part's triangles again: 223 us against 171)
constexpr float CULL_M = 1e-4f;
#ifdef PFD_STATS     // diagnostic build only (tools/pfd_pairs.py): pairs a lane needed / lane slots the wave spent on evaluations
__device__ unsigned long long pfd_stat[4];
#define PFD_COUNT(i, v) do { if (lane == (int)__builtin_ctzll(__ballot(true))) atomicAdd(&pfd_stat[i], (unsigned long long)(v)); } while (0)
#else
#define PFD_COUNT(i, v) do { } while (0)
#endif

__device__ __forceinline__ float4 tri_sphere(const TriRec& r) {
    const f3 v0 = mk3(r.v0x, r.v0y, r.v0z), q0 = mk3(r.v1x, r.v1y, r.v1z) - v0, q1 = mk3(r.v2x, r.v2y, r.v2z) - v0;
    float k = 1.0f;
    bool bounded = true;
    if (r.nn > kEps) {                                  // the plane branch exists for this triangle
        const float dt = r.denom - kEps;                // ~ denom_true
        bounded = dt > 0.0f && (r.nn + kEps) <= 1.001f * r.nn;
        k = bounded ? r.denom / dt : 1.0f;
        bounded = bounded && k < 64.0f;
    }
    const f3 a = k * q0, b = k * q1;                    // scaled triangle: v0, v0 + a, v0 + b
    const f3 cc = (1.0f / 3.0f) * (a + b);              // centroid - v0
    const f3 d1 = a - cc, d2 = b - cc;
    const float r2 = fmaxf(dot(cc, cc), fmaxf(dot(d1, d1), dot(d2, d2)));
    const f3 c = v0 + cc;
    const float R = bounded ? sqrtf(r2) * ((1.0f + 2.0f * CULL_M) / (1.0f - CULL_M)) : INFINITY;
    return make_float4(c.x, c.y, c.z, R);
}

__global__ __launch_bounds__(256) void mesh_point_fwd_kernel(const float* __restrict__ verts,
                                                             const float* __restrict__ points,
                                                             const int32_t* __restrict__ faces,
                                                             const int32_t* __restrict__ part_first,
                                                             const int64_t* __restrict__ seg, int V, int P, int n_parts,
                                                             int splits, float* __restrict__ dists,
                                                             int32_t* __restrict__ idxs) {
    // A workgroup = 64 points x 4 waves: every wave holds the SAME 64 points (one per lane) and scans its quarter of each staged
    // block of triangles; the four minima meet in LDS at the end.  The pair loop is a chain of dependent divisions and LDS
    // broadcasts, so what it needs is waves: 64-point groups give B x P / 64 workgroups (2048 at B = 64: 256-point groups left
    // the chip at one or two waves per SIMD and a launch took 600 us whatever B was).  Every wave stages ITS OWN 64 triangles of a
    // block (one record per lane) and reads only those, so the scan needs no workgroup barrier at all -- a first version staged
    // 192 triangles cooperatively behind 45 barriers per group and spent most of its time waiting at them.
    // The triangle stage and the scratch of the point sort share their bytes: the sort is over before the first stage is built.
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[CH * (sizeof(TriRec) + sizeof(float4))];
    TriRec* const s_tri = reinterpret_cast<TriRec*>(s_raw);
    float4* const s_sph = reinterpret_cast<float4*>(s_raw + CH * sizeof(TriRec));
    int* const s_hist = reinterpret_cast<int*>(s_raw);                  // 512 cell counters
    uint8_t* const s_owner = s_raw + 2048;                              // 512 cell owners
    float* const s_box = reinterpret_cast<float*>(s_raw + 2560);        // 4 waves x (min, max)
    __shared__ uint16_t s_list[LIST_CAP];
    __shared__ float s_rd[4 * 64];                                      // per wave, per lane: (distance, index) for the cross-wave minima
    __shared__ int s_ri[4 * 64];
    __shared__ int s_n, s_hi, s_wsum[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // (Round 5 measured an EXHAUSTIVE role for meshes the cull cannot work on -- 256 points per workgroup, every pair, chosen per
    //  sample inside the launch from 64 sampled triangles and points -- as the round-4 brief asked: 1610-1850 us against 1862 for the
    //  adaptive loop below on the mesh collapsed to 1 % at B = 64, and SLOWER inside the steps that are in that state (config 3's
    //  first steps: 532-840 us per launch against 495; whole steps of configs 3 and 5 with and without it: no difference): removed
    //  again.  What those steps really paid for was the labelled form.)
    int w = blockIdx.x;
    int split;
    if (seg) {
        // labelled clouds: the split index varies SLOWEST.  Workgroups are dealt round-robin over the 8 XCDs, and the groups of a
        // part are dealt to splits 0, 1, ...: with `split = id % 8` every non-empty workgroup of a launch whose parts have one or
        // two groups each landed on XCDs 0 and 1 (a quarter of the chip: 361 us against 118 for one workgroup per part)
        const int per_split = (int)gridDim.x / splits;
        split = w / per_split; w %= per_split;
    } else { split = w % splits; w /= splits; }
    const int part = w % n_parts;
    const int b = w / n_parts;
    const float* vb = verts + (int64_t)b * V * 3;
    const float* pb = points + (int64_t)b * P * 3;
    const int f0 = part_first[part], f1 = part_first[part + 1];

    int pbeg, n_mine;                  // this workgroup's points: s_list[0 .. n_mine) + pbeg, or the plain range when !listed
    bool listed = true;
    int g_first = 0, g_step = 1;       // the groups of 64 listed points this workgroup walks: g_first, g_first + g_step, ...
    if (seg && P <= LIST_CAP) {
        // Labelled cloud (JointICPLoss): EVERY workgroup of a (sample, part) compacts the part's members of the whole cloud, in
        // index order (each wave a contiguous quarter: count, exchange, write -- no LDS atomics, so all `splits` workgroups hold
        // the same list), and takes every splits-th group of 64 of them.  Round 4 cut the cloud into index ranges instead, one
        // workgroup each: with the labels of a real depth crop (most points on the palm and two or three fingers; uniform
        // labels only in the synthetic timing scenario) one workgroup then owned ~20 groups and the launch was its latency --
        // 1.4 ms per launch in the first steps of config 5, 0.38 ms in its fitted state, against 0.2 / 0.14 ms with the groups
        // dealt out.  Which workgroup evaluates a point does not change its result.
        const int quarter = ((P + 255) / 256) * 64;                     // points per wave, a multiple of 64
        const int w0 = wave * quarter, w1 = min(P, w0 + quarter);
        constexpr int LAB_IT = LIST_CAP / 256;                          // <= 10 labels per lane, all loads in flight at once
        bool mine[LAB_IT];
        int cnt = 0;
#pragma unroll
        for (int it = 0; it < LAB_IT; ++it) {
            const int p = w0 + it * 64 + lane;
            const bool in = it * 64 < quarter && p < w1;
            const int64_t lab = in ? seg[(int64_t)b * P + p] : 0;
            if (in && part == 0 && split == 0 && (lab < 1 || lab > n_parts)) {          // no part: the reference's masked-out zeros
                dists[(int64_t)b * P + p] = 0.f;
                idxs[(int64_t)b * P + p] = -1;
            }
            mine[it] = in && lab == part + 1;
        }
#pragma unroll
        for (int it = 0; it < LAB_IT; ++it) cnt += __popcll(__ballot(mine[it]));
        if (lane == 0) s_wsum[wave] = cnt;
        __syncthreads();
        int off = 0;
        for (int q = 0; q < wave; ++q) off += s_wsum[q];
        n_mine = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
#pragma unroll
        for (int it = 0; it < LAB_IT; ++it) {
            const unsigned long long mask = __ballot(mine[it]);
            if (mine[it]) s_list[off + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)(w0 + it * 64 + lane);
            off += __popcll(mask);
        }
        __syncthreads();
        pbeg = 0;
        g_first = split; g_step = splits;
    } else if (seg) {
        const int per = (P + splits - 1) / splits;
        pbeg = split * per;
        const int pend = min(P, pbeg + per);
        if (t == 0) s_n = 0;
        __syncthreads();
        for (int p = pbeg + t; p < pend; p += 256) {
            const int64_t lab = seg[(int64_t)b * P + p];
            if (part == 0 && (lab < 1 || lab > n_parts)) {              // no part: the reference's masked-out zeros
                dists[(int64_t)b * P + p] = 0.f;
                idxs[(int64_t)b * P + p] = -1;
            }
            if (lab == part + 1) s_list[atomicAdd(&s_n, 1)] = (uint16_t)(p - pbeg);
        }
        __syncthreads();
        n_mine = s_n;
    } else if (P <= LIST_CAP) {
        // the whole cloud of the sample in Morton-cell order (every workgroup of the sample builds the same cells; the order
        // inside a cell is whatever the LDS atomics give -- results do not depend on it), this workgroup takes 256 of them
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int p = t; p < P; p += 256) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { const float x = pb[p * 3 + c]; lo[c] = fminf(lo[c], x); hi[c] = fmaxf(hi[c], x); }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o, 64)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o, 64)); }
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { s_box[wave * 6 + c] = lo[c]; s_box[wave * 6 + 3 + c] = hi[c]; }
        }
        for (int i = t; i < 512; i += 256) s_hist[i] = 0;
        __syncthreads();
        float sc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = fminf(fminf(s_box[c], s_box[6 + c]), fminf(s_box[12 + c], s_box[18 + c]));
            hi[c] = fmaxf(fmaxf(s_box[3 + c], s_box[9 + c]), fmaxf(s_box[15 + c], s_box[21 + c]));
            const float ext = hi[c] - lo[c];
            sc[c] = (ext > 0.f && ext < INFINITY) ? 8.0f / ext : 0.f;
        }
        auto cell = [&](int p) {
            int k = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float u = (pb[p * 3 + c] - lo[c]) * sc[c];
                const int q = (u >= 0.f) ? min(7, (int)u) : 0;            // (NaN / inf coordinates land in cell 0)
                k |= ((q & 1) << c) | ((q & 2) << (c + 2)) | ((q & 4) << (c + 4));
            }
            return k;
        };
        for (int p = t; p < P; p += 256) atomicAdd(&s_hist[cell(p)], 1);
        if (t == 0) { s_n = 0x7fffffff; s_hi = 0; }
        __syncthreads();
        {                                                               // exclusive scan of the 512 cell counts
            const int a0 = s_hist[2 * t], a1 = s_hist[2 * t + 1];
            int v = a0 + a1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o, 64); if (lane >= o) v += u; }
            if (lane == 63) s_wsum[wave] = v;
            __syncthreads();
            int base = 0;
            for (int q = 0; q < wave; ++q) base += s_wsum[q];
            const int excl = base + v - (a0 + a1);
            s_hist[2 * t] = excl; s_hist[2 * t + 1] = excl + a0;
            // a cell belongs to the workgroup whose 64-slot window holds the cell's FIRST slot: whole cells, so that the
            // partition of the cloud over the sample's workgroups does not depend on the order inside a cell
            const int o0 = excl >> 6, o1 = (excl + a0) >> 6;
            s_owner[2 * t] = (uint8_t)min(o0, 255); s_owner[2 * t + 1] = (uint8_t)min(o1, 255);
            if (a0 > 0 && o0 == split) { atomicMin(&s_n, excl); atomicMax(&s_hi, excl + a0); }
            if (a1 > 0 && o1 == split) { atomicMin(&s_n, excl + a0); atomicMax(&s_hi, excl + a0 + a1); }
        }
        __syncthreads();
        const int first = s_n;
        n_mine = max(0, s_hi - first);                                  // <= 63 + the largest cell <= P <= LIST_CAP
        for (int p = t; p < P; p += 256) {
            const int k = cell(p);
            if (s_owner[k] == split) s_list[atomicAdd(&s_hist[k], 1) - first] = (uint16_t)p;
        }
        __syncthreads();
        pbeg = 0;
    } else {
        // clouds too large for the LDS list: plain index ranges (the cull still applies, lane-coherence is what it is)
        listed = false;
        pbeg = 0; n_mine = 0;
    }
    // the groups of 64 points this workgroup walks: its list (part-compacted, or Morton-cell sorted), else slots split * 64 + lane
    const int n_groups = listed ? (n_mine + GP - 1) / GP : 1;

    TriRec* const w_tri = s_tri + wave * 64;            // this wave's stage: 64 records + spheres, written and read by this wave only
    float4* const w_sph = s_sph + wave * 64;
    // wave w takes the triangle blocks w, w + 4, w + 8, ... of 64: its stage for block k holds triangles f0 + k * 64 ...
    constexpr int bs = 64;                               // (32-triangle blocks for the ~100-triangle hand parts: no faster)
    auto stage = [&](int k, bool full) -> int {
        const int first = f0 + k * bs;
        const int cnt = min(bs, f1 - first);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // (this wave's earlier reads of the stage are done)
        __builtin_amdgcn_wave_barrier();
        if (lane < cnt) {
            const int32_t* fc = faces + (first + lane) * 3;
            const f3 v0 = ld3(vb + fc[0] * 3), v1 = ld3(vb + fc[1] * 3), v2 = ld3(vb + fc[2] * 3);
            if (full) {
                const TriRec r = make_tri(v0, v1, v2, lane);
                w_tri[lane] = r;
                w_sph[lane] = tri_sphere(r);
            } else {                                     // the seed only ranks triangles: plain centroids, no square roots / divisions
                const f3 c = (1.0f / 3.0f) * ((v0 + v1) + v2);
                w_sph[lane] = make_float4(c.x, c.y, c.z, 0.f);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // LDS operations of one wave execute in order: the stage is
        __builtin_amdgcn_wave_barrier();                                // written before any lane of this wave reads it
        return cnt;
    };
    const int n_blocks = (f1 - f0 + bs - 1) / bs;
    __syncthreads();                                     // the sort scratch (which shares the stage's bytes) is dead
    for (int g = g_first; g < n_groups; g += g_step) {
        bool live;
        int p;
        if (listed) { live = g * GP + lane < n_mine; p = live ? pbeg + s_list[g * GP + lane] : 0; }
        else { p = split * GP + lane; live = p < P; p = live ? p : 0; }
        const f3 pt = live ? ld3(pb + p * 3) : mk3(0.f, 0.f, 0.f);
        const float pt1 = fabsf(pt.x) + fabsf(pt.y) + fabsf(pt.z);         // (plane_skip's rounding margin)
        float best = INFINITY, thr = INFINITY;          // thr = sqrt(best) * (1 + m) * 1.0011 / (1 - m), refreshed with best
        int bi = -1;
        // ---- seed: the nearest centroid among every 2nd triangle of this wave's blocks; the four waves' candidates meet in LDS
        //      and every wave evaluates the winner exactly ----
        if (f1 > f0) {
            float dmin = INFINITY;
            int smin = f0;
            for (int k = wave; k < n_blocks; k += 4) {
                const int cnt = stage(k, false);
                for (int q = 0; q < cnt; q += 2) {
                    const float4 s = w_sph[q];
                    const float dx = pt.x - s.x, dy = pt.y - s.y, dz = pt.z - s.z;
                    const float d2 = dx * dx + dy * dy + dz * dz;
                    if (d2 < dmin) { dmin = d2; smin = f0 + k * bs + q; }
                }
            }
            __syncthreads();                             // (the previous group's final exchange is over)
            s_rd[t] = dmin; s_ri[t] = smin;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float d = s_rd[k * 64 + lane]; if (d < dmin) { dmin = d; smin = s_ri[k * 64 + lane]; } }
            if (live) {
                const int32_t* fc = faces + smin * 3;
                const TriRec r = make_tri(ld3(vb + fc[0] * 3), ld3(vb + fc[1] * 3), ld3(vb + fc[2] * 3), 0);
                best = point_tri_dist2(pt, r);
                bi = smin;
                thr = sqrtf(best) * ((1.0f + CULL_M) * 1.0011f / (1.0f - CULL_M));
            }
        }
        // ---- the group's points as ONE ball (centre of their box, radius to the farthest) and its loosest threshold: a triangle
        //      with |c_tri - c_ball| - r_ball > R_tri + thr_max is skipped by every lane's own test, so one lane per TRIANGLE can
        //      discard 64 of them per instruction; NaN-ignoring min / max keep a lane with broken coordinates from widening it ----
        float bx0 = live ? pt.x : INFINITY, by0 = live ? pt.y : INFINITY, bz0 = live ? pt.z : INFINITY;
        float bx1 = live ? pt.x : -INFINITY, by1 = live ? pt.y : -INFINITY, bz1 = live ? pt.z : -INFINITY;
        float tmax = live ? thr : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            bx0 = fminf(bx0, __shfl_xor(bx0, o, 64)); by0 = fminf(by0, __shfl_xor(by0, o, 64)); bz0 = fminf(bz0, __shfl_xor(bz0, o, 64));
            bx1 = fmaxf(bx1, __shfl_xor(bx1, o, 64)); by1 = fmaxf(by1, __shfl_xor(by1, o, 64)); bz1 = fmaxf(bz1, __shfl_xor(bz1, o, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64));
        }
        const f3 cw = mk3(0.5f * (bx0 + bx1), 0.5f * (by0 + by1), 0.5f * (bz0 + bz1));
        float rw = 0.f;
        if (live) { const f3 d = pt - cw; rw = sqrtf(dot(d, d)); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rw = fmaxf(rw, __shfl_xor(rw, o, 64));
        const float ball_reach = (rw + tmax) * (1.0f + CULL_M);       // (inf / NaN when some lane has no finite threshold: nothing is discarded)
        // ---- scan: this wave's blocks, no workgroup barrier ----
        const int n_live = __popcll(__ballot(live));
        bool dense = false;
        int since = 0;
        for (int k = wave; k < n_blocks; k += 4) {
            const int cnt = stage(k, true);
            bool keep = false;
            if (lane < cnt) {
                const float4 s = w_sph[lane];
                const float dx = cw.x - s.x, dy = cw.y - s.y, dz = cw.z - s.z;
                const float dc = sqrtf(dx * dx + dy * dy + dz * dz) * (1.0f - CULL_M);
                keep = !(dc > s.w + ball_reach);
            }
            unsigned long long cand = __ballot(keep);    // the triangles of this block the group's ball can reach, lowest first
            if (dense && ++since < 4) {
                // the last tested block was evaluated almost whole (a mesh beside the cloud, or collapsed to a blob as a freshly
                // initialised network predicts it: every triangle is then a near-minimiser of every point): the per-triangle tests
                // only cost -- the plain loop over the stage, whose LDS reads the compiler runs ahead of the arithmetic (the
                // candidate loop cannot: 2.0 ms per launch in config 5's first steps against 0.9 for the brute-force kernel of round
                // 3).  Evaluating a triangle the cull would have skipped never changes the result: it cannot hold a minimiser.
                // Every 4th block is tested again.
                if (live) {
                    const float was = best;
                    for (int q = 0; q < cnt; ++q) {
                        const bool sk = bi >= 0 && plane_skip(pt, pt1, w_tri[q], best);
                        const unsigned long long nsk = __ballot(!sk);
                        if (!nsk) continue;                                                // (the live lanes agree: its plane passes far from all of them)
                        PFD_COUNT(0, __popcll(nsk)); PFD_COUNT(1, __popcll(__ballot(true))); PFD_COUNT(2, 1);
                        if (sk) continue;
                        const float d = point_tri_dist2(pt, w_tri[q]);
                        const int id = f0 + k * bs + q;
                        if (bi < 0 || d < best || (d == best && id < bi)) { best = d; bi = id; }
                    }
                    if (best != was) thr = sqrtf(best) * ((1.0f + CULL_M) * 1.0011f / (1.0f - CULL_M));
                }
                continue;
            }
            since = 0;
            const int n_cand = __popcll(cand);
            int evals = 0;
            while (cand) {
                const int q = __builtin_ctzll(cand);
                cand &= cand - 1;
                const float4 s = w_sph[q];
                const float dx = pt.x - s.x, dy = pt.y - s.y, dz = pt.z - s.z;
                const float lim = s.w + thr;
                bool need = live && !(dx * dx + dy * dy + dz * dz > lim * lim);            // (NaN distances are never skipped)
                if (need && bi >= 0 && (plane_skip(pt, pt1, w_tri[q], best) || tight_skip(pt, pt1, w_tri[q], s, dx * dx + dy * dy + dz * dz, best))) need = false;
                const unsigned long long who = __ballot(need);
                if (!who) continue;                                                        // nobody needs this triangle
                PFD_COUNT(0, __popcll(who)); PFD_COUNT(1, n_live); PFD_COUNT(3, 1);
                evals += __popcll(who);
                if (need) {
                    const float d = point_tri_dist2(pt, w_tri[q]);
                    const int id = f0 + k * bs + q;
                    if (bi < 0 || d < best || (d == best && id < bi)) {
                        best = d; bi = id;
                        thr = sqrtf(best) * ((1.0f + CULL_M) * 1.0011f / (1.0f - CULL_M));
                    }
                }
            }
            dense = n_cand * 2 > cnt && evals * 2 > n_cand * n_live;
        }
        // ---- the four waves' minima: smallest distance, then smallest index (the oracle's tie rule) ----
        __syncthreads();
        s_rd[t] = best; s_ri[t] = bi;
        __syncthreads();
        if (wave == 0 && live) {
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const float d = s_rd[k * 64 + lane];
                const int id = s_ri[k * 64 + lane];
                if (id >= 0 && (bi < 0 || d < best || (d == best && id < bi))) { best = d; bi = id; }
            }
            dists[(int64_t)b * P + p] = (bi < 0) ? 0.f : best;
            idxs[(int64_t)b * P + p] = bi;
        }
    }
}

constexpr int BWD_MAX_V = 1024;

template <bool DET>
__global__ __launch_bounds__(256) void mesh_point_bwd_kernel(const float* __restrict__ verts,
                                                             const float* __restrict__ points,
                                                             const int32_t* __restrict__ faces,
                                                             const int32_t* __restrict__ idxs,
                                                             const float* __restrict__ gd, int V, int P, int per_wg,
                                                             float* __restrict__ gverts, float* __restrict__ gpoints) {
    typedef Acc<DET> A;                               // float atomics, or order-independent fixed point (deterministic mode)
    __shared__ typename A::T s_g[BWD_MAX_V * 3];
    const int t = threadIdx.x;
    const int wgs = (P + per_wg - 1) / per_wg;
    const int b = blockIdx.x / wgs, chunk = blockIdx.x % wgs;
    for (int e = t; e < V * 3; e += 256) s_g[e] = 0;
    __syncthreads();
    const float* vb = verts + (int64_t)b * V * 3;
    const int pend = min(P, (chunk + 1) * per_wg);
    for (int p = chunk * per_wg + t; p < pend; p += 256) {
        const int64_t o = (int64_t)b * P + p;
        const int fi = idxs[o];
        f3 gp = mk3(0.f, 0.f, 0.f);
        const float g = gd[o];
        if (fi >= 0 && g != 0.f) {
            const int32_t* fc = faces + fi * 3;
            f3 g0, g1, g2;
            point_tri_backward(ld3(points + o * 3), ld3(vb + fc[0] * 3), ld3(vb + fc[1] * 3), ld3(vb + fc[2] * 3), g, gp,
                               g0, g1, g2);
            A::add(&s_g[fc[0] * 3], g0.x); A::add(&s_g[fc[0] * 3 + 1], g0.y); A::add(&s_g[fc[0] * 3 + 2], g0.z);
            A::add(&s_g[fc[1] * 3], g1.x); A::add(&s_g[fc[1] * 3 + 1], g1.y); A::add(&s_g[fc[1] * 3 + 2], g1.z);
            A::add(&s_g[fc[2] * 3], g2.x); A::add(&s_g[fc[2] * 3 + 1], g2.y); A::add(&s_g[fc[2] * 3 + 2], g2.z);
        }
        if (gpoints) { gpoints[o * 3] = gp.x; gpoints[o * 3 + 1] = gp.y; gpoints[o * 3 + 2] = gp.z; }
    }
    __syncthreads();
    for (int e = t; e < V * 3; e += 256) {
        const float v = A::get(s_g[e]);
        if (v != 0.f) atomicAdd(gverts + (int64_t)b * V * 3 + e, v);
    }
}

}  // namespace

extern "C" int dsf_point_face_dist_forward(const float* points, const int64_t* points_first_idx, const float* tris,
                                           const int64_t* tris_first_idx, int N, int64_t P, int64_t T,
                                           int64_t max_points, float* dists, int64_t* idxs, dsf_stream_t stream) {
    DSF_CHECK_ARG(points_first_idx && tris_first_idx && dists && idxs && N >= 0 && P >= 0 && T >= 0);
    if (N == 0 || P == 0) return DSF_OK;
    DSF_CHECK_ARG(points && (tris || T == 0) && max_points > 0);
    hipLaunchKernelGGL(pfd_packed_fwd_kernel, dim3((unsigned)((max_points + 255) / 256), N), dim3(256), 0,
                       (hipStream_t)stream, points, points_first_idx, tris, tris_first_idx, N, P, T, dists, idxs);
    return dsf_launch_status();
}

extern "C" int dsf_point_face_dist_backward(const float* points, const float* tris, const int64_t* idxs,
                                            const float* grad_dists, int64_t P, int64_t T, float* grad_points,
                                            float* grad_tris, dsf_stream_t stream) {
    DSF_CHECK_ARG(idxs && grad_dists && grad_points && grad_tris && P >= 0 && T >= 0);
    if (dsf_zero_async(grad_points, sizeof(float) * 3 * P, (hipStream_t)stream) != hipSuccess ||
        dsf_zero_async(grad_tris, sizeof(float) * 9 * T, (hipStream_t)stream) != hipSuccess)
        return DSF_ERR_LAUNCH;
    if (P == 0) return DSF_OK;
    hipLaunchKernelGGL(pfd_packed_bwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       points, tris, idxs, grad_dists, P, grad_points, grad_tris);
    return dsf_launch_status();
}

extern "C" int dsf_mesh_point_dist_forward(const float* verts, const float* points, const int32_t* faces,
                                           const int32_t* part_first, const int64_t* seg, int B, int V, int P,
                                           int n_parts, float* dists, int32_t* idxs, dsf_stream_t stream) {
    DSF_CHECK_ARG(verts && points && faces && part_first && dists && idxs);
    DSF_CHECK_ARG(B >= 0 && V > 0 && P >= 0 && n_parts >= 1 && (seg || n_parts == 1));
    if (B == 0 || P == 0) return DSF_OK;
    // seg == NULL: every point meets every triangle -> 64 points per workgroup (4 waves share them and split the triangles);
    // with labels one workgroup per (sample, part) compacts its part's members of the cloud (<= LIST_CAP points per range)
    // and walks them in groups of 64.
    // labelled clouds that fit the LDS list: 8 workgroups per (sample, part) deal out the part's groups of 64 members between them
    // (one per 256 points of the cloud); larger clouds: one workgroup per SEG_RANGE points
    int splits = seg ? (P <= LIST_CAP ? (P + 255) / 256 : (P + SEG_RANGE - 1) / SEG_RANGE) : (P + GP - 1) / GP;
    if (const char* e = getenv("DSF_PFD_SPLITS")) { const int v = atoi(e); if (seg && P <= LIST_CAP && v >= 1 && v <= 64) splits = v; }
    if (splits < 1) splits = 1;
    hipLaunchKernelGGL(mesh_point_fwd_kernel, dim3((unsigned)(B * n_parts * splits)), dim3(256), 0, (hipStream_t)stream,
                       verts, points, faces, part_first, seg, V, P, n_parts, splits, dists, idxs);
    return dsf_launch_status();
}

extern "C" int dsf_mesh_point_dist_backward(const float* verts, const float* points, const int32_t* faces,
                                            const int32_t* idxs, const float* grad_dists, int B, int V, int P,
                                            float* grad_verts, float* grad_points, dsf_stream_t stream) {
    DSF_CHECK_ARG(verts && points && faces && idxs && grad_dists && grad_verts && B >= 0 && V > 0 && V <= BWD_MAX_V);
    if (dsf_zero_async(grad_verts, sizeof(float) * 3 * (size_t)B * V, (hipStream_t)stream) != hipSuccess)
        return DSF_ERR_LAUNCH;
    if (B == 0 || P == 0) return DSF_OK;
    if (dsf_deterministic()) {                    // one workgroup per sample: its fixed-point table holds the whole sum
        hipLaunchKernelGGL(mesh_point_bwd_kernel<true>, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, verts, points,
                           faces, idxs, grad_dists, V, P, P, grad_verts, grad_points);
        return dsf_launch_status();
    }
    const int per_wg = 1024;
    const int wgs = (P + per_wg - 1) / per_wg;
    hipLaunchKernelGGL(mesh_point_bwd_kernel<false>, dim3((unsigned)(B * wgs)), dim3(256), 0, (hipStream_t)stream, verts, points,
                       faces, idxs, grad_dists, V, P, per_wg, grad_verts, grad_points);
    return dsf_launch_status();
}

#ifdef PFD_STATS
extern "C" int dsf_pfd_stats(unsigned long long* out) {
    unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pfd_stat), sizeof(z)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(pfd_stat), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif
