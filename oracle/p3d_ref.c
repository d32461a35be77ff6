/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by
 * the product path (dsf_amd/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * Plain-C scalar restatement of the pytorch3d==0.4.0 ops the reference calls
 * (the wheel is NOT vendored under /root/reference and is not installable
 * here, so this follows the published v0.4.0 algorithm as restated in
 * SURVEY.md Appendix A; "parity unpinned" against the real wheel):
 *
 *   rasterize_meshes fwd/bwd   call sites  render_model/mano_layer.py:1021-1023,
 *                              1053-1055, 1082-1084 (settings :946-951)
 *   point_face_dist  fwd/bwd   call sites  metric/meshLoss.py:52, 63
 *   PerspectiveCameras screen->NDC transform   mano_layer.py:935-945
 *
 * Every float op is a separate IEEE-754 binary32 operation (compile with
 * -ffp-contract=off): the HIP kernels must reproduce these bits for the index
 * outputs to be identical.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define K_EPS 1e-8f

/* ---------------- A.1 camera: world -> (x_ndc, y_ndc, z_view) ---------------- */
/* R = diag(-1,-1,1), T = 0 (mano_layer.py:935-938); screen-space intrinsics with
 * image_size=(W,H): fx' = fx/(W/2), px' = -(px - W/2)/(W/2); z is overwritten
 * with the view-space z by MeshRasterizer.transform. */
void orc_project_verts(const float* verts, int64_t n, float fx, float fy, float px, float py,
                       float W, float H, float* out) {
    const float hw = W / 2.0f, hh = H / 2.0f;
    const float fxn = fx / hw, fyn = fy / hh;
    const float pxn = -(px - hw) / hw, pyn = -(py - hh) / hh;
    for (int64_t i = 0; i < n; ++i) {
        const float xv = -verts[3 * i + 0], yv = -verts[3 * i + 1], zv = verts[3 * i + 2];
        const float ox = xv * fxn + zv * pxn;
        const float oy = yv * fyn + zv * pyn;
        out[3 * i + 0] = ox / zv;
        out[3 * i + 1] = oy / zv;
        out[3 * i + 2] = zv;
    }
}

/* ---------------- A.2 rasterize_meshes forward (naive path) ---------------- */
static inline float pix_to_ndc(int i, int S) { return -1.0f + (2 * i + 1.0f) / (float)S; }

static inline float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

static inline float fmin3(float a, float b, float c) { return fminf(fminf(a, b), c); }
static inline float fmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

static float point_line_dist2(float px, float py, float ax, float ay, float bx, float by) {
    const float bax = bx - ax, bay = by - ay;
    const float l2 = bax * bax + bay * bay;
    float t = (bax * (px - ax) + bay * (py - ay)) / l2;
    if (l2 <= K_EPS) return (px - bx) * (px - bx) + (py - by) * (py - by);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float qx = ax + t * bax, qy = ay + t * bay;
    const float dx = px - qx, dy = py - qy;
    return dx * dx + dy * dy;
}

/* face_verts: (F_total,3,3) of (x_ndc, y_ndc, z_view). Outputs are (N,S,S[,3]).
 * blur_radius = 0, faces_per_pixel = 1, perspective_correct = clip = cull = false.
 * bary/dists may be NULL. */
void orc_rasterize_meshes(const float* face_verts, const int64_t* mesh_first, const int64_t* mesh_nfaces,
                          int N, int S, int64_t* pix_to_face, float* zbuf, float* bary, float* dists) {
    const int64_t npix = (int64_t)N * S * S;
    for (int64_t i = 0; i < npix; ++i) {
        pix_to_face[i] = -1;
        zbuf[i] = -1.0f;
        if (dists) dists[i] = -1.0f;
        if (bary) bary[3 * i] = bary[3 * i + 1] = bary[3 * i + 2] = -1.0f;
    }
    for (int n = 0; n < N; ++n) {
        for (int64_t f = mesh_first[n]; f < mesh_first[n] + mesh_nfaces[n]; ++f) {
            const float* v = face_verts + 9 * f;
            const float x0 = v[0], y0 = v[1], z0 = v[2], x1 = v[3], y1 = v[4], z1 = v[5], x2 = v[6], y2 = v[7], z2 = v[8];
            if (fmax3(z0, z1, z2) < 0.0f) continue;
            const float face_area = edge_fn(x0, y0, x1, y1, x2, y2);
            if (face_area <= K_EPS && face_area >= -K_EPS) continue;
            const float xmin = fmin3(x0, x1, x2), xmax = fmax3(x0, x1, x2);
            const float ymin = fmin3(y0, y1, y2), ymax = fmax3(y0, y1, y2);
            const float area = edge_fn(x2, y2, x0, y0, x1, y1) + K_EPS;
            for (int yi = 0; yi < S; ++yi) {
                const float yf = pix_to_ndc(yi, S);
                if (yf > ymax || yf < ymin) continue;
                for (int xi = 0; xi < S; ++xi) {
                    const float xf = pix_to_ndc(xi, S);
                    if (xf > xmax || xf < xmin) continue;
                    const float w0 = edge_fn(xf, yf, x1, y1, x2, y2) / area;
                    const float w1 = edge_fn(xf, yf, x2, y2, x0, y0) / area;
                    const float w2 = edge_fn(xf, yf, x0, y0, x1, y1) / area;
                    const float pz = w0 * z0 + w1 * z1 + w2 * z2;
                    if (pz < 0.0f) continue;
                    if (!(w0 > 0.0f && w1 > 0.0f && w2 > 0.0f)) continue;     /* blur 0: outside => rejected */
                    /* image axes are reversed: +Y up, +X left */
                    const int64_t o = ((int64_t)n * S + (S - 1 - yi)) * S + (S - 1 - xi);
                    if (pix_to_face[o] < 0 || pz < zbuf[o]) {                /* strict <: lowest face wins ties */
                        pix_to_face[o] = f;
                        zbuf[o] = pz;
                        if (bary) { bary[3 * o] = w0; bary[3 * o + 1] = w1; bary[3 * o + 2] = w2; }
                        if (dists) {
                            const float d = fmin3(point_line_dist2(xf, yf, x0, y0, x1, y1),
                                                  point_line_dist2(xf, yf, x0, y0, x2, y2),
                                                  point_line_dist2(xf, yf, x1, y1, x2, y2));
                            dists[o] = -d;
                        }
                    }
                }
            }
        }
    }
}

/* ---------------- A.3 rasterize_meshes backward (grad_zbuf only) ---------------- */
/* grad_face_verts (F_total,3,3) must be zero-initialised by the caller. */
void orc_rasterize_backward_zbuf(const float* face_verts, const int64_t* pix_to_face, const float* grad_zbuf,
                                 int N, int S, float* grad_face_verts) {
    for (int n = 0; n < N; ++n)
        for (int yo = 0; yo < S; ++yo)
            for (int xo = 0; xo < S; ++xo) {
                const int64_t o = ((int64_t)n * S + yo) * S + xo;
                const int64_t f = pix_to_face[o];
                if (f < 0) continue;
                const float g = grad_zbuf[o];
                const float xf = pix_to_ndc(S - 1 - xo, S), yf = pix_to_ndc(S - 1 - yo, S);
                const float* v = face_verts + 9 * f;
                const float x0 = v[0], y0 = v[1], z0 = v[2], x1 = v[3], y1 = v[4], z1 = v[5], x2 = v[6], y2 = v[7], z2 = v[8];
                const float area = edge_fn(x2, y2, x0, y0, x1, y1) + K_EPS;
                const float e0 = edge_fn(xf, yf, x1, y1, x2, y2);
                const float e1 = edge_fn(xf, yf, x2, y2, x0, y0);
                const float e2 = edge_fn(xf, yf, x0, y0, x1, y1);
                const float gw[3] = {g * z0, g * z1, g * z2};          /* dL/dw_i */
                const float e[3] = {e0, e1, e2};
                /* w_i = e_i / area.  d e_i terms and d area terms (area = Edge(v2; v0, v1)). */
                float gx[3] = {0, 0, 0}, gy[3] = {0, 0, 0};
                /* vertex order of e_i = Edge(p; a_i, b_i): (1,2), (2,0), (0,1) */
                const int ia[3] = {1, 2, 0}, ib[3] = {2, 0, 1};
                const float X[3] = {x0, x1, x2}, Y[3] = {y0, y1, y2};
                float garea = 0.0f;
                for (int i = 0; i < 3; ++i) {
                    const float ge = gw[i] / area;
                    garea += gw[i] * (-e[i] / (area * area));
                    const int a = ia[i], b = ib[i];
                    /* Edge(p;a,b) = (px-ax)(by-ay) - (py-ay)(bx-ax) */
                    gx[a] += ge * (yf - Y[b]);
                    gy[a] += ge * (X[b] - xf);
                    gx[b] += ge * (Y[a] - yf);
                    gy[b] += ge * (xf - X[a]);
                }
                /* area = Edge(p=v2; a=v0, b=v1) */
                gx[2] += garea * (y1 - y0);
                gy[2] += garea * (x0 - x1);
                gx[0] += garea * (y2 - y1);
                gy[0] += garea * (x1 - x2);
                gx[1] += garea * (y0 - y2);
                gy[1] += garea * (x2 - x0);
                const float w[3] = {e0 / area, e1 / area, e2 / area};
                float* gf = grad_face_verts + 9 * f;
                for (int i = 0; i < 3; ++i) {
                    gf[3 * i + 0] += gx[i];
                    gf[3 * i + 1] += gy[i];
                    gf[3 * i + 2] += g * w[i];
                }
            }
}

/* ---------------- A.4 point_face_dist ---------------- */
typedef struct { float x, y, z; } v3;
static inline v3 sub3(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline v3 add3(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline v3 mul3(float s, v3 a) { v3 r = {s * a.x, s * a.y, s * a.z}; return r; }
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b) {
    v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}

static float point_seg_dist2_3d(v3 p, v3 a, v3 b) {
    const v3 ba = sub3(b, a);
    const float l2 = dot3(ba, ba);
    if (l2 <= K_EPS) { const v3 d = sub3(p, b); return dot3(d, d); }
    const float t = dot3(ba, sub3(p, a)) / l2;
    const float tt = fminf(fmaxf(t, 0.0f), 1.0f);
    const v3 d = sub3(p, add3(a, mul3(tt, ba)));
    return dot3(d, d);
}

static float point_tri_dist2(v3 p, v3 v0, v3 v1, v3 v2) {
    v3 nrm = cross3(sub3(v2, v0), sub3(v1, v0));
    const float nn = sqrtf(dot3(nrm, nrm));
    const float inv_den = nn + K_EPS;
    nrm.x = nrm.x / inv_den; nrm.y = nrm.y / inv_den; nrm.z = nrm.z / inv_den;
    const float t = dot3(sub3(v0, p), nrm);
    const v3 p0 = add3(p, mul3(t, nrm));
    /* barycentrics of p0 (Gram-matrix form) */
    const v3 q0 = sub3(v1, v0), q1 = sub3(v2, v0), q2 = sub3(p0, v0);
    const float d00 = dot3(q0, q0), d01 = dot3(q0, q1), d11 = dot3(q1, q1);
    const float d20 = dot3(q2, q0), d21 = dot3(q2, q1);
    const float denom = d00 * d11 - d01 * d01 + K_EPS;
    const float w1 = (d11 * d20 - d01 * d21) / denom;
    const float w2 = (d00 * d21 - d01 * d20) / denom;
    const float w0 = 1.0f - w1 - w2;
    const int inside = (0.0f <= w0 && w0 <= 1.0f) && (0.0f <= w1 && w1 <= 1.0f) && (0.0f <= w2 && w2 <= 1.0f);
    if (inside && nn > K_EPS) return t * t;
    const float e01 = point_seg_dist2_3d(p, v0, v1);
    const float e02 = point_seg_dist2_3d(p, v0, v2);
    const float e12 = point_seg_dist2_3d(p, v1, v2);
    float d = (e01 > e02) ? e02 : e01;
    d = (d > e12) ? e12 : d;
    return d;
}

static inline v3 ld3(const float* p) { v3 r = {p[0], p[1], p[2]}; return r; }

/* points (P,3), tris (T,3,3); CSR first-idx arrays of length N; batch element n
 * owns points [pf[n], pf[n+1]) (last one up to P) and tris likewise.
 * ties -> lowest packed triangle index. */
void orc_point_face_dist_forward(const float* points, const int64_t* points_first, const float* tris,
                                 const int64_t* tris_first, int N, int64_t P, int64_t T,
                                 float* dists, int64_t* idxs) {
    for (int n = 0; n < N; ++n) {
        const int64_t p0 = points_first[n], p1 = (n + 1 < N) ? points_first[n + 1] : P;
        const int64_t t0 = tris_first[n], t1 = (n + 1 < N) ? tris_first[n + 1] : T;
        for (int64_t p = p0; p < p1; ++p) {
            const v3 pt = ld3(points + 3 * p);
            float best = INFINITY;
            int64_t bi = -1;
            for (int64_t t = t0; t < t1; ++t) {
                const float d = point_tri_dist2(pt, ld3(tris + 9 * t), ld3(tris + 9 * t + 3), ld3(tris + 9 * t + 6));
                if (d < best || bi < 0) { best = d; bi = t; }
            }
            dists[p] = (bi < 0) ? 0.0f : best;
            idxs[p] = bi;
        }
    }
}

static void seg_backward(v3 p, v3 a, v3 b, float g, v3* gp, v3* ga, v3* gb) {
    const v3 ba = sub3(b, a);
    const float l2 = dot3(ba, ba);
    if (l2 <= K_EPS) {               /* dist = |p-b|^2 */
        const v3 d = mul3(2.0f * g, sub3(p, b));
        *gp = add3(*gp, d); *gb = sub3(*gb, d);
        return;
    }
    const float t = dot3(ba, sub3(p, a)) / l2;
    if (t < 0.0f) {
        const v3 d = mul3(2.0f * g, sub3(p, a));
        *gp = add3(*gp, d); *ga = sub3(*ga, d);
    } else if (t > 1.0f) {
        const v3 d = mul3(2.0f * g, sub3(p, b));
        *gp = add3(*gp, d); *gb = sub3(*gb, d);
    } else {                          /* interior: envelope theorem, diff is orthogonal to ba */
        const v3 diff = sub3(p, add3(a, mul3(t, ba)));
        const v3 d = mul3(2.0f * g, diff);
        *gp = add3(*gp, d);
        *ga = sub3(*ga, mul3(1.0f - t, d));
        *gb = sub3(*gb, mul3(t, d));
    }
}

/* grad_points (P,3) and grad_tris (T,3,3) must be zero-initialised. */
void orc_point_face_dist_backward(const float* points, const float* tris, const int64_t* idxs,
                                  const float* grad_dists, int64_t P, float* grad_points, float* grad_tris) {
    for (int64_t p = 0; p < P; ++p) {
        const int64_t ti = idxs[p];
        if (ti < 0) continue;
        const float g = grad_dists[p];
        const v3 pt = ld3(points + 3 * p);
        const v3 v0 = ld3(tris + 9 * ti), v1 = ld3(tris + 9 * ti + 3), v2 = ld3(tris + 9 * ti + 6);
        v3 gp = {0, 0, 0}, g0 = {0, 0, 0}, g1 = {0, 0, 0}, g2 = {0, 0, 0};
        const v3 e2 = sub3(v2, v0), e1 = sub3(v1, v0);
        v3 raw = cross3(e2, e1);
        const float nn = sqrtf(dot3(raw, raw));
        const float den = nn + K_EPS;
        v3 nrm = {raw.x / den, raw.y / den, raw.z / den};
        const float t = dot3(sub3(v0, pt), nrm);
        const v3 p0 = add3(pt, mul3(t, nrm));
        const v3 q2 = sub3(p0, v0);
        const float d00 = dot3(e1, e1), d01 = dot3(e1, e2), d11 = dot3(e2, e2);
        const float d20 = dot3(q2, e1), d21 = dot3(q2, e2);
        const float denom = d00 * d11 - d01 * d01 + K_EPS;
        const float w1 = (d11 * d20 - d01 * d21) / denom;
        const float w2 = (d00 * d21 - d01 * d20) / denom;
        const float w0 = 1.0f - w1 - w2;
        const int inside = (0.0f <= w0 && w0 <= 1.0f) && (0.0f <= w1 && w1 <= 1.0f) && (0.0f <= w2 && w2 <= 1.0f);
        if (inside && nn > K_EPS) {
            /* dist = t^2, t = (v0-p).n, n = raw/(|raw|+eps) */
            const float gt = 2.0f * g * t;
            gp = mul3(-gt, nrm);
            g0 = mul3(gt, nrm);
            const v3 gn = mul3(gt, sub3(v0, pt));                 /* dL/dn */
            /* n = raw/den, den = |raw|+eps: dL/draw = gn/den - raw (gn.raw)/(den^2 |raw|) */
            const float s = dot3(gn, raw) / (den * den * nn);
            const v3 graw = sub3(mul3(1.0f / den, gn), mul3(s, raw));
            /* raw = e2 x e1: dL/de2 = e1 x graw, dL/de1 = graw x e2 */
            const v3 ge2 = cross3(e1, graw), ge1 = cross3(graw, e2);
            g2 = add3(g2, ge2); g1 = add3(g1, ge1);
            g0 = sub3(sub3(g0, ge2), ge1);
        } else {
            const float e01 = point_seg_dist2_3d(pt, v0, v1);
            const float e02 = point_seg_dist2_3d(pt, v0, v2);
            const float e12 = point_seg_dist2_3d(pt, v1, v2);
            if (e01 <= e02 && e01 <= e12) seg_backward(pt, v0, v1, g, &gp, &g0, &g1);
            else if (e02 <= e01 && e02 <= e12) seg_backward(pt, v0, v2, g, &gp, &g0, &g2);
            else seg_backward(pt, v1, v2, g, &gp, &g1, &g2);
        }
        grad_points[3 * p] += gp.x; grad_points[3 * p + 1] += gp.y; grad_points[3 * p + 2] += gp.z;
        float* gt_ = grad_tris + 9 * ti;
        gt_[0] += g0.x; gt_[1] += g0.y; gt_[2] += g0.z;
        gt_[3] += g1.x; gt_[4] += g1.y; gt_[5] += g1.z;
        gt_[6] += g2.x; gt_[7] += g2.y; gt_[8] += g2.z;
    }
}
