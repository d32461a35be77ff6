// K1/K2/K8: mesh rasteriser for gfx950 with pytorch3d==0.4.0 semantics
// (SURVEY.md Appendix A), in two forms:
//
//  * full mode  -- the pytorch3d._C.rasterize_meshes contract: (N,S,S) fragments.
//    64x16-pixel tiles, one 256-thread workgroup per tile, 4 pixels per lane so that
//    every output row segment is written with 16-byte stores; the faces of the tile's
//    mesh are binned into an LDS list (bbox vs tile test, LDS-atomic append), then every
//    lane z-tests its pixels against the staged records.  Tiles outside the mesh's
//    screen bbox only stream the background value.  Workgroup ids are remapped so the
//    tiles of one mesh share an XCD (its L2 holds that mesh's face records).
//
//  * crop mode  -- the fused leaf of Render.render: only the <=128x128 raster pixels
//    that survive resize(640->480 rows) + nearest crop warp are evaluated.  One wave per
//    8x8 crop tile; the projected vertices, packed face indices and per-face raster-pixel
//    bboxes of the sample are staged once per workgroup in LDS.  A wave ballots the
//    bbox-vs-tile test over 64 faces at a time and compacts the candidates; then one LANE
//    per candidate face walks only the tile pixels inside its bbox and merges
//    (z bits << 32 | face) keys with 64-bit LDS atomic-min, which is exactly the naive
//    path's "strict < on z, lowest face index wins exact ties".
//
// Every float expression that decides coverage / index selection is written as separate
// IEEE binary32 operations in the oracle's order (file is compiled with -ffp-contract=off;
// hipcc's fp32 division and sqrt are correctly rounded by default).
#include "common.h"
#include <cstdlib>

namespace {

constexpr float kEps = 1e-8f;

__device__ __forceinline__ float pix_to_ndc(int i, int S) { return -1.0f + (2 * i + 1.0f) / (float)S; }

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}
__device__ __forceinline__ float min3(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

__device__ __forceinline__ float point_line_dist2(float px, float py, float ax, float ay, float bx, float by) {
    const float bax = bx - ax, bay = by - ay;
    const float l2 = bax * bax + bay * bay;
    float t = (bax * (px - ax) + bay * (py - ay)) / l2;
    if (l2 <= kEps) return (px - bx) * (px - bx) + (py - by) * (py - by);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float qx = ax + t * bax, qy = ay + t * bay;
    const float dx = px - qx, dy = py - qy;
    return dx * dx + dy * dy;
}

struct CamN { float fxn, fyn, pxn, pyn; };
__device__ __forceinline__ CamN cam_ndc(const dsf_camera& c) {
    const float hw = c.img_w / 2.0f, hh = c.img_h / 2.0f;
    CamN r;
    r.fxn = c.fx / hw; r.fyn = c.fy / hh;
    r.pxn = -(c.px - hw) / hw; r.pyn = -(c.py - hh) / hh;
    return r;
}
// world -> (x_ndc, y_ndc, z_view); camera R = diag(-1,-1,1), T = 0 (mano_layer.py:935-938)
__device__ __forceinline__ void project(const CamN& k, float X, float Y, float Z, float& xn, float& yn, float& zv) {
    const float xv = -X, yv = -Y;
    zv = Z;
    const float ox = xv * k.fxn + zv * k.pxn;
    const float oy = yv * k.fyn + zv * k.pyn;
    xn = ox / zv;
    yn = oy / zv;
}

__global__ void project_face_verts_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces,
                                          dsf_camera cam, int N, int V, int F, float* __restrict__ out) {
    const CamN k = cam_ndc(cam);
    const int64_t total = (int64_t)N * F * 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t nf = i / 3;
        const int c = (int)(i % 3);
        const int n = (int)(nf / F), f = (int)(nf % F);
        const int v = faces[f * 3 + c];
        const float* p = verts + ((int64_t)n * V + v) * 3;
        float xn, yn, zv;
        project(k, p[0], p[1], p[2], xn, yn, zv);
        out[i * 3] = xn; out[i * 3 + 1] = yn; out[i * 3 + 2] = zv;
    }
}

// ----------------------------------------------------------------------------------------------
// full mode
// ----------------------------------------------------------------------------------------------
constexpr int TILE_W = 64, TILE_H = 16, BIN_CAP = 512;

struct FaceRec {          // 64 B
    float x0, y0, z0, x1, y1, z1, x2, y2, z2;
    float area;           // Edge(v2; v0, v1) + eps
    float xmin, xmax, ymin, ymax;
    int id;               // packed face index
    int pad;
};

__global__ void mesh_bbox_kernel(const float* __restrict__ fv, const int64_t* __restrict__ first,
                                 const int64_t* __restrict__ count, float* __restrict__ bbox) {
    __shared__ float red[4][4];
    const int n = blockIdx.x, t = threadIdx.x;
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    const int64_t f0 = first[n], nf = count[n];
    for (int64_t i = t; i < nf * 3; i += blockDim.x) {
        const float x = fv[(f0 * 3 + i) * 3], y = fv[(f0 * 3 + i) * 3 + 1];
        xmin = fminf(xmin, x); xmax = fmaxf(xmax, x); ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
    }
    for (int o = 32; o > 0; o >>= 1) {
        xmin = fminf(xmin, __shfl_xor(xmin, o, 64)); xmax = fmaxf(xmax, __shfl_xor(xmax, o, 64));
        ymin = fminf(ymin, __shfl_xor(ymin, o, 64)); ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
    }
    if ((t & 63) == 0) { red[t >> 6][0] = xmin; red[t >> 6][1] = xmax; red[t >> 6][2] = ymin; red[t >> 6][3] = ymax; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            xmin = fminf(xmin, red[w][0]); xmax = fmaxf(xmax, red[w][1]);
            ymin = fminf(ymin, red[w][2]); ymax = fmaxf(ymax, red[w][3]);
        }
        bbox[n * 4] = xmin; bbox[n * 4 + 1] = xmax; bbox[n * 4 + 2] = ymin; bbox[n * 4 + 3] = ymax;
    }
}

__global__ __launch_bounds__(256) void raster_full_kernel(const float* __restrict__ fv,
                                                          const int64_t* __restrict__ first,
                                                          const int64_t* __restrict__ count,
                                                          const float* __restrict__ mesh_bbox, int N, int S,
                                                          int tiles_x, int tiles_y, int64_t* __restrict__ p2f,
                                                          float* __restrict__ zbuf, float* __restrict__ bary,
                                                          float* __restrict__ dists) {
    __shared__ FaceRec s_rec[BIN_CAP];
    __shared__ int s_cnt;
    const int t = threadIdx.x;
    // XCD-aware remap: blocks L, L+8, ... share an XCD -> give them the same mesh.
    const int tiles = tiles_x * tiles_y;
    const int L = blockIdx.x;
    const int n = (L & 7) + 8 * (L / (8 * tiles));
    const int tile = (L >> 3) % tiles;
    if (n >= N) return;
    const int ty = tile / tiles_x, tx = tile % tiles_x;
    const int yo = ty * TILE_H + (t >> 4);
    const int xo0 = tx * TILE_W + (t & 15) * 4;

    // tile bounds in NDC (pixel centres); output pixel (yo,xo) samples NDC of index S-1-o
    const int xo_lo = tx * TILE_W, xo_hi = min(xo_lo + TILE_W, S) - 1;
    const int yo_lo = ty * TILE_H, yo_hi = min(yo_lo + TILE_H, S) - 1;
    const float txmin = pix_to_ndc(S - 1 - xo_hi, S), txmax = pix_to_ndc(S - 1 - xo_lo, S);
    const float tymin = pix_to_ndc(S - 1 - yo_hi, S), tymax = pix_to_ndc(S - 1 - yo_lo, S);

    float bz[4];
    int bf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { bz[k] = INFINITY; bf[k] = -1; }
    const float yf = pix_to_ndc(S - 1 - yo, S);
    float xf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xf[k] = pix_to_ndc(S - 1 - (xo0 + k), S);

    bool tile_live = true;
    if (mesh_bbox) {
        const float* bb = mesh_bbox + n * 4;
        tile_live = !(bb[0] > txmax || bb[1] < txmin || bb[2] > tymax || bb[3] < tymin);
    }
    const int64_t f0 = first[n];
    const int nf = tile_live ? (int)count[n] : 0;
    for (int base = 0; base < nf; base += BIN_CAP) {
        if (t == 0) s_cnt = 0;
        __syncthreads();
        for (int q = t; q < BIN_CAP && base + q < nf; q += 256) {
            const int64_t f = f0 + base + q;
            const float* v = fv + f * 9;
            FaceRec r;
            r.x0 = v[0]; r.y0 = v[1]; r.z0 = v[2]; r.x1 = v[3]; r.y1 = v[4]; r.z1 = v[5]; r.x2 = v[6]; r.y2 = v[7]; r.z2 = v[8];
            const float zmax = max3(r.z0, r.z1, r.z2);
            const float face_area = edge_fn(r.x0, r.y0, r.x1, r.y1, r.x2, r.y2);
            const bool degenerate = (face_area <= kEps && face_area >= -kEps);
            r.xmin = min3(r.x0, r.x1, r.x2); r.xmax = max3(r.x0, r.x1, r.x2);
            r.ymin = min3(r.y0, r.y1, r.y2); r.ymax = max3(r.y0, r.y1, r.y2);
            const bool miss = r.xmin > txmax || r.xmax < txmin || r.ymin > tymax || r.ymax < tymin;
            if (!(zmax < 0.0f) && !degenerate && !miss) {
                r.area = edge_fn(r.x2, r.y2, r.x0, r.y0, r.x1, r.y1) + kEps;
                r.id = (int)f; r.pad = 0;
                s_rec[atomicAdd(&s_cnt, 1)] = r;
            }
        }
        __syncthreads();
        const int cnt = s_cnt;
        for (int q = 0; q < cnt; ++q) {
            const FaceRec r = s_rec[q];
            if (yf > r.ymax || yf < r.ymin) continue;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float x = xf[k];
                if (x > r.xmax || x < r.xmin) continue;
                const float w0 = edge_fn(x, yf, r.x1, r.y1, r.x2, r.y2) / r.area;
                const float w1 = edge_fn(x, yf, r.x2, r.y2, r.x0, r.y0) / r.area;
                const float w2 = edge_fn(x, yf, r.x0, r.y0, r.x1, r.y1) / r.area;
                const float pz = w0 * r.z0 + w1 * r.z1 + w2 * r.z2;
                if (pz < 0.0f) continue;
                if (!(w0 > 0.0f && w1 > 0.0f && w2 > 0.0f)) continue;
                if (bf[k] < 0 || pz < bz[k] || (pz == bz[k] && r.id < bf[k])) { bz[k] = pz; bf[k] = r.id; }
            }
        }
        __syncthreads();
    }

    if (yo >= S) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xo = xo0 + k;
        if (xo >= S) continue;
        const int64_t o = ((int64_t)n * S + yo) * S + xo;
        const int f = bf[k];
        p2f[o] = (f >= 0) ? (int64_t)f : (int64_t)-1;
        zbuf[o] = (f >= 0) ? bz[k] : -1.0f;
        if (bary || dists) {
            float w0 = -1.f, w1 = -1.f, w2 = -1.f, d = -1.f;
            if (f >= 0) {
                const float* v = fv + (int64_t)f * 9;
                const float x0 = v[0], y0 = v[1], x1 = v[3], y1 = v[4], x2 = v[6], y2 = v[7];
                const float area = edge_fn(x2, y2, x0, y0, x1, y1) + kEps;
                w0 = edge_fn(xf[k], yf, x1, y1, x2, y2) / area;
                w1 = edge_fn(xf[k], yf, x2, y2, x0, y0) / area;
                w2 = edge_fn(xf[k], yf, x0, y0, x1, y1) / area;
                d = -min3(point_line_dist2(xf[k], yf, x0, y0, x1, y1), point_line_dist2(xf[k], yf, x0, y0, x2, y2),
                          point_line_dist2(xf[k], yf, x1, y1, x2, y2));
            }
            if (bary) { bary[o * 3] = w0; bary[o * 3 + 1] = w1; bary[o * 3 + 2] = w2; }
            if (dists) dists[o] = d;
        }
    }
}

// d(zbuf)/d(face_verts) for one covered pixel (Appendix A.3): returns grads in g[9]
// (x,y,z per vertex) given upstream g_z and the pixel's NDC position.
__device__ __forceinline__ void zbuf_pixel_backward(const float* v, float xf, float yf, float gz, float* g) {
    const float X[3] = {v[0], v[3], v[6]}, Y[3] = {v[1], v[4], v[7]}, Z[3] = {v[2], v[5], v[8]};
    const float area = edge_fn(X[2], Y[2], X[0], Y[0], X[1], Y[1]) + kEps;
    const float e[3] = {edge_fn(xf, yf, X[1], Y[1], X[2], Y[2]), edge_fn(xf, yf, X[2], Y[2], X[0], Y[0]),
                        edge_fn(xf, yf, X[0], Y[0], X[1], Y[1])};
    float gx[3] = {0.f, 0.f, 0.f}, gy[3] = {0.f, 0.f, 0.f};
    float garea = 0.f;
    const int ia[3] = {1, 2, 0}, ib[3] = {2, 0, 1};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float gw = gz * Z[i];
        const float ge = gw / area;
        garea += gw * (-e[i] / (area * area));
        const int a = ia[i], b = ib[i];
        gx[a] += ge * (yf - Y[b]); gy[a] += ge * (X[b] - xf);
        gx[b] += ge * (Y[a] - yf); gy[b] += ge * (xf - X[a]);
    }
    gx[2] += garea * (Y[1] - Y[0]); gy[2] += garea * (X[0] - X[1]);
    gx[0] += garea * (Y[2] - Y[1]); gy[0] += garea * (X[1] - X[2]);
    gx[1] += garea * (Y[0] - Y[2]); gy[1] += garea * (X[2] - X[0]);
#pragma unroll
    for (int i = 0; i < 3; ++i) { g[i * 3] = gx[i]; g[i * 3 + 1] = gy[i]; g[i * 3 + 2] = gz * (e[i] / area); }
}

__global__ void raster_full_bwd_kernel(const float* __restrict__ fv, const int64_t* __restrict__ p2f,
                                       const float* __restrict__ gzbuf, int64_t npix, int S,
                                       float* __restrict__ gfv) {
    for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < npix; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = p2f[o];
        if (f < 0) continue;
        const float gz = gzbuf[o];
        if (gz == 0.f) continue;
        const int xo = (int)(o % S), yo = (int)((o / S) % S);
        float g[9];
        zbuf_pixel_backward(fv + f * 9, pix_to_ndc(S - 1 - xo, S), pix_to_ndc(S - 1 - yo, S), gz, g);
#pragma unroll
        for (int k = 0; k < 9; ++k) atomicAdd(gfv + f * 9 + k, g[k]);
    }
}

// ----------------------------------------------------------------------------------------------
// crop set-up: center2d, M (comToBounds + Offset2Trans), closed-form inverse
// ----------------------------------------------------------------------------------------------
__global__ void crop_setup_kernel(const float* __restrict__ c3, const float* __restrict__ cube, dsf_camera cam, int B,
                                  int crop, float* __restrict__ c2, float* __restrict__ M,
                                  int32_t* __restrict__ bounds, float* __restrict__ minv) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float X = c3[b * 3], Y = c3[b * 3 + 1], Z = c3[b * 3 + 2];
    const float u = X * cam.fx / (Z + 1e-8f) + cam.px;         // points3DToImg: eps on u only
    const float v = Y * cam.fy / Z + cam.py;
    c2[b * 3] = u; c2[b * 3 + 1] = v; c2[b * 3 + 2] = Z;
    const float sx = cube[b * 3], sy = cube[b * 3 + 1];
    const int xs = (int)floorf((u * Z / cam.fx - sx / 2.0f) / Z * cam.fx + 0.5f);
    const int xe = (int)floorf((u * Z / cam.fx + sx / 2.0f) / Z * cam.fx + 0.5f);
    const int ys = (int)floorf((v * Z / cam.fy - sy / 2.0f) / Z * cam.fy + 0.5f);
    const int ye = (int)floorf((v * Z / cam.fy + sy / 2.0f) / Z * cam.fy + 0.5f);
    if (bounds) { bounds[b * 4] = xs; bounds[b * 4 + 1] = xe; bounds[b * 4 + 2] = ys; bounds[b * 4 + 3] = ye; }
    const int wb = xe - xs, hb = ye - ys;
    const bool wide = wb > hb;
    const int sz0 = wide ? crop : (int)((float)(wb * crop) / (float)hb);
    const int sz1 = wide ? (int)((float)(hb * crop) / (float)wb) : crop;
    const float s = wide ? (float)crop / (float)wb : (float)crop / (float)hb;
    const float ox = (float)(int)floorf((float)crop / 2.0f - (float)sz0 / 2.0f);
    const float oy = (float)(int)floorf((float)crop / 2.0f - (float)sz1 / 2.0f);
    float* m = M + b * 9;
    const float m02 = s * (float)(-xs) + ox, m12 = s * (float)(-ys) + oy;
    m[0] = s; m[1] = 0.f; m[2] = m02; m[3] = 0.f; m[4] = s; m[5] = m12; m[6] = 0.f; m[7] = 0.f; m[8] = 1.f;
    if (minv) {
        const float r = 1.0f / s;
        float* q = minv + b * 9;
        q[0] = r; q[1] = 0.f; q[2] = -(m02 * r); q[3] = 0.f; q[4] = r; q[5] = -(m12 * r); q[6] = 0.f; q[7] = 0.f; q[8] = 1.f;
    }
}

// ----------------------------------------------------------------------------------------------
// crop mode
// ----------------------------------------------------------------------------------------------
constexpr int CROP_MAX_F = 1664;          // LDS face table capacity (MANO: 1554)
constexpr int CROP_MAX_V = 832;

// crop pixel (i=row, j=col) -> raster pixel (ry, rx); false when the warp reads the zero padding
__device__ __forceinline__ bool crop_to_raster(const float* mi, const int32_t* __restrict__ rowmap, float img_w,
                                               float img_h, int i, int j, int& ry, int& rx) {
    const float x = (float)j, y = (float)i;
    const float sx = (mi[0] * x + mi[1] * y) + mi[2];           // torch CPU matmul order: mul, mul, add, add
    const float sy = (mi[3] * x + mi[4] * y) + mi[5];
    const float gx = (sx / img_w) * 2.0f - 1.0f;
    const float gy = (sy / img_h) * 2.0f - 1.0f;
    const float fx = rintf((gx + 1.0f) * (img_w / 2.0f) - 0.5f);   // grid_sample unnormalize + nearbyint
    const float fy = rintf((gy + 1.0f) * (img_h / 2.0f) - 0.5f);
    if (!(fx >= 0.0f && fx < img_w && fy >= 0.0f && fy < img_h)) return false;
    rx = (int)fx;
    ry = rowmap[(int)fy];
    return ry >= 0;
}

// per-face conservative bbox in raster pixel indices (column rx has x_ndc = 1 - (2rx+1)/S); lo = 1 > hi = 0: the face covers nothing
__device__ __forceinline__ uint2 crop_face_box(float x0, float y0, float z0, float x1, float y1, float z1, float x2, float y2, float z2, int S) {
    const float zmax = max3(z0, z1, z2);
    const float face_area = edge_fn(x0, y0, x1, y1, x2, y2);
    const bool degenerate = (face_area <= kEps && face_area >= -kEps);
    uint2 box = make_uint2(1u, 1u);
    if (!(zmax < 0.0f) && !degenerate) {
        const float xmin = min3(x0, x1, x2), xmax = max3(x0, x1, x2);
        const float ymin = min3(y0, y1, y2), ymax = max3(y0, y1, y2);
        const float fxlo = 0.5f * ((float)S * (1.0f - xmax) - 1.0f), fxhi = 0.5f * ((float)S * (1.0f - xmin) - 1.0f);
        const float fylo = 0.5f * ((float)S * (1.0f - ymax) - 1.0f), fyhi = 0.5f * ((float)S * (1.0f - ymin) - 1.0f);
        if (fxhi > -2.0f && fyhi > -2.0f && fxlo < (float)S + 1.0f && fylo < (float)S + 1.0f) {
            const int xlo = max(0, (int)ceilf(fmaxf(fxlo, -4.0f)) - 1), xhi = min(S - 1, (int)floorf(fminf(fxhi, (float)S + 4.0f)) + 1);
            const int ylo = max(0, (int)ceilf(fmaxf(fylo, -4.0f)) - 1), yhi = min(S - 1, (int)floorf(fminf(fyhi, (float)S + 4.0f)) + 1);
            if (xlo <= xhi && ylo <= yhi)
                box = make_uint2((uint32_t)xlo | ((uint32_t)xhi << 16), (uint32_t)ylo | ((uint32_t)yhi << 16));
        }
    }
    return box;
}

constexpr int CROP_MAX_ROWS = 1024;
constexpr int BIN_TILES = 8;                    // tiles a workgroup bins faces for at a time: two per wave
constexpr int CBIN_CAP = 320;                   // candidate faces per tile list (a fuller tile is scanned by the four waves directly)
constexpr int HEAVY = 64;                       // tiles with more candidates are shared by the four waves

// Crop mode, forward (round 5).  A workgroup owns tiles spread over the crop (8x8 crop pixels each, one lane per pixel) and
// handles them BIN_TILES at a time:
//   1. each wave maps its (up to two) tiles to raster pixels (the two nearest-neighbour maps of resize + crop warp) and
//      publishes the tile's raster bounding box;
//   2. all 256 threads bin the sample's faces into the tiles' candidate lists in LDS (face box vs tile box, LDS-atomic
//      append) -- "LDS staging of per-tile triangle bins";
//   3. a wave gives every candidate of its tile a lane, which walks the tile pixels inside its face's box and merges
//      (z bits << 32 | face) keys into the tile's 64 keys with 64-bit LDS atomic-min: exactly "strict < on z, lowest face
//      index wins exact ties", whatever the order of the list;
//   4. tiles with more than HEAVY candidates (fingers on top of each other: 200-300) are left to the end of the batch and
//      shared by all four waves, 64 list entries each per round.
// Lanes whose edge functions do not all share the sign of the face area cannot be covered (w_i > 0 is a sign statement), so
// the three IEEE divisions are only executed for the surviving lanes -- the accepted arithmetic is still the oracle's,
// operation for operation.
// Why this shape (tools/crop_stamps.py: s_memtime stamps per tile in a diagnostic build): rounds 1-4 gave every wave two
// tiles and had it scan all 1554 face boxes per tile, 64 at a time.  Of a 78 us launch at B = 32 the mean wave needed 13 us;
// a tile with three candidates still cost 12 k cycles (the 25 dependent ballot rounds over the face boxes), the wave that
// owned the heaviest tile 110 k cycles = 51 us; the per-workgroup prologue 700 cycles -- the "25 us floor of projection +
// face boxes" that rounds 2-4 reported was the host's launch overhead inside the timing loop, and a pre-pass launch that
// hoisted the prologue changed nothing (79.0 against 79.5 us).
// LDS: projected vertices 10 KB, face boxes 13 KB, row table 2 KB, lists 5 KB, keys 4 KB -> 35 KB, four workgroups per CU
// (the packed face indices of rounds 1-4, 13 KB, are read from L2 now: three dwords per candidate).
#ifdef CROP_STAMP       // diagnostic build only (tools/crop_stamps.py): per-tile s_memtime stamps of the wave that writes the tile
__device__ unsigned long long* crop_stamp_buf = nullptr;
#endif
__global__ __launch_bounds__(256) void render_crop_fwd_kernel(const float* __restrict__ verts,
                                                              const int32_t* __restrict__ faces,
                                                              const float* __restrict__ minv,
                                                              const int32_t* __restrict__ rowmap,
                                                              const float* __restrict__ center_z,
                                                              const float* __restrict__ cube_z, dsf_camera cam, int V,
                                                              int F, int S, int crop, int wg_per_sample,
                                                              float* __restrict__ img, int32_t* __restrict__ p2f) {
    __shared__ float s_pv[CROP_MAX_V * 3];          // projected verts (x_ndc, y_ndc, z_view)
    __shared__ uint2 s_box[CROP_MAX_F];              // x = lo_x | hi_x<<16, y = lo_y | hi_y<<16 (raster pixels); lo>hi = never
    __shared__ uint16_t s_rows[CROP_MAX_ROWS];
    __shared__ float s_mi[6];
    __shared__ unsigned long long s_key[BIN_TILES][64];      // (z bits << 32 | face) per tile pixel
    __shared__ unsigned short s_bin[BIN_TILES][CBIN_CAP];      // candidate faces per tile
    __shared__ int s_bcnt[BIN_TILES];                         // list lengths (may exceed CBIN_CAP: the tile then scans all faces)
    __shared__ int s_tbox[BIN_TILES][4];                      // raster box of the tile (x0, x1, y0, y1); x1 < x0: nothing is binned for it
    __shared__ int s_state[BIN_TILES];                        // 0 done / nothing to do, 1 heavy: waits for the cooperative pass, 2 heavy + list overflowed
    __shared__ unsigned short s_cand[4][128];                 // per-wave candidates of an overflowed tile
    __shared__ short s_colrx[BIN_TILES][8], s_rowry[BIN_TILES][8];   // raster column of tile column j / raster row of tile row i (-1: padding)
    __shared__ float s_colx[BIN_TILES][8], s_rowy[BIN_TILES][8];     // their NDC coordinates
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.x / wg_per_sample, part = blockIdx.x % wg_per_sample;
    const CamN k = cam_ndc(cam);
    const int n_rows = (int)cam.img_h;

    if (t < 6) s_mi[t] = minv[b * 9 + t];
    for (int r = t; r < n_rows && r < CROP_MAX_ROWS; r += 256) s_rows[r] = (uint16_t)max(rowmap[r], 0);
    for (int v = t; v < V; v += 256) {
        const float* p = verts + ((int64_t)b * V + v) * 3;
        float xn, yn, zv;
        project(k, p[0], p[1], p[2], xn, yn, zv);
        s_pv[v * 3] = xn; s_pv[v * 3 + 1] = yn; s_pv[v * 3 + 2] = zv;
    }
    __syncthreads();
    for (int f = t; f < F; f += 256) {
        const int a = faces[f * 3], c1 = faces[f * 3 + 1], c2 = faces[f * 3 + 2];
        s_box[f] = crop_face_box(s_pv[a * 3], s_pv[a * 3 + 1], s_pv[a * 3 + 2], s_pv[c1 * 3], s_pv[c1 * 3 + 1], s_pv[c1 * 3 + 2],
                                 s_pv[c2 * 3], s_pv[c2 * 3 + 1], s_pv[c2 * 3 + 2], S);
    }

    const bool normalise = (center_z != nullptr);
    const float cz = normalise ? center_z[b] : 0.f;
    const float half = normalise ? cube_z[b] / 2.0f : 1.f;
    const float zmin_c = cz - half, zmax_c = cz + half;
    const int tiles_axis = crop >> 3;
    const int n_tiles = tiles_axis * tiles_axis;
#ifdef CROP_STAMP
    const uint64_t stamp_k0 = __builtin_amdgcn_s_memtime();
#endif
    // tile m of this workgroup.  128-pixel crops (16 x 16 tiles): the workgroup id is the LOW bits of the tile's Morton code and
    // m the high bits, so a workgroup's tiles are spread over the whole crop (the heavy tiles -- overlapping fingers -- sit
    // together); other sizes: part + m * workgroups per sample
    const int tiles_per_wg = (n_tiles + wg_per_sample - 1) / wg_per_sample;
    const bool morton = tiles_axis == 16 && (wg_per_sample & (wg_per_sample - 1)) == 0 && wg_per_sample <= 256;
    const int wg_bits = 31 - __clz(max(wg_per_sample, 1));
    auto tile_of = [&](int m) -> int {
        if (m >= tiles_per_wg) return -1;
        if (!morton) { const int tl = part + m * wg_per_sample; return tl < n_tiles ? tl : -1; }
        const unsigned code = (unsigned)part | ((unsigned)m << wg_bits);
        if (code >= 256u) return -1;
        const unsigned tx = (code & 1u) | ((code >> 1) & 2u) | ((code >> 2) & 4u) | ((code >> 3) & 8u);
        const unsigned ty = ((code >> 1) & 1u) | ((code >> 2) & 2u) | ((code >> 3) & 4u) | ((code >> 4) & 8u);
        return (int)(ty * 16u + tx);
    };
    // one candidate face, one lane: the tile pixels inside its raster box, merged into the tile's keys
    auto face_lane = [&](const int f, const int q) {
        const uint2 box = s_box[f];
        const int xlo = (int)(box.x & 0xFFFF), xhi = (int)(box.x >> 16), ylo = (int)(box.y & 0xFFFF), yhi = (int)(box.y >> 16);
        int j0 = 8, j1 = -1, i0 = 8, i1 = -1;            // pixel sub-rectangle [i0,i1] x [j0,j1] of the tile inside the face's box
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int cx = s_colrx[q][u], cy = s_rowry[q][u];
            if (cx >= xlo && cx <= xhi) { j0 = min(j0, u); j1 = max(j1, u); }
            if (cy >= ylo && cy <= yhi) { i0 = min(i0, u); i1 = max(i1, u); }
        }
        if (j0 > j1 || i0 > i1) return;
        const int a = faces[f * 3] * 3, c1 = faces[f * 3 + 1] * 3, c2 = faces[f * 3 + 2] * 3;
        const float x0 = s_pv[a], y0 = s_pv[a + 1], z0 = s_pv[a + 2];
        const float x1 = s_pv[c1], y1 = s_pv[c1 + 1], z1 = s_pv[c1 + 2];
        const float x2 = s_pv[c2], y2 = s_pv[c2 + 1], z2 = s_pv[c2 + 2];
        const float xmin = min3(x0, x1, x2), xmax = max3(x0, x1, x2);
        const float ymin = min3(y0, y1, y2), ymax = max3(y0, y1, y2);
        const float area = edge_fn(x2, y2, x0, y0, x1, y1) + kEps;
        for (int pi = i0; pi <= i1; ++pi) {
            const float py = s_rowy[q][pi];
            if (s_rowry[q][pi] < 0 || py > ymax || py < ymin) continue;
            for (int pj = j0; pj <= j1; ++pj) {
                const float px = s_colx[q][pj];
                if (s_colrx[q][pj] < 0 || px > xmax || px < xmin) continue;
                const float e0 = edge_fn(px, py, x1, y1, x2, y2);
                const float e1 = edge_fn(px, py, x2, y2, x0, y0);
                const float e2 = edge_fn(px, py, x0, y0, x1, y1);
                // exact pre-test: e/area > 0 needs equal, non-zero signs (area == 0: leave it to the division)
                if (area > 0.0f) { if (!(e0 > 0.0f && e1 > 0.0f && e2 > 0.0f)) continue; }
                else if (area < 0.0f) { if (!(e0 < 0.0f && e1 < 0.0f && e2 < 0.0f)) continue; }
                const float w0 = e0 / area, w1 = e1 / area, w2 = e2 / area;
                const float pz = w0 * z0 + w1 * z1 + w2 * z2;
                if (pz < 0.0f) continue;
                if (!(w0 > 0.0f && w1 > 0.0f && w2 > 0.0f)) continue;
                // (z bits, face) ordered lexicographically == strict < on z with lowest-face ties
                const unsigned long long key = ((unsigned long long)__float_as_uint(pz + 0.0f) << 32) | (unsigned)f;
                atomicMin(&s_key[q][pi * 8 + pj], key);
            }
        }
    };

    for (int m0 = 0; m0 < tiles_per_wg; m0 += BIN_TILES) {
        // ---- 1. this wave's tiles of the batch: slots q = wave and wave + 4 ----
        int tl[2], rx_[2], ry_[2];
        bool valid_[2], sep_[2], scan_[2];
        float xf_[2], yf_[2];
#ifdef CROP_STAMP
        uint64_t stamp_t0[2];
        int stamp_n[2] = {0, 0}, stamp_coop[2] = {0, 0};
#endif
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = wave + 4 * h;
            const int tile = tile_of(m0 + q);
            tl[h] = tile; rx_[h] = 0; ry_[h] = 0; valid_[h] = false; sep_[h] = false; scan_[h] = false; xf_[h] = 0.f; yf_[h] = 0.f;
#ifdef CROP_STAMP
            stamp_t0[h] = __builtin_amdgcn_s_memtime();
#endif
            if (lane == 0) { s_bcnt[q] = 0; s_state[q] = 0; s_tbox[q][0] = 0; s_tbox[q][1] = -1; s_tbox[q][2] = 0; s_tbox[q][3] = -1; }
            if (tile < 0) continue;
            s_key[q][lane] = ~0ull;
            const int ty = tile / tiles_axis, tx = tile % tiles_axis;
            const int i = ty * 8 + (lane >> 3), j = tx * 8 + (lane & 7);
            // crop pixel -> source pixel of the resized image -> raster pixel (zero padding outside)
            int ry = 0, rx = 0;
            bool valid;
            {
                const float x = (float)j, y = (float)i;
                const float sx = (s_mi[0] * x + s_mi[1] * y) + s_mi[2];       // torch CPU matmul order: mul, mul, add, add
                const float sy = (s_mi[3] * x + s_mi[4] * y) + s_mi[5];
                const float gx = (sx / cam.img_w) * 2.0f - 1.0f;
                const float gy = (sy / cam.img_h) * 2.0f - 1.0f;
                const float fx = rintf((gx + 1.0f) * (cam.img_w / 2.0f) - 0.5f);   // grid_sample unnormalize + nearbyint
                const float fy = rintf((gy + 1.0f) * (cam.img_h / 2.0f) - 0.5f);
                valid = fx >= 0.0f && fx < cam.img_w && fy >= 0.0f && fy < cam.img_h;
                if (valid) {
                    rx = (int)fx;
                    const int sy_i = (int)fy;
                    ry = (sy_i < CROP_MAX_ROWS) ? (int)s_rows[sy_i] : rowmap[sy_i];
                }
            }
            const float xf = pix_to_ndc(S - 1 - rx, S), yf = pix_to_ndc(S - 1 - ry, S);
            // raster-pixel bbox of the 64 sample points of this tile (wave reduction)
            int t_x0 = valid ? rx : 0x7fffffff, t_x1 = valid ? rx : -1, t_y0 = valid ? ry : 0x7fffffff, t_y1 = valid ? ry : -1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                t_x0 = min(t_x0, __shfl_xor(t_x0, o, 64)); t_x1 = max(t_x1, __shfl_xor(t_x1, o, 64));
                t_y0 = min(t_y0, __shfl_xor(t_y0, o, 64)); t_y1 = max(t_y1, __shfl_xor(t_y1, o, 64));
            }
            // Is the tile a separable grid (raster column depends on j only, raster row on i only)?  It is
            // unless the ~1e-7 off-diagonal LAPACK noise of M^-1 flips a rounding; then the pixel-parallel scan below.
            const int col_rx = __shfl(valid ? rx : -1, lane & 7, 64), col_ok = __shfl((int)valid, lane & 7, 64);
            const int row_ry = __shfl(valid ? ry : -1, lane & 56, 64), row_ok = __shfl((int)valid, lane & 56, 64);
            const bool sep_lane = valid ? (rx == col_rx && ry == row_ry && col_ok && row_ok) : !(col_ok && row_ok);
            const bool separable = __all(sep_lane);
            const bool any_px = t_x1 >= 0;                        // else the whole tile reads the zero padding
            if (separable && any_px) {
                if (lane < 8) { s_colrx[q][lane] = (short)(col_ok ? rx : -1); s_colx[q][lane] = xf; }                      // lane j: row 0, column j
                if ((lane & 7) == 0) { s_rowry[q][lane >> 3] = (short)(row_ok ? ry : -1); s_rowy[q][lane >> 3] = yf; }   // lane 8i: row i, column 0
                if (lane == 0) { s_tbox[q][0] = t_x0; s_tbox[q][1] = t_x1; s_tbox[q][2] = t_y0; s_tbox[q][3] = t_y1; }
            }
            rx_[h] = rx; ry_[h] = ry; valid_[h] = valid; xf_[h] = xf; yf_[h] = yf; sep_[h] = separable && any_px; scan_[h] = !separable && any_px;
        }
        __syncthreads();                                 // face boxes (first batch), tile boxes, zeroed counters and keys are in LDS
        // ---- 2. bin the faces: every thread its faces against the batch's tile boxes ----
        {
            int bx0[BIN_TILES], bx1[BIN_TILES], by0[BIN_TILES], by1[BIN_TILES];
#pragma unroll
            for (int q = 0; q < BIN_TILES; ++q) { bx0[q] = s_tbox[q][0]; bx1[q] = s_tbox[q][1]; by0[q] = s_tbox[q][2]; by1[q] = s_tbox[q][3]; }
            for (int f = t; f < F; f += 256) {
                const uint2 box = s_box[f];
                const int xlo = (int)(box.x & 0xFFFF), xhi = (int)(box.x >> 16), ylo = (int)(box.y & 0xFFFF), yhi = (int)(box.y >> 16);
                if (xlo > xhi) continue;
#pragma unroll
                for (int q = 0; q < BIN_TILES; ++q) {
                    if (bx0[q] <= bx1[q] && xlo <= bx1[q] && xhi >= bx0[q] && ylo <= by1[q] && yhi >= by0[q]) {
                        const int pos = atomicAdd(&s_bcnt[q], 1);
                        if (pos < CBIN_CAP) s_bin[q][pos] = (unsigned short)f;
                    }
                }
            }
        }
        __syncthreads();
        // ---- 3. this wave's tiles: light ones now, heavy ones are left to pass 4 ----
        auto write_tile = [&](const int h, const int q) {
            float bz = INFINITY;
            int bf = -1;
            if (sep_[h]) {
                const unsigned long long kmin = s_key[q][lane];
                if (kmin != ~0ull) { bz = __uint_as_float((unsigned)(kmin >> 32)); bf = (int)(kmin & 0xFFFFFFFFu); }
            } else if (scan_[h]) {
                // ---- pixel-parallel scan (a tile that is not a separable grid, or whose list overflowed): every lane z-tests its
                //      own pixel against each face whose box meets the tile, in ascending face order ----
                int t_x0 = valid_[h] ? rx_[h] : 0x7fffffff, t_x1 = valid_[h] ? rx_[h] : -1, t_y0 = valid_[h] ? ry_[h] : 0x7fffffff, t_y1 = valid_[h] ? ry_[h] : -1;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    t_x0 = min(t_x0, __shfl_xor(t_x0, o, 64)); t_x1 = max(t_x1, __shfl_xor(t_x1, o, 64));
                    t_y0 = min(t_y0, __shfl_xor(t_y0, o, 64)); t_y1 = max(t_y1, __shfl_xor(t_y1, o, 64));
                }
                const float xf = xf_[h], yf = yf_[h];
                for (int base = 0; base < F; base += 64) {
                    const int fme = base + lane;
                    bool hit = false;
                    if (fme < F) {
                        const uint2 box = s_box[fme];
                        const int xlo = (int)(box.x & 0xFFFF), xhi = (int)(box.x >> 16), ylo = (int)(box.y & 0xFFFF), yhi = (int)(box.y >> 16);
                        hit = xlo <= xhi && xlo <= t_x1 && xhi >= t_x0 && ylo <= t_y1 && yhi >= t_y0;
                    }
                    unsigned long long mask = __ballot(hit);
                    while (mask) {
                        const int bit = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        const int f = base + bit;
                        const int a = faces[f * 3] * 3, c1 = faces[f * 3 + 1] * 3, c2 = faces[f * 3 + 2] * 3;     // wave-uniform
                        const float x0 = s_pv[a], y0 = s_pv[a + 1], x1 = s_pv[c1], y1 = s_pv[c1 + 1], x2 = s_pv[c2], y2 = s_pv[c2 + 1];
                        const float xmin = min3(x0, x1, x2), xmax = max3(x0, x1, x2);
                        const float ymin = min3(y0, y1, y2), ymax = max3(y0, y1, y2);
                        if (!valid_[h] || xf > xmax || xf < xmin || yf > ymax || yf < ymin) continue;
                        const float area = edge_fn(x2, y2, x0, y0, x1, y1) + kEps;
                        const float e0 = edge_fn(xf, yf, x1, y1, x2, y2);
                        const float e1 = edge_fn(xf, yf, x2, y2, x0, y0);
                        const float e2 = edge_fn(xf, yf, x0, y0, x1, y1);
                        if (area > 0.0f) { if (!(e0 > 0.0f && e1 > 0.0f && e2 > 0.0f)) continue; }
                        else if (area < 0.0f) { if (!(e0 < 0.0f && e1 < 0.0f && e2 < 0.0f)) continue; }
                        const float w0 = e0 / area, w1 = e1 / area, w2 = e2 / area;
                        const float z0 = s_pv[a + 2], z1 = s_pv[c1 + 2], z2 = s_pv[c2 + 2];
                        const float pz = w0 * z0 + w1 * z1 + w2 * z2;
                        if (pz < 0.0f) continue;
                        if (!(w0 > 0.0f && w1 > 0.0f && w2 > 0.0f)) continue;
                        if (bf < 0 || pz < bz) { bz = pz; bf = f; }      // faces visited in ascending order: ties keep the lowest
                    }
                }
            }
            // zbuf -> background 0 (:1085) -> zero-padded nearest crop -> normalize_img (:1289-1299)
            float d = (bf >= 0) ? bz : -1.0f;
            if (d <= 0.0f) d = 0.0f;
            if (!valid_[h]) d = 0.0f;
            float o = d;
            if (normalise) {
                if (o == -1.0f || o == 0.0f) o = zmax_c;
                if (o > zmax_c) o = zmax_c;
                if (o < zmin_c) o = zmin_c;
                o = (o - cz) / half;
            }
            const int tile = tl[h];
            const int i = (tile / tiles_axis) * 8 + (lane >> 3), j = (tile % tiles_axis) * 8 + (lane & 7);
            const int64_t idx = ((int64_t)b * crop + i) * crop + j;
            img[idx] = o;
            if (p2f) p2f[idx] = (valid_[h] && bf >= 0 && bz > 0.0f) ? bf : -1;
#ifdef CROP_STAMP
            if (lane == 0 && crop_stamp_buf) {
                unsigned long long* st = crop_stamp_buf + ((size_t)b * n_tiles + tile) * 4;
                st[0] = __builtin_amdgcn_s_memtime() - stamp_t0[h]; st[1] = (unsigned long long)stamp_n[h]; st[2] = (unsigned long long)stamp_coop[h]; st[3] = stamp_t0[h] - stamp_k0;
            }
#endif
        };
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = wave + 4 * h;
            if (tl[h] < 0) continue;
            const int n = sep_[h] ? s_bcnt[q] : 0;
#ifdef CROP_STAMP
            stamp_n[h] = n;
#endif
            // a list that overflowed (a mesh collapsed into one or two tiles -- what a freshly initialised MANO head predicts: all
            // 1554 faces in one tile) is not used: the four waves scan the face boxes themselves in pass 4, 64 at a time each
            if (sep_[h] && n > HEAVY) {
                if (lane == 0) s_state[q] = (n > CBIN_CAP) ? 2 : 1;
#ifdef CROP_STAMP
                stamp_coop[h] = 1;
#endif
                continue;
            }
            if (sep_[h])
                for (int c0 = 0; c0 < n; c0 += 64)
                    if (c0 + lane < n) face_lane((int)s_bin[q][c0 + lane], q);
            write_tile(h, q);                            // (LDS operations of one wave execute in order: its atomics precede its reads)
        }
        __syncthreads();
        // ---- 4. the heavy tiles of the batch, all four waves on each; the owner writes it ----
#pragma unroll
        for (int h = 0; h < 2; ++h) {                    // (h stays a compile-time index of the per-tile registers)
#pragma unroll 1
            for (int r = 0; r < 4; ++r) {
                const int q = r + 4 * h;
                const int state = s_state[q];            // (the same value for every wave: written before the barrier above)
                if (state == 0) continue;
                if (state == 1) {
                    const int n = s_bcnt[q];
                    for (int c0 = wave * 64; c0 < n; c0 += 256)
                        if (c0 + lane < n) face_lane((int)s_bin[q][c0 + lane], q);
                } else {
                    // overflowed list: this wave's blocks of 64 faces (wave, wave + 4, ...), candidates compacted 128 at a time
                    const int bx0 = s_tbox[q][0], bx1 = s_tbox[q][1], by0 = s_tbox[q][2], by1 = s_tbox[q][3];
                    int n_cand = 0;
                    for (int base = wave * 64; base < F || n_cand > 0; base += 256) {
                        const int fme = base + lane;
                        bool hit = false;
                        if (base < F && fme < F) {
                            const uint2 box = s_box[fme];
                            const int xlo = (int)(box.x & 0xFFFF), xhi = (int)(box.x >> 16), ylo = (int)(box.y & 0xFFFF), yhi = (int)(box.y >> 16);
                            hit = xlo <= xhi && xlo <= bx1 && xhi >= bx0 && ylo <= by1 && yhi >= by0;
                        }
                        const unsigned long long mask = __ballot(hit);
                        if (hit) s_cand[wave][n_cand + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)fme;
                        n_cand += __popcll(mask);
                        if (n_cand > 64 || base + 256 >= F) {            // flush (LDS operations of one wave execute in order)
                            __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the wave's own LDS writes have landed
                            __builtin_amdgcn_wave_barrier();
                            for (int c0 = 0; c0 < n_cand; c0 += 64)
                                if (c0 + lane < n_cand) face_lane((int)s_cand[wave][c0 + lane], q);
                            __builtin_amdgcn_s_waitcnt(0xc07f);           // this round's reads of s_cand are done before it is refilled
                            __builtin_amdgcn_wave_barrier();
                            n_cand = 0;
                        }
                        if (base >= F) break;
                    }
                }
                __syncthreads();
                if (r == wave) write_tile(h, q);
            }
        }
        __syncthreads();                                 // the next batch reuses the lists, counters and keys
    }
}

template <bool DET>
__global__ __launch_bounds__(256) void render_crop_bwd_kernel(const float* __restrict__ verts,
                                                              const int32_t* __restrict__ faces,
                                                              const float* __restrict__ minv,
                                                              const int32_t* __restrict__ rowmap,
                                                              const float* __restrict__ center_z,
                                                              const float* __restrict__ cube_z, dsf_camera cam,
                                                              const int32_t* __restrict__ p2f,
                                                              const float* __restrict__ gimg, int V, int F, int S,
                                                              int crop, int wg_per_sample, float* __restrict__ gverts) {
    __shared__ float s_pv[CROP_MAX_V * 3];
    typedef Acc<DET> A;                               // float atomics, or order-independent fixed point (deterministic mode)
    __shared__ typename A::T s_g[CROP_MAX_V * 3];     // grads w.r.t. (x_ndc, y_ndc, z_view) per vertex
    __shared__ float s_mi[6];
    const int t = threadIdx.x;
    const int b = blockIdx.x / wg_per_sample, part = blockIdx.x % wg_per_sample;
    const CamN k = cam_ndc(cam);
    if (t < 6) s_mi[t] = minv[b * 9 + t];
    for (int v = t; v < V; v += 256) {
        const float* p = verts + ((int64_t)b * V + v) * 3;
        float xn, yn, zv;
        project(k, p[0], p[1], p[2], xn, yn, zv);
        s_pv[v * 3] = xn; s_pv[v * 3 + 1] = yn; s_pv[v * 3 + 2] = zv;
        s_g[v * 3] = 0; s_g[v * 3 + 1] = 0; s_g[v * 3 + 2] = 0;
    }
    __syncthreads();
    const bool normalise = (center_z != nullptr);
    const float cz = normalise ? center_z[b] : 0.f;
    const float half = normalise ? cube_z[b] / 2.0f : 1.f;
    const float zmin_c = cz - half, zmax_c = cz + half;
    const int npx = crop * crop;
    for (int q = part * 256 + t; q < npx; q += wg_per_sample * 256) {
        const int64_t idx = (int64_t)b * npx + q;
        const int f = p2f[idx];
        if (f < 0) continue;
        float gz = gimg[idx];
        if (gz == 0.f) continue;
        const int i = q / crop, j = q % crop;
        int ry = 0, rx = 0;
        if (!crop_to_raster(s_mi, rowmap, cam.img_w, cam.img_h, i, j, ry, rx)) continue;
        const float xf = pix_to_ndc(S - 1 - rx, S), yf = pix_to_ndc(S - 1 - ry, S);
        const int a = faces[f * 3], c1 = faces[f * 3 + 1], c2 = faces[f * 3 + 2];
        float v[9] = {s_pv[a * 3], s_pv[a * 3 + 1], s_pv[a * 3 + 2], s_pv[c1 * 3], s_pv[c1 * 3 + 1], s_pv[c1 * 3 + 2],
                      s_pv[c2 * 3], s_pv[c2 * 3 + 1], s_pv[c2 * 3 + 2]};
        if (normalise) {
            // recompute the pixel depth to see whether normalize_img clamped it (no gradient then)
            const float area = edge_fn(v[6], v[7], v[0], v[1], v[3], v[4]) + kEps;
            const float w0 = edge_fn(xf, yf, v[3], v[4], v[6], v[7]) / area;
            const float w1 = edge_fn(xf, yf, v[6], v[7], v[0], v[1]) / area;
            const float w2 = edge_fn(xf, yf, v[0], v[1], v[3], v[4]) / area;
            const float pz = w0 * v[2] + w1 * v[5] + w2 * v[8];
            if (pz > zmax_c || pz < zmin_c) continue;
            gz = gz / half;
        }
        float g[9];
        zbuf_pixel_backward(v, xf, yf, gz, g);
        const int vid[3] = {a, c1, c2};
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            A::add(&s_g[vid[u] * 3], g[u * 3]);
            A::add(&s_g[vid[u] * 3 + 1], g[u * 3 + 1]);
            A::add(&s_g[vid[u] * 3 + 2], g[u * 3 + 2]);
        }
    }
    __syncthreads();
    // chain through the projection: x_ndc = (-X fxn + Z pxn)/Z, y likewise, z_view = Z
    for (int v = t; v < V; v += 256) {
        const float gxn = A::get(s_g[v * 3]), gyn = A::get(s_g[v * 3 + 1]), gzv = A::get(s_g[v * 3 + 2]);
        if (gxn == 0.f && gyn == 0.f && gzv == 0.f) continue;
        const float Z = s_pv[v * 3 + 2];
        const float gX = -gxn * k.fxn / Z, gY = -gyn * k.fyn / Z;
        const float gZ = gzv + gxn * (k.pxn - s_pv[v * 3]) / Z + gyn * (k.pyn - s_pv[v * 3 + 1]) / Z;
        float* o = gverts + ((int64_t)b * V + v) * 3;
        atomicAdd(o, gX); atomicAdd(o + 1, gY); atomicAdd(o + 2, gZ);
    }
}

inline int crop_wg_per_sample(int B, int tiles) {
    // ~2 workgroups per CU on the 256-CU chip (each re-stages the sample's vertices / face boxes),
    // at least one tile per wave
    int g = 1;
    int target = (B > 32) ? 2048 : 1024;       // measured at B = 32 / 64 / 128: 1024 -> 49.8 / 118.9 / 171.3 us, 2048 -> 59.3 / 111.8 / 133.3 us
    if (const char* e = getenv("DSF_CROP_WG_TARGET")) target = atoi(e);
    while (g < tiles / 4 && B * g < target) g *= 2;
    return g;
}

}  // namespace

extern "C" int dsf_project_face_verts(const float* verts, const int32_t* faces, const dsf_camera* cam, int N, int V,
                                      int F, float* face_verts, dsf_stream_t stream) {
    DSF_CHECK_ARG(verts && faces && cam && face_verts && N >= 0 && V > 0 && F >= 0);
    const int64_t total = (int64_t)N * F * 3;
    if (total == 0) return DSF_OK;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(project_face_verts_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, verts, faces, *cam, N, V,
                       F, face_verts);
    return dsf_launch_status();
}

extern "C" int dsf_rasterize_meshes(const float* face_verts, const int64_t* mesh_to_face_first_idx,
                                    const int64_t* num_faces_per_mesh, int N, int64_t F_total, int image_size,
                                    float blur_radius, int faces_per_pixel, int perspective_correct,
                                    int clip_barycentric_coords, int cull_backfaces, int64_t* pix_to_face, float* zbuf,
                                    float* bary, float* dists, float* workspace, dsf_stream_t stream) {
    DSF_CHECK_ARG(face_verts && mesh_to_face_first_idx && num_faces_per_mesh && pix_to_face && zbuf);
    DSF_CHECK_ARG(N >= 0 && image_size > 0 && F_total >= 0 && F_total < (1ll << 31));
    if (blur_radius != 0.0f || faces_per_pixel != 1 || perspective_correct || clip_barycentric_coords || cull_backfaces)
        return DSF_ERR_UNSUPPORTED;
    if (N == 0) return DSF_OK;
    if (workspace)
        hipLaunchKernelGGL(mesh_bbox_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, face_verts,
                           mesh_to_face_first_idx, num_faces_per_mesh, workspace);
    const int tiles_x = (image_size + TILE_W - 1) / TILE_W, tiles_y = (image_size + TILE_H - 1) / TILE_H;
    const int groups = (N + 7) / 8;
    hipLaunchKernelGGL(raster_full_kernel, dim3(groups * 8 * tiles_x * tiles_y), dim3(256), 0, (hipStream_t)stream,
                       face_verts, mesh_to_face_first_idx, num_faces_per_mesh, workspace, N, image_size, tiles_x, tiles_y,
                       pix_to_face, zbuf, bary, dists);
    return dsf_launch_status();
}

extern "C" int dsf_rasterize_meshes_backward(const float* face_verts, const int64_t* pix_to_face,
                                             const float* grad_zbuf, const float* grad_bary, const float* grad_dists,
                                             int N, int64_t F_total, int image_size, float* grad_face_verts,
                                             dsf_stream_t stream) {
    DSF_CHECK_ARG(face_verts && pix_to_face && grad_zbuf && grad_face_verts && N >= 0 && image_size > 0);
    if (grad_bary || grad_dists) return DSF_ERR_UNSUPPORTED;     // the reference consumes zbuf only (:1023)
    if (dsf_zero_async(grad_face_verts, sizeof(float) * 9 * F_total, (hipStream_t)stream) != hipSuccess)
        return DSF_ERR_LAUNCH;
    const int64_t npix = (int64_t)N * image_size * image_size;
    if (npix == 0) return DSF_OK;
    const int grid = (int)((npix + 255) / 256 < 8192 ? (npix + 255) / 256 : 8192);
    hipLaunchKernelGGL(raster_full_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, face_verts, pix_to_face,
                       grad_zbuf, npix, image_size, grad_face_verts);
    return dsf_launch_status();
}

extern "C" int dsf_crop_setup(const float* center3d, const float* cube, const dsf_camera* cam, int B, int crop,
                              float* center2d, float* M, int32_t* bounds, float* minv_closed, dsf_stream_t stream) {
    DSF_CHECK_ARG(center3d && cube && cam && center2d && M && B >= 0 && crop > 0);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(crop_setup_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, center3d, cube, *cam, B,
                       crop, center2d, M, bounds, minv_closed);
    return dsf_launch_status();
}

extern "C" int dsf_render_crop_forward(const float* verts, const int32_t* faces, const float* minv,
                                       const int32_t* resize_rowmap, const float* center_z, const float* cube_z,
                                       const dsf_camera* cam, int B, int V, int F, int raster_size, int crop, float* img,
                                       int32_t* pix_to_face, dsf_stream_t stream) {
    DSF_CHECK_ARG(verts && faces && minv && resize_rowmap && cam && img);
    DSF_CHECK_ARG(B >= 0 && V > 0 && V <= CROP_MAX_V && F >= 0 && F <= CROP_MAX_F && raster_size <= 65535);
    DSF_CHECK_ARG(crop > 0 && (crop & 7) == 0 && crop <= 2048 && (center_z == nullptr) == (cube_z == nullptr));
    if ((int)cam->img_w != raster_size) return DSF_ERR_UNSUPPORTED;   // resize keeps columns (mano_layer.py:1233-1242)
    if (B == 0) return DSF_OK;
    const int tiles = (crop >> 3) * (crop >> 3);
    const int g = crop_wg_per_sample(B, tiles);
    hipLaunchKernelGGL(render_crop_fwd_kernel, dim3(B * g), dim3(256), 0, (hipStream_t)stream, verts, faces, minv,
                       resize_rowmap, center_z, cube_z, *cam, V, F, raster_size, crop, g, img, pix_to_face);
    return dsf_launch_status();
}

#ifdef CROP_STAMP
extern "C" int dsf_crop_stamp_buffer(void* p) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    return hipMemcpyToSymbol(HIP_SYMBOL(crop_stamp_buf), &q, sizeof(q)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int dsf_render_crop_backward(const float* verts, const int32_t* faces, const float* minv,
                                        const int32_t* resize_rowmap, const float* center_z, const float* cube_z,
                                        const dsf_camera* cam, const int32_t* pix_to_face, const float* grad_img, int B,
                                        int V, int F, int raster_size, int crop, float* grad_verts,
                                        dsf_stream_t stream) {
    DSF_CHECK_ARG(verts && faces && minv && resize_rowmap && cam && pix_to_face && grad_img && grad_verts);
    DSF_CHECK_ARG(B >= 0 && V > 0 && V <= CROP_MAX_V && F >= 0 && crop > 0);
    DSF_CHECK_ARG((center_z == nullptr) == (cube_z == nullptr));
    if (dsf_zero_async(grad_verts, sizeof(float) * 3 * (size_t)B * V, (hipStream_t)stream) != hipSuccess)
        return DSF_ERR_LAUNCH;
    if (B == 0) return DSF_OK;
    if (dsf_deterministic()) {
        // one workgroup per sample: the fixed-point LDS table holds the whole sum, the flush adds one value onto a zero
        hipLaunchKernelGGL(render_crop_bwd_kernel<true>, dim3(B), dim3(256), 0, (hipStream_t)stream, verts, faces, minv,
                           resize_rowmap, center_z, cube_z, *cam, pix_to_face, grad_img, V, F, raster_size, crop, 1, grad_verts);
        return dsf_launch_status();
    }
    const int g = (B >= 512) ? 1 : (B >= 128 ? 2 : 4);
    hipLaunchKernelGGL(render_crop_bwd_kernel<false>, dim3(B * g), dim3(256), 0, (hipStream_t)stream, verts, faces, minv,
                       resize_rowmap, center_z, cube_z, *cam, pix_to_face, grad_img, V, F, raster_size, crop, g,
                       grad_verts);
    return dsf_launch_status();
}
