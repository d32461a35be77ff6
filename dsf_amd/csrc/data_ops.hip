// Depth data path on the device: the reference's test-phase `loader.__getitem__` crop
// (`Crop_Image_deep_pp` + `normalize_img`, data/render_loader.py:748-810, 738-745, with `comToBounds` :356-364 and
// `getCrop` :867-905) straight from raw depth frames resident in HBM, replacing its numpy / cv2 DataLoader workers.
// Integer decisions (crop bounds, resize size, nearest-neighbour source pixel) are evaluated in double exactly as the
// numpy code does; one workgroup per frame, two passes over the dsize x dsize output (maximum, then normalisation): a
// frame's crop touches at most ~(2 * cube / z * f)^2 source pixels, so the kernel is latency- not bandwidth-bound.
#include "common.h"

namespace {

struct CropGeom {
    int xs, ys, wb, hb, sz_w, sz_h, x0, y0;
    double ifx, ify, zs, ze, scale;
};

__device__ __forceinline__ CropGeom crop_geom(const double* com, const double* size, double fx, double fy, int dsize) {
    CropGeom g;
    const double u = com[0], v = com[1], z = com[2];
    g.zs = z - size[2] / 2.;
    g.ze = z + size[2] / 2.;
    g.xs = (int)floor((u * z / fx - size[0] / 2.) / z * fx + 0.5);
    const int xe = (int)floor((u * z / fx + size[0] / 2.) / z * fx + 0.5);
    g.ys = (int)floor((v * z / fy - size[1] / 2.) / z * fy + 0.5);
    const int ye = (int)floor((v * z / fy + size[1] / 2.) / z * fy + 0.5);
    g.wb = xe - g.xs; g.hb = ye - g.ys;
    if (g.wb > g.hb) { g.sz_w = dsize; g.sz_h = (int)((double)(g.hb * dsize) / (double)g.wb); }
    else { g.sz_w = (int)((double)(g.wb * dsize) / (double)g.hb); g.sz_h = dsize; }
    g.scale = (g.hb > g.wb) ? (double)g.sz_h / (double)g.hb : (double)g.sz_w / (double)g.wb;
    g.ifx = 1.0 / ((double)g.sz_w / (double)g.wb);      // OpenCV resizeNN: source = min(floor(dst * ifx), size - 1)
    g.ify = 1.0 / ((double)g.sz_h / (double)g.hb);
    g.x0 = (int)floor(dsize / 2. - g.sz_w / 2.);
    g.y0 = (int)floor(dsize / 2. - g.sz_h / 2.);
    return g;
}

// un-normalised crop value of output pixel (oy, ox)
// (T = float, or uint16_t: the sensors' raw 16-bit millimetre frames -- NYU / ICVL / HANDS17 png, render_loader.py:201-218 --
//  whose conversion to float32 is exact, so both element types give the same bits)
template <typename T>
__device__ __forceinline__ float crop_pixel(const T* __restrict__ frame, int Hd, int Wd, const CropGeom& g, int oy, int ox) {
    const int ry = oy - g.y0, rx = ox - g.x0;
    if ((unsigned)ry >= (unsigned)g.sz_h || (unsigned)rx >= (unsigned)g.sz_w) return 0.f;
    const int sy = min((int)floor(ry * g.ify), g.hb - 1), sx = min((int)floor(rx * g.ifx), g.wb - 1);
    const int iy = g.ys + sy, ix = g.xs + sx;
    if ((unsigned)iy >= (unsigned)Hd || (unsigned)ix >= (unsigned)Wd) return 0.f;
    float v = (float)frame[(int64_t)iy * Wd + ix];
    if (v != 0.f) {
        if ((double)v < g.zs) v = (float)g.zs;           // in front of the cube: onto its front face
        else if ((double)v > g.ze) v = 0.f;              // behind it: background
    }
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void depth_crop_normalize_kernel(const T* __restrict__ depth, const double* __restrict__ com,
                                                                   const double* __restrict__ cube, double fx, double fy, int Hd,
                                                                   int Wd, int dsize, float* __restrict__ img,
                                                                   double* __restrict__ trans, float* __restrict__ raw) {
    __shared__ float red[4];
    const int b = blockIdx.x, t = threadIdx.x;
    const double* c = com + 3 * b;
    const double* s = cube + 3 * b;
    const CropGeom g = crop_geom(c, s, fx, fy, dsize);
    const T* frame = depth + (int64_t)b * Hd * Wd;
    const int n = dsize * dsize;
    if (t == 0 && trans) {                               // off . scale . trans (render_loader.py:810), row-major 3 x 3
        double* m = trans + 9 * b;
        m[0] = g.scale; m[1] = 0.; m[2] = g.scale * (double)(-g.xs) + (double)g.x0;
        m[3] = 0.; m[4] = g.scale; m[5] = g.scale * (double)(-g.ys) + (double)g.y0;
        m[6] = 0.; m[7] = 0.; m[8] = 1.;
    }
    float mx = 0.f;                                      // the crop holds zeros or positive depths
    for (int p = t; p < n; p += 256) {
        const float v = crop_pixel(frame, Hd, Wd, g, p / dsize, p % dsize);
        if (raw) raw[(int64_t)b * n + p] = v;
        mx = fmaxf(mx, v);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    const float premax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const double z = c[2], half = s[2] / 2.;
    const double far_d = z + half, near_d = z - half;
    const float far_f = (float)far_d, near_f = (float)near_d, z_f = (float)z, half_f = (float)half;
    for (int p = t; p < n; p += 256) {
        float v = crop_pixel(frame, Hd, Wd, g, p / dsize, p % dsize);
        if (v == premax) v = far_f;                      // normalize_img (:738-745), statement by statement
        if (v == 0.f) v = far_f;
        if ((double)v >= far_d) v = far_f;
        if ((double)v <= near_d) v = near_f;
        v = v - z_f;
        v = v / half_f;
        img[(int64_t)b * n + p] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Training-phase augmentation of an already cropped frame: `augmentCrop` (data/render_loader.py:653-695)
// = `rotateHand` (:458-497, cv2.getRotationMatrix2D + cv2.warpAffine), `moveCoM` (:427-456) or `scaleHand` (:499-527)
// (both through `recropHand` :403-424, cv2.warpPerspective), then `normalize_img` (:738-745) -- one workgroup per frame.
// OpenCV's nearest-neighbour rules as restated in oracle/data_ref.py (imgproc/imgwarp.cpp): warpPerspective evaluates the
// source position in double per 64 x 16 block and rounds half to even; warpAffine works in fixed point with 10 fractional
// bits.  Coordinates in double, rounded to float where the reference stores float32 (jointImgTo3D / joint3DToImg).
// ------------------------------------------------------------------------------------------------------------------
struct AugGeom {
    int kind;                          // 0 copy, 1 perspective (scale + shift), 2 affine fixed point
    int moved;                         // the centre moved ('com'): joints are re-expressed relative to the new one
    double mi00, mi02, mi12;           // kind 1: inverse map  x_src = mi00 x + mi02,  y_src = mi00 y + mi12
    double a00, a01, a02, a10, a11, a12;   // kind 2: inverted rotation (dst -> src)
    double zs, ze;                     // kind 1: crop cube in z (of the centre / cube recropHand is given)
    float nv;                          // values below it are outliers of the warp -> background
    double com[3], cube[3], M[6];      // outputs
};

__device__ __forceinline__ void aug_img_to_3d(const double* uvd, double fx, double fy, double fu, double fv, int flip, float* o) {
    o[0] = (float)((uvd[0] - fu) * uvd[2] / fx);
    o[1] = (float)((double)flip * (uvd[1] - fv) * uvd[2] / fy);
    o[2] = (float)uvd[2];
}
__device__ __forceinline__ void aug_to_img(const double* xyz, double fx, double fy, double fu, double fv, int flip, float* o) {
    o[0] = (float)(xyz[0] * fx / xyz[2] + fu);
    o[1] = (float)((double)flip * xyz[1] * fy / xyz[2] + fv);
    o[2] = (float)xyz[2];
}
// comToTransform (:366-401): s, tx, ty of [[s,0,tx],[0,s,ty],[0,0,1]]
__device__ __forceinline__ void aug_com_to_transform(const double* com, const double* size, double fx, double fy, int S, double* m) {
    const double u = com[0], v = com[1], z = com[2];
    const int xs = (int)floor((u * z / fx - size[0] / 2.) / z * fx + 0.5), xe = (int)floor((u * z / fx + size[0] / 2.) / z * fx + 0.5);
    const int ys = (int)floor((v * z / fy - size[1] / 2.) / z * fy + 0.5), ye = (int)floor((v * z / fy + size[1] / 2.) / z * fy + 0.5);
    const int wb = xe - xs, hb = ye - ys;
    double sc, sz0, sz1;
    if (wb > hb) { sc = (double)S / (double)wb; sz0 = (double)S; sz1 = (double)(hb * S) / (double)wb; }
    else { sc = (double)S / (double)hb; sz0 = (double)(wb * S) / (double)hb; sz1 = (double)S; }
    const int x0 = (int)floor(S / 2. - sz0 / 2.), y0 = (int)floor(S / 2. - sz1 / 2.);
    m[0] = sc; m[1] = sc * (double)(-xs) + (double)x0; m[2] = sc * (double)(-ys) + (double)y0;
}

__global__ __launch_bounds__(256) void depth_augment_kernel(const float* __restrict__ crop, const float* __restrict__ joints,
                                                            const double* __restrict__ com, const double* __restrict__ cube,
                                                            const double* __restrict__ M, const int32_t* __restrict__ mode,
                                                            const double* __restrict__ off, const double* __restrict__ rot,
                                                            const double* __restrict__ sc, double fx, double fy, double fu,
                                                            double fv, int flip, int S, int J, float* __restrict__ img,
                                                            float* __restrict__ joints_out, double* __restrict__ cube_out,
                                                            double* __restrict__ com_out, double* __restrict__ M_out) {
    __shared__ float s_mx[4], s_mn[4];
    __shared__ AugGeom g;
    __shared__ float s_c3[3], s_c3n[3];                 // jointImgTo3D of the old / new centre
    __shared__ double s_cs, s_sn;                       // cos / sin of the joint rotation
    const int b = blockIdx.x, t = threadIdx.x, n = S * S;
    const float* src = crop + (int64_t)b * n;
    float mx = 0.f, mn = INFINITY;
    for (int p = t; p < n; p += 256) { const float v = src[p]; mx = fmaxf(mx, v); if (v > 0.f) mn = fminf(mn, v); }
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o)); mn = fminf(mn, __shfl_xor(mn, o)); }
    if ((t & 63) == 0) { s_mx[t >> 6] = mx; s_mn[t >> 6] = mn; }
    __syncthreads();
    const float premax = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    const float dmin = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
    if (t == 0) {
        const double* c = com + 3 * b; const double* cb = cube + 3 * b; const double* m = M + 9 * b;
        const int md = mode[b];
        g.kind = 0; g.moved = 0;
        g.nv = dmin - 1.0f;                              // np.min(dpt[dpt > 0]) - 1 (float32)
        for (int a = 0; a < 3; ++a) { g.com[a] = c[a]; g.cube[a] = cb[a]; }
        g.M[0] = m[0]; g.M[1] = m[1]; g.M[2] = m[2]; g.M[3] = m[3]; g.M[4] = m[4]; g.M[5] = m[5];
        float c3[3];
        aug_img_to_3d(c, fx, fy, fu, fv, flip, c3);
        s_c3[0] = c3[0]; s_c3[1] = c3[1]; s_c3[2] = c3[2];
        s_c3n[0] = c3[0]; s_c3n[1] = c3[1]; s_c3n[2] = c3[2];
        s_cs = 1.0; s_sn = 0.0;
        auto close0 = [](double v, double ref) { return fabs(v - ref) <= 1e-8 + 1e-5 * fabs(ref); };     // np.allclose
        if (premax != 0.f) {
            if (md == 1 && !(close0(off[3 * b], 0.) && close0(off[3 * b + 1], 0.) && close0(off[3 * b + 2], 0.))) {          // 'com'
                const double moved[3] = {(double)c3[0] + off[3 * b], (double)c3[1] + off[3 * b + 1], (double)c3[2] + off[3 * b + 2]};
                float nc[3];
                aug_to_img(moved, fx, fy, fu, fv, flip, nc);
                const double ncd[3] = {(double)nc[0], (double)nc[1], (double)nc[2]};
                if (!(close0(c[2], 0.) || close0(ncd[2], 0.))) {
                    double mn3[3];
                    aug_com_to_transform(ncd, cb, fx, fy, S, mn3);
                    // A = Mnew . inv(M) (scale + shift), then its inverse, each entry with one rounding (oracle: affine_inverse)
                    const double r = 1.0 / m[0];
                    const double a00 = mn3[0] * r, a02 = mn3[0] * (-(m[2] * r)) + mn3[1], a12 = mn3[0] * (-(m[5] * r)) + mn3[2];
                    const double ra = 1.0 / a00;
                    g.kind = 1; g.mi00 = ra; g.mi02 = -(a02 * ra); g.mi12 = -(a12 * ra);
                    g.zs = ncd[2] - cb[2] / 2.; g.ze = ncd[2] + cb[2] / 2.;
                    g.M[0] = mn3[0]; g.M[1] = 0.; g.M[2] = mn3[1]; g.M[3] = 0.; g.M[4] = mn3[0]; g.M[5] = mn3[2];
                }
                g.moved = 1;
                float n3[3];
                aug_img_to_3d(ncd, fx, fy, fu, fv, flip, n3);
                s_c3n[0] = n3[0]; s_c3n[1] = n3[1]; s_c3n[2] = n3[2];
                g.com[0] = ncd[0]; g.com[1] = ncd[1]; g.com[2] = ncd[2];
            } else if (md == 0 && !close0(rot[b], 0.)) {                                                                   // 'rot'
                const double r = fmod(fmod(rot[b], 360.) + 360., 360.);                 // np.mod: result has the divisor's sign
                // getRotationMatrix2D((S/2, S/2), -r, 1), inverted as warpAffine does
                const double ang = -r * 3.14159265358979323846 / 180.0;
                const double al = cos(ang), be = sin(ang), cx = (double)(float)(S / 2), cy = cx;
                double m0 = al, m1 = be, m2 = (1 - al) * cx - be * cy, m3 = -be, m4 = al, m5 = be * cx + (1 - al) * cy;
                double D = m0 * m4 - m1 * m3;
                D = D != 0. ? 1. / D : 0.;
                const double A11 = m4 * D, A22 = m0 * D;
                m0 = A11; m1 *= -D; m3 *= -D; m4 = A22;
                const double b1 = -m0 * m2 - m1 * m5, b2 = -m3 * m2 - m4 * m5;
                g.kind = 2; g.a00 = m0; g.a01 = m1; g.a02 = b1; g.a10 = m3; g.a11 = m4; g.a12 = b2;
                const double a = r * 3.14159265358979323846 / 180.;
                s_cs = cos(a); s_sn = sin(a);
            } else if (md == 2 && !close0(sc[b], 1.)) {                                                                    // 'sc'
                const double ncube[3] = {cb[0] * sc[b], cb[1] * sc[b], cb[2] * sc[b]};
                if (!close0(c[2], 0.)) {
                    double mn3[3];
                    aug_com_to_transform(c, ncube, fx, fy, S, mn3);
                    const double r = 1.0 / m[0];
                    const double a00 = mn3[0] * r, a02 = mn3[0] * (-(m[2] * r)) + mn3[1], a12 = mn3[0] * (-(m[5] * r)) + mn3[2];
                    const double ra = 1.0 / a00;
                    g.kind = 1; g.mi00 = ra; g.mi02 = -(a02 * ra); g.mi12 = -(a12 * ra);
                    g.zs = c[2] - cb[2] / 2.; g.ze = c[2] + cb[2] / 2.;              // recropHand gets the OLD cube (:521)
                    g.M[0] = mn3[0]; g.M[1] = 0.; g.M[2] = mn3[1]; g.M[3] = 0.; g.M[4] = mn3[0]; g.M[5] = mn3[2];
                }
                g.cube[0] = ncube[0]; g.cube[1] = ncube[1]; g.cube[2] = ncube[2];
            }
        }
        for (int a = 0; a < 3; ++a) { cube_out[3 * b + a] = g.cube[a]; com_out[3 * b + a] = g.com[a]; }
        double* mo = M_out + 9 * b;
        for (int a = 0; a < 6; ++a) mo[a] = g.M[a];
        mo[6] = 0.; mo[7] = 0.; mo[8] = 1.;
    }
    __syncthreads();
    // ---- pixels: warp, outlier / cube thresholds, normalize_img with the INPUT crop's maximum ----
    const double z = g.com[2], half = g.cube[2] / 2.;
    const double far_d = z + half, near_d = z - half;
    const float far_f = (float)far_d, near_f = (float)near_d, z_f = (float)z, half_f = (float)half;
    const int bw = (S >= 64) ? 64 : S;                  // warpPerspective block width for S x S (16 rows x 64 columns)
    for (int p = t; p < n; p += 256) {
        const int y = p / S, x = p % S;
        float v;
        if (g.kind == 0) {
            v = src[p];
        } else if (g.kind == 1) {
            const double bx = (double)((x / bw) * bw), x1 = (double)x - bx;
            const double X0 = (g.mi00 * bx + 0.0 * (double)y) + g.mi02, Y0 = (0.0 * bx + g.mi00 * (double)y) + g.mi12;
            const double fxs = (X0 + g.mi00 * x1) * 1.0, fys = (Y0 + 0.0 * x1) * 1.0;
            const long long sx = (long long)rint(fxs), sy = (long long)rint(fys);
            v = (sx >= 0 && sx < S && sy >= 0 && sy < S) ? src[sy * S + sx] : 0.f;
            if (v < g.nv) v = 0.f;
            if ((double)v < g.zs && v != 0.f) v = (float)g.zs;
            else if ((double)v > g.ze && v != 0.f) v = 0.f;
        } else {
            const long long X0 = (long long)rint((g.a01 * (double)y + g.a02) * 1024.) + 512, Y0 = (long long)rint((g.a11 * (double)y + g.a12) * 1024.) + 512;
            const long long sx = (X0 + (long long)rint(g.a00 * (double)x * 1024.)) >> 10, sy = (Y0 + (long long)rint(g.a10 * (double)x * 1024.)) >> 10;
            v = (sx >= 0 && sx < S && sy >= 0 && sy < S) ? src[sy * S + sx] : 0.f;
            if (dmin != INFINITY && v < g.nv) v = 0.f;
        }
        if (v == premax) v = far_f;                      // normalize_img (:738-745), statement by statement
        if (v == 0.f) v = far_f;
        if ((double)v >= far_d) v = far_f;
        if ((double)v <= near_d) v = near_f;
        v = v - z_f;
        v = v / half_f;
        img[(int64_t)b * n + p] = v;
    }
    // ---- joints ----
    for (int jn = t; jn < J; jn += 256) {
        const float* ji = joints + ((int64_t)b * J + jn) * 3;
        float* jo = joints_out + ((int64_t)b * J + jn) * 3;
        if (g.kind == 2) {
            // joint3DToImg(joints + com3D) -> rotatePoint2D about com[0:2] (float32 after every statement) -> jointImgTo3D - com3D
            const float w3[3] = {ji[0] + s_c3[0], ji[1] + s_c3[1], ji[2] + s_c3[2]};
            const double wd[3] = {(double)w3[0], (double)w3[1], (double)w3[2]};
            float uv[3];
            aug_to_img(wd, fx, fy, fu, fv, flip, uv);
            const float p0 = (float)((double)uv[0] - com[3 * b]), p1 = (float)((double)uv[1] - com[3 * b + 1]);
            const float r0 = (float)((double)p0 * s_cs - (double)p1 * s_sn), r1 = (float)((double)p0 * s_sn + (double)p1 * s_cs);
            const float q0 = (float)((double)r0 + com[3 * b]), q1 = (float)((double)r1 + com[3 * b + 1]);
            const double qd[3] = {(double)q0, (double)q1, (double)uv[2]};
            float o3[3];
            aug_img_to_3d(qd, fx, fy, fu, fv, flip, o3);
            jo[0] = o3[0] - s_c3[0]; jo[1] = o3[1] - s_c3[1]; jo[2] = o3[2] - s_c3[2];
        } else if (g.moved) {
            // moveCoM: joints + jointImgTo3D(com) - jointImgTo3D(new_com) (float32 arrays)
            jo[0] = (ji[0] + s_c3[0]) - s_c3n[0]; jo[1] = (ji[1] + s_c3[1]) - s_c3n[1]; jo[2] = (ji[2] + s_c3[2]) - s_c3n[2];
        } else {
            jo[0] = ji[0]; jo[1] = ji[1]; jo[2] = ji[2];
        }
    }
}

}  // namespace

extern "C" int dsf_depth_augment_crop(const float* crop, const float* joints, const double* com, const double* cube,
                                      const double* M, const int32_t* mode, const double* off, const double* rot,
                                      const double* sc, double fx, double fy, double fu, double fv, int flip, int B, int S, int J,
                                      float* img, float* joints_out, double* cube_out, double* com_out, double* M_out,
                                      dsf_stream_t stream) {
    DSF_CHECK_ARG(crop && joints && com && cube && M && mode && off && rot && sc && img && joints_out && cube_out && com_out && M_out);
    DSF_CHECK_ARG(B >= 0 && S > 0 && S <= 4096 && J >= 0 && fx > 0. && fy > 0. && (flip == 1 || flip == -1));
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(depth_augment_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, crop, joints, com, cube, M, mode, off, rot, sc,
                       fx, fy, fu, fv, flip, S, J, img, joints_out, cube_out, com_out, M_out);
    return dsf_launch_status();
}

extern "C" int dsf_depth_crop_normalize(const float* depth, const double* com, const double* cube, double fx, double fy, int B,
                                        int Hd, int Wd, int dsize, float* img, double* trans, float* raw_crop,
                                        dsf_stream_t stream) {
    DSF_CHECK_ARG(depth && com && cube && img && B >= 0 && Hd > 0 && Wd > 0 && dsize > 0 && fx > 0. && fy > 0.);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(depth_crop_normalize_kernel<float>, dim3(B), dim3(256), 0, (hipStream_t)stream, depth, com, cube, fx, fy, Hd, Wd,
                       dsize, img, trans, raw_crop);
    return dsf_launch_status();
}

// the same on RAW 16-bit frames ("consuming raw 640x480 u16 depth"): no float32 copy of the frame is ever made
extern "C" int dsf_depth_crop_normalize_u16(const uint16_t* depth, const double* com, const double* cube, double fx, double fy, int B,
                                            int Hd, int Wd, int dsize, float* img, double* trans, float* raw_crop,
                                            dsf_stream_t stream) {
    DSF_CHECK_ARG(depth && com && cube && img && B >= 0 && Hd > 0 && Wd > 0 && dsize > 0 && fx > 0. && fy > 0.);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(depth_crop_normalize_kernel<uint16_t>, dim3(B), dim3(256), 0, (hipStream_t)stream, depth, com, cube, fx, fy, Hd,
                       Wd, dsize, img, trans, raw_crop);
    return dsf_launch_status();
}
