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
__device__ __forceinline__ float crop_pixel(const float* __restrict__ frame, int Hd, int Wd, const CropGeom& g, int oy, int ox) {
    const int ry = oy - g.y0, rx = ox - g.x0;
    if ((unsigned)ry >= (unsigned)g.sz_h || (unsigned)rx >= (unsigned)g.sz_w) return 0.f;
    const int sy = min((int)floor(ry * g.ify), g.hb - 1), sx = min((int)floor(rx * g.ifx), g.wb - 1);
    const int iy = g.ys + sy, ix = g.xs + sx;
    if ((unsigned)iy >= (unsigned)Hd || (unsigned)ix >= (unsigned)Wd) return 0.f;
    float v = frame[(int64_t)iy * Wd + ix];
    if (v != 0.f) {
        if ((double)v < g.zs) v = (float)g.zs;           // in front of the cube: onto its front face
        else if ((double)v > g.ze) v = 0.f;              // behind it: background
    }
    return v;
}

__global__ __launch_bounds__(256) void depth_crop_normalize_kernel(const float* __restrict__ depth, const double* __restrict__ com,
                                                                   const double* __restrict__ cube, double fx, double fy, int Hd,
                                                                   int Wd, int dsize, float* __restrict__ img,
                                                                   double* __restrict__ trans, float* __restrict__ raw) {
    __shared__ float red[4];
    const int b = blockIdx.x, t = threadIdx.x;
    const double* c = com + 3 * b;
    const double* s = cube + 3 * b;
    const CropGeom g = crop_geom(c, s, fx, fy, dsize);
    const float* frame = depth + (int64_t)b * Hd * Wd;
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

}  // namespace

extern "C" int dsf_depth_crop_normalize(const float* depth, const double* com, const double* cube, double fx, double fy, int B,
                                        int Hd, int Wd, int dsize, float* img, double* trans, float* raw_crop,
                                        dsf_stream_t stream) {
    DSF_CHECK_ARG(depth && com && cube && img && B >= 0 && Hd > 0 && Wd > 0 && dsize > 0 && fx > 0. && fy > 0.);
    if (B == 0) return DSF_OK;
    hipLaunchKernelGGL(depth_crop_normalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, depth, com, cube, fx, fy, Hd, Wd,
                       dsize, img, trans, raw_crop);
    return dsf_launch_status();
}
