// Self-intersection volume of the watertight hand parts: the reference's offline metric
// `self_intersection` (/root/reference/eval_coll.py:611-626; the same construction in util/intersect.py:102-107) --
//     for every pair (s, t), t > s, not parent / child:   volume += #{ surface voxels of part t inside part s } * pitch^3
// where the voxels come from trimesh's `mesh.voxelized(pitch)` (subdivide every face until its edges are <= pitch / 2, round
// every vertex of the subdivided mesh to the voxel lattice, keep the unique cells) and "inside" is trimesh's ray-parity
// `mesh.contains(points)`.  The reference runs it with trimesh on the CPU, one mesh at a time (minutes per test set).
//
// Here: three kernels over a batch of meshes, all decisions in EXACT arithmetic --
//  1. vox_origin_kernel   per (sample, part): lattice anchor of the part's bounding box (and the overflow check);
//  2. vox_mark_kernel     one wave per face: midpoint subdivision of a triangle gives, after n levels, the barycentric
//                         lattice (i a + j b + k c) / 2^n, and all four children of a face have the parent's edge lengths
//                         halved -- so the depth is uniform per face (smallest n with longest edge / 2^n <= pitch / 2) and
//                         the vertex set of trimesh's `subdivide_to_size` is that lattice.  Lanes walk the lattice points in
//                         float64 (exact: float32 inputs, small integer weights), round to the voxel cell (half to even, as
//                         numpy.round) and set the cell's bit in the part's occupancy mask (atomicOr: duplicates collapse,
//                         the result is order-independent);
//  3. vox_inside_kernel   per (sample, pair, slab of mask words): faces of part s in LDS; every set bit of part t is a point
//                         (cell index * pitch); +z ray parity against s with an exact float64 edge-function test and the
//                         top-left rule on shared edges (a point on a shared edge is counted for exactly one of the two
//                         triangles), so the parity is the geometric truth for a closed mesh; integer counts.
// Integer outputs: bit-exact against the numpy oracle (oracle/volume_ref.py) by construction.
#include "common.h"

namespace {

constexpr int MAX_PARTS = 32;

struct VolP {
    int B, V, n_parts, G;              // G: cells per axis of a part's occupancy grid (multiple of 32)
    double pitch, max_edge;
};

__device__ __forceinline__ int part_of_face(const int* __restrict__ part_first, int n_parts, int f) {
    int lo = 0, hi = n_parts - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (part_first[mid] <= f) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// ---- 1. lattice anchor per (sample, part): origin = floor(min / pitch) - 1 per axis; flags an extent that does not fit ----
__global__ __launch_bounds__(256) void vox_origin_kernel(const float* __restrict__ verts, const int* __restrict__ faces,
                                                         const int* __restrict__ part_first, VolP p, int* __restrict__ origin,
                                                         float* __restrict__ bbox, int* __restrict__ overflow) {
    const int b = blockIdx.x / p.n_parts, part = blockIdx.x % p.n_parts;
    const float* v = verts + (int64_t)b * p.V * 3;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int f = part_first[part] + threadIdx.x; f < part_first[part + 1]; f += 256)
        for (int c = 0; c < 3; ++c) {
            const int vi = faces[f * 3 + c];
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], v[vi * 3 + a]); hi[a] = fmaxf(hi[a], v[vi * 3 + a]); }
        }
    __shared__ float s_lo[3][256], s_hi[3][256];
    for (int a = 0; a < 3; ++a) { s_lo[a][threadIdx.x] = lo[a]; s_hi[a][threadIdx.x] = hi[a]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int a = 0; a < 3; ++a) {
                s_lo[a][threadIdx.x] = fminf(s_lo[a][threadIdx.x], s_lo[a][threadIdx.x + s]);
                s_hi[a][threadIdx.x] = fmaxf(s_hi[a][threadIdx.x], s_hi[a][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const bool empty = part_first[part + 1] == part_first[part];
        const double l = empty ? 0.0 : (double)s_lo[a][0], h = empty ? 0.0 : (double)s_hi[a][0];
        const int o = (int)floor(l / p.pitch) - 1;
        origin[blockIdx.x * 3 + a] = o;
        bbox[blockIdx.x * 6 + a] = (float)l;
        bbox[blockIdx.x * 6 + 3 + a] = (float)h;
        if ((int)ceil(h / p.pitch) + 1 - o >= p.G) atomicOr(overflow, 1);
    }
}

// ---- 2. occupancy: one wave per face --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vox_mark_kernel(const float* __restrict__ verts, const int* __restrict__ faces,
                                                       const int* __restrict__ part_first, const int* __restrict__ origin,
                                                       VolP p, int n_faces, uint32_t* __restrict__ mask,
                                                       int* __restrict__ overflow) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t wf = (int64_t)blockIdx.x * 4 + wave;
    if (wf >= (int64_t)p.B * n_faces) return;
    const int b = (int)(wf / n_faces), f = (int)(wf % n_faces);
    const int part = part_of_face(part_first, p.n_parts, f);
    const float* v = verts + (int64_t)b * p.V * 3;
    double a[3], bb[3], c[3];
    for (int k = 0; k < 3; ++k) {
        a[k] = (double)v[faces[f * 3 + 0] * 3 + k];
        bb[k] = (double)v[faces[f * 3 + 1] * 3 + k];
        c[k] = (double)v[faces[f * 3 + 2] * 3 + k];
    }
    auto len = [](const double* x, const double* y) {
        const double dx = y[0] - x[0], dy = y[1] - x[1], dz = y[2] - x[2];
        return sqrt(dx * dx + dy * dy + dz * dz);
    };
    const double longest = fmax(len(a, bb), fmax(len(bb, c), len(c, a)));
    int n = 0;
    while (longest / (double)(1 << n) > p.max_edge) {         // trimesh: subdivide while any edge is LONGER than max_edge
        if (++n > 10) { if (lane == 0) atomicOr(overflow, 2); return; }      // trimesh's max_iter = 10 ("max_iter exceeded")
    }
    const int N = 1 << n;
    const int total = (N + 1) * (N + 2) / 2;
    const int* org = origin + ((int64_t)b * p.n_parts + part) * 3;
    uint32_t* m = mask + ((int64_t)b * p.n_parts + part) * ((int64_t)p.G * p.G * p.G / 32);
    const double inv = 1.0 / (double)N;
    for (int idx = lane; idx < total; idx += 64) {
        // idx -> (i, j), 0 <= j <= N - i, rows of decreasing length
        int i = 0, rem = idx;
        while (rem > N - i) { rem -= N - i + 1; ++i; }
        const int j = rem, k = N - i - j;
        int cell[3];
        for (int ax = 0; ax < 3; ++ax) {
            const double q = ((double)i * a[ax] + (double)j * bb[ax] + (double)k * c[ax]) * inv;     // exact in float64
            cell[ax] = (int)rint(q / p.pitch) - org[ax];                                              // half to even, as numpy.round
        }
        if ((unsigned)cell[0] >= (unsigned)p.G || (unsigned)cell[1] >= (unsigned)p.G || (unsigned)cell[2] >= (unsigned)p.G) {
            atomicOr(overflow, 1);
            continue;
        }
        const int64_t bit = ((int64_t)cell[2] * p.G + cell[1]) * p.G + cell[0];
        atomicOr(m + (bit >> 5), 1u << (bit & 31));
    }
}

// ---- 3. voxels of part t inside part s ----------------------------------------------------------------------------
// +z ray from q: crossings with triangle (a, b, c) whose xy projection contains q (exact edge functions in float64,
// top-left rule on zero) and whose plane lies above q at that point.
__device__ __forceinline__ bool ray_hits_above(double qx, double qy, double qz, const float* t) {
    double ax = t[0], ay = t[1], az = t[2], bx = t[3], by = t[4], bz = t[5], cx = t[6], cy = t[7], cz = t[8];
    double area = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
    if (area == 0.0) return false;                                 // edge-on: no transversal crossing
    if (area < 0.0) { double s; s = bx; bx = cx; cx = s; s = by; by = cy; cy = s; s = bz; bz = cz; cz = s; area = -area; }
    auto edge = [&](double ux, double uy, double vx, double vy, bool& ok) {
        const double dx = vx - ux, dy = vy - uy;
        const double e = dx * (qy - uy) - dy * (qx - ux);           // both products exact, the difference correctly rounded
        ok = e > 0.0 || (e == 0.0 && (dy > 0.0 || (dy == 0.0 && dx < 0.0)));
        return e;
    };
    bool o0, o1, o2;
    const double e_bc = edge(bx, by, cx, cy, o0);
    if (!o0) return false;
    const double e_ca = edge(cx, cy, ax, ay, o1);
    if (!o1) return false;
    const double e_ab = edge(ax, ay, bx, by, o2);
    if (!o2) return false;
    const double z = (e_bc * az + e_ca * bz + e_ab * cz) / area;
    return z > qz;
}

__global__ __launch_bounds__(256) void vox_inside_kernel(const float* __restrict__ verts, const int* __restrict__ faces,
                                                         const int* __restrict__ part_first, const int* __restrict__ origin,
                                                         const float* __restrict__ bbox, const uint32_t* __restrict__ mask,
                                                         const int* __restrict__ pairs, int n_pairs, int slabs, VolP p,
                                                         int max_faces, unsigned long long* __restrict__ count,
                                                         int* __restrict__ pair_count) {
    extern __shared__ float s_tri[];                               // part s: faces x 9 floats
    int blk = blockIdx.x;
    const int slab = blk % slabs; blk /= slabs;
    const int pr = blk % n_pairs; const int b = blk / n_pairs;
    const int s = pairs[pr * 2], t = pairs[pr * 2 + 1];
    const float* bs = bbox + ((int64_t)b * p.n_parts + s) * 6;
    const float* bt = bbox + ((int64_t)b * p.n_parts + t) * 6;
    // the voxel centres of t lie within pitch / 2 of t's surface: boxes further apart than that cannot contribute
    const float slack = (float)p.pitch;
    for (int a = 0; a < 3; ++a)
        if (bt[a] - slack > bs[3 + a] || bt[3 + a] + slack < bs[a]) return;
    const int f0 = part_first[s], nf = part_first[s + 1] - f0;
    const float* v = verts + (int64_t)b * p.V * 3;
    for (int i = threadIdx.x; i < nf * 9; i += 256) {
        const int f = i / 9, r = i % 9;
        s_tri[i] = v[faces[(f0 + f) * 3 + r / 3] * 3 + r % 3];
    }
    __syncthreads();
    const int64_t words = (int64_t)p.G * p.G * p.G / 32;
    const uint32_t* m = mask + ((int64_t)b * p.n_parts + t) * words;
    const int* org = origin + ((int64_t)b * p.n_parts + t) * 3;
    const int64_t per = (words + slabs - 1) / slabs;
    const int64_t w0 = slab * per, w1 = min(words, w0 + per);
    int local = 0;
    for (int64_t w = w0 + threadIdx.x; w < w1; w += 256) {
        uint32_t bits = m[w];
        while (bits) {
            const int bitpos = __ffs(bits) - 1;
            bits &= bits - 1;
            const int64_t bit = w * 32 + bitpos;
            const int cx = (int)(bit % p.G), cy = (int)((bit / p.G) % p.G), cz = (int)(bit / ((int64_t)p.G * p.G));
            const double qx = (double)(cx + org[0]) * p.pitch, qy = (double)(cy + org[1]) * p.pitch, qz = (double)(cz + org[2]) * p.pitch;
            if (qx < bs[0] || qx > bs[3] || qy < bs[1] || qy > bs[4] || qz < bs[2] || qz > bs[5]) continue;
            int crossings = 0;
            for (int f = 0; f < nf; ++f) crossings += ray_hits_above(qx, qy, qz, s_tri + f * 9) ? 1 : 0;
            local += crossings & 1;
        }
    }
    // workgroup total (integers: any order gives the same sum)
    __shared__ int s_sum[256];
    s_sum[threadIdx.x] = local;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) s_sum[threadIdx.x] += s_sum[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_sum[0]) {
        atomicAdd(count + b, (unsigned long long)s_sum[0]);
        if (pair_count) atomicAdd(pair_count + (int64_t)b * n_pairs + pr, s_sum[0]);
    }
}

}  // namespace

extern "C" {

int64_t dsf_part_volume_workspace_bytes(int B, int n_parts, int grid) {
    if (B < 0 || n_parts <= 0 || grid <= 0 || (grid & 31)) return -1;
    const int64_t words = (int64_t)grid * grid * grid / 32;
    // [overflow flag + pad 16 B][origin B*P*3 int][bbox B*P*6 float][masks B*P*words u32]
    return 16 + (int64_t)B * n_parts * (3 * 4 + 6 * 4) + (int64_t)B * n_parts * words * 4;
}

int dsf_part_intersection_volume(const float* verts, const int32_t* faces, const int32_t* part_first, const int32_t* pairs,
                                 int B, int V, int n_parts, int n_faces, int n_pairs, int max_part_faces, double pitch, int grid,
                                 void* workspace, unsigned long long* count, int32_t* pair_count, int32_t* status,
                                 dsf_stream_t stream) {
    DSF_CHECK_ARG(verts && faces && part_first && pairs && workspace && count && status);
    DSF_CHECK_ARG(B >= 0 && V > 0 && n_parts > 0 && n_parts <= MAX_PARTS && n_faces > 0 && n_pairs >= 0 && pitch > 0.0);
    DSF_CHECK_ARG(grid >= 32 && (grid & 31) == 0 && grid <= 512 && max_part_faces > 0 && max_part_faces * 36 <= 64 * 1024);
    hipStream_t st = (hipStream_t)stream;
    if (dsf_zero_async(count, sizeof(unsigned long long) * (size_t)(B > 0 ? B : 1), st) != hipSuccess) return DSF_ERR_LAUNCH;
    if (pair_count && B * n_pairs > 0 &&
        dsf_zero_async(pair_count, sizeof(int32_t) * (size_t)B * n_pairs, st) != hipSuccess) return DSF_ERR_LAUNCH;
    if (dsf_zero_async(status, sizeof(int32_t), st) != hipSuccess) return DSF_ERR_LAUNCH;
    if (B == 0 || n_pairs == 0) return DSF_OK;
    VolP p = {B, V, n_parts, grid, pitch, pitch / 2.0};           // trimesh voxelize_subdivide: edge_factor = 2
    char* ws = (char*)workspace;
    int* origin = (int*)(ws + 16);
    float* bbox = (float*)(origin + (int64_t)B * n_parts * 3);
    uint32_t* mask = (uint32_t*)(bbox + (int64_t)B * n_parts * 6);
    const int64_t words = (int64_t)grid * grid * grid / 32;
    if (dsf_zero_async(mask, (size_t)B * n_parts * words * 4, st) != hipSuccess) return DSF_ERR_LAUNCH;
    hipLaunchKernelGGL(vox_origin_kernel, dim3(B * n_parts), dim3(256), 0, st, verts, faces, part_first, p, origin, bbox, status);
    const int64_t waves = (int64_t)B * n_faces;
    hipLaunchKernelGGL(vox_mark_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, verts, faces, part_first, origin, p,
                       n_faces, mask, status);
    const int slabs = 8;
    hipLaunchKernelGGL(vox_inside_kernel, dim3((unsigned)(B * n_pairs * slabs)), dim3(256), (size_t)max_part_faces * 36, st, verts,
                       faces, part_first, origin, bbox, mask, pairs, n_pairs, slabs, p, max_part_faces, count, pair_count);
    return dsf_launch_status();
}

}  // extern "C"
