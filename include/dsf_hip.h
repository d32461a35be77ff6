/*
 * dsf_hip.h -- C ABI of libdsf_hip.so, the MI355X (gfx950) native layer of the
 * DSF training hot path.  This is the drop-in boundary: plain pointers (device
 * memory unless stated), sizes and a HIP stream; no torch types.  Every entry
 * point returns an int status (DSF_OK = 0); kernels never allocate, free or
 * synchronise, so every call is safe under HIP-graph capture.  All tensors are
 * contiguous row-major, fp32 unless noted.
 *
 * Each entry point cites the reference interface it replaces
 * (paths relative to the reference repository root).
 */
#ifndef DSF_HIP_H
#define DSF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSF_OK 0
#define DSF_ERR_INVALID_ARG 1
#define DSF_ERR_UNSUPPORTED 2
#define DSF_ERR_LAUNCH 3

#define DSF_MANO_VERTS 779      /* 778 + wrist cap vertex (render_model/mano_layer.py:636-637) */
#define DSF_MANO_JOINTS 21
#define DSF_MANO_FACES 1554
#define DSF_MANO_SAVE_FLOATS 5248   /* per-sample forward state: staging between the two forward kernels, kept for the backward pass */
#define DSF_MANO_BWD_SCRATCH_FLOATS 2560   /* per-sample staging between the two backward kernels */
#define DSF_N_SPHERES 66

typedef void* dsf_stream_t;     /* hipStream_t */

int dsf_abi_version(void);
const char* dsf_status_string(int status);

/* ------------------------------------------------------------------------------------
 * K5  MANO layer: shape/pose blendshapes, Rodrigues, kinematic chain, LBS, joint
 * regression, wrist cap, output affine.
 * Replaces MANO_SMPL.forward (render_model/mano_layer.py:573-641) and
 * MANO_SMPL.get_mano_vertices (:643-693) -- ~40 torch kernels + a 15-step Python loop.
 * ---------------------------------------------------------------------------------- */
typedef struct dsf_mano_model {
    const float* v_template;    /* (778,3)                                   mano_layer.py:112-113 */
    const float* shapedirs;     /* (10, 2334)                                :116-120 */
    const float* posedirs;      /* (135, 2334)                               :142-145 */
    const float* j_regressor;   /* (778, 21) dense, 16 regressed + 5 tips    :123-132 */
    const float* j_template;    /* (16,3)   = J_regressor[:, :16]^T v_template   (host precompute) */
    const float* j_shapedirs;   /* (10, 48) = J_regressor[:, :16]^T shapedirs_k  (host precompute) */
    const float* hands_comp;    /* (45,45)                                   :135-136 */
    const float* hands_mean;    /* (45)                                      :138-139 */
    const float* weights;       /* (778,16)                                  :149-154 */
    const int32_t* parents;     /* (16), parents[0] = -1                     :147 */
    const int32_t* wrist_ring;  /* (16) vertex ids averaged into vertex 778  :636 */
    const int32_t* jreg_rowptr; /* (22) CSR of j_regressor^T (joint-major) */
    const int32_t* jreg_col;    /* (nnz) vertex ids */
    const float* jreg_val;      /* (nnz) */
} dsf_mano_model;

/* rot: (B,rot_dim) rot_dim 3 = axis-angle, 4 = quaternion(w,x,y,z); theta: (B,ncomp) PCA
 * coefficients, ncomp <= 45; beta: (B,10); cam: (B,4) = scale|trans or NULL.
 * out = ((raw * k1) * k2) * cam[0] + cam[1:4]   (k1 = 1000, k2 = global_scale in get_mano_vertices;
 * k1 = k2 = 1 and cam = NULL reproduce MANO_SMPL.forward).
 * verts (B,779,3), joints (B,21,3), Rs (B,15,3,3) (may be NULL), save (B,DSF_MANO_SAVE_FLOATS): REQUIRED since ABI 2 --
 * the blendshape launch (8 samples x 256 columns per workgroup) hands v_posed and the pose state to the per-sample
 * skinning launch through it, and dsf_mano_backward reads it.
 * param_stride: floats between consecutive samples of beta / theta / rot / cam; 0 = each array tightly packed.  With
 * param_stride = 62 the four pointers can be column offsets into the network's (B,62) output row
 * [rot 3 | theta 45 | beta 10 | cam 4] (Render._split, mano_layer.py:1071-1076): no slicing copies. */
int dsf_mano_forward(const dsf_mano_model* m, const float* beta, const float* theta, const float* rot,
                     const float* cam, int B, int ncomp, int rot_dim, int param_stride, float k1, float k2,
                     float* verts, float* joints, float* Rs, float* save, dsf_stream_t stream);

/* grad_verts (B,779,3) / grad_joints (B,21,3): either may be NULL (= zeros).
 * Outputs (all written, not accumulated): grad_beta (B,10), grad_theta (B,ncomp),
 * grad_rot (B,rot_dim), grad_cam (B,4) or NULL; param_stride applies to the inputs and to the four gradients
 * (column offsets into one (B,62) gradient row).
 * scratch (B,DSF_MANO_BWD_SCRATCH_FLOATS), since ABI 2: staging between the per-sample skinning / chain launch and the
 * blendshape-reduction launch; contents undefined before and after the call, `save` is left untouched (a retained graph
 * can be differentiated again). */
int dsf_mano_backward(const dsf_mano_model* m, const float* theta, const float* rot, const float* cam,
                      const float* save, const float* grad_verts, const float* grad_joints,
                      int B, int ncomp, int rot_dim, int param_stride, float k1, float k2,
                      float* grad_beta, float* grad_theta, float* grad_rot, float* grad_cam,
                      float* scratch, dsf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K1/K2  Mesh rasteriser (pytorch3d==0.4.0 semantics).
 * Replaces pytorch3d._C.rasterize_meshes / rasterize_meshes_backward as driven by
 * MeshRasterizer (render_model/mano_layer.py:946-952, call sites :1022,1054,1083,1117,
 * 1194,1211) for blur_radius=0, faces_per_pixel=1, perspective_correct=False,
 * clip_barycentric_coords=False, cull_backfaces=False (anything else: DSF_ERR_UNSUPPORTED).
 * ---------------------------------------------------------------------------------- */
typedef struct dsf_camera {
    float fx, fy, px, py;       /* screen-space intrinsics (mano_layer.py:939-945) */
    float img_w, img_h;         /* PerspectiveCameras image_size = (640, 480) */
} dsf_camera;

/* World verts (N,V,3) -> face_verts (N*F,3,3) of (x_ndc, y_ndc, z_view): camera R=diag(-1,-1,1),
 * T=0, then Meshes packing verts_packed[faces_packed] (faces (F,3) int32 shared by all meshes). */
int dsf_project_face_verts(const float* verts, const int32_t* faces, const dsf_camera* cam,
                           int N, int V, int F, float* face_verts, dsf_stream_t stream);

/* face_verts (F_total,3,3); mesh_to_face_first_idx / num_faces_per_mesh (N) int64 (device).
 * Outputs (N,S,S[,3]): pix_to_face int64 (packed face index or -1), zbuf, bary (may be NULL),
 * dists (may be NULL).  Exact-z ties resolve to the lowest face index (= the naive path).
 * workspace: (N,4) floats of scratch for the per-mesh screen bbox (tiles outside it skip the
 * face scan), or NULL to disable that culling. */
int dsf_rasterize_meshes(const float* face_verts, const int64_t* mesh_to_face_first_idx,
                         const int64_t* num_faces_per_mesh, int N, int64_t F_total, int image_size,
                         float blur_radius, int faces_per_pixel, int perspective_correct,
                         int clip_barycentric_coords, int cull_backfaces,
                         int64_t* pix_to_face, float* zbuf, float* bary, float* dists,
                         float* workspace, dsf_stream_t stream);

/* grad_bary / grad_dists may be NULL (= zeros; the reference only consumes zbuf).
 * grad_face_verts (F_total,3,3) is zeroed by this call, then accumulated with float atomics. */
int dsf_rasterize_meshes_backward(const float* face_verts, const int64_t* pix_to_face,
                                  const float* grad_zbuf, const float* grad_bary, const float* grad_dists,
                                  int N, int64_t F_total, int image_size, float* grad_face_verts,
                                  dsf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K1+K8 fused "crop mode": the whole leaf of Render.render / normal_render / mesh2img /
 * getDepth (render_model/mano_layer.py:1078-1092, 1190-1218):
 *   world verts -> raster (640x640, Appendix A) -> background -> 0 (:1085) -> resize to
 *   480x640 (:1233-1242) -> nearest crop warp with torch.inverse(M) (:1244-1260) ->
 *   normalize_img (:1289-1299),
 * evaluated only at the <=128*128 raster pixels the crop reads (bit-identical to the full
 * chain, ~25x less work, no 11.5 MB/mesh fragment traffic).
 * ---------------------------------------------------------------------------------- */
/* center3d (B,3) mm, cube (B,3) mm -> center2d (B,3) (points3DToImg :1318-1324), M (B,3,3)
 * (comToBounds :1133-1141 + Offset2Trans :1143-1169), bounds (B,4) int32 xs,xe,ys,ye (may be NULL),
 * minv_closed (B,3,3) closed-form inverse of the affine M (may be NULL). */
int dsf_crop_setup(const float* center3d, const float* cube, const dsf_camera* cam, int B, int crop,
                   float* center2d, float* M, int32_t* bounds, float* minv_closed, dsf_stream_t stream);

/* verts (B,V,3) world mm; faces (F,3) int32; minv (B,3,3) = the reference's torch.inverse(M);
 * resize_rowmap (img_h) int32: source raster row for each row of the resized image (derived
 * from torch's own affine_grid+grid_sample at init); center_z, cube_z (B) for
 * normalize_img (pass NULL for both to get metric depth with 0 background).
 * Outputs: img (B,1,crop,crop); pix_to_face (B,crop,crop) int32 local face id or -1 (may be NULL). */
int dsf_render_crop_forward(const float* verts, const int32_t* faces, const float* minv,
                            const int32_t* resize_rowmap, const float* center_z, const float* cube_z,
                            const dsf_camera* cam, int B, int V, int F, int raster_size, int crop,
                            float* img, int32_t* pix_to_face, dsf_stream_t stream);

/* grad_img (B,1,crop,crop) -> grad_verts (B,V,3) world (zeroed, then accumulated).
 * Gradient flows only through zbuf of covered, un-clamped pixels (Appendix A.3). */
int dsf_render_crop_backward(const float* verts, const int32_t* faces, const float* minv,
                             const int32_t* resize_rowmap, const float* center_z, const float* cube_z,
                             const dsf_camera* cam, const int32_t* pix_to_face,
                             const float* grad_img, int B, int V, int F, int raster_size, int crop,
                             float* grad_verts, dsf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K3/K4  Point -> triangle squared distance.
 * Replaces pytorch3d._C.point_face_dist_forward / _backward (metric/meshLoss.py:52, 63).
 * ---------------------------------------------------------------------------------- */
/* points (P,3); tris (T,3,3); *_first_idx (N) int64 CSR starts; dists (P), idxs (P) int64
 * (packed triangle index, ties -> lowest). max_points is accepted for signature parity. */
int dsf_point_face_dist_forward(const float* points, const int64_t* points_first_idx, const float* tris,
                                const int64_t* tris_first_idx, int N, int64_t P, int64_t T,
                                int64_t max_points, float* dists, int64_t* idxs, dsf_stream_t stream);

/* grad_points (P,3) and grad_tris (T,3,3) are zeroed by this call. */
int dsf_point_face_dist_backward(const float* points, const float* tris, const int64_t* idxs,
                                 const float* grad_dists, int64_t P, int64_t T, float* grad_points,
                                 float* grad_tris, dsf_stream_t stream);

/* Fused batched variant used by ICPLoss / JointICPLoss / FingerICPLoss (metric/meshLoss.py:347-395):
 * verts (B,V,3), points (B,P,3), faces (Fcat,3) int32 = concatenation of n_parts face lists with
 * part_first (n_parts+1) int32 offsets; seg (B,P) int64 labels or NULL.
 *   seg == NULL (ICPLoss, n_parts = 1): every point is tested against part 0.
 *   seg != NULL: point p is tested against part seg[p]-1 only (label 0 -> no part, dist 0);
 *   this is exactly the subset the reference keeps after its 15x replicated launch.
 * Outputs: dists (B,P), idxs (B,P) int32 = index into the concatenated face list (or -1). */
int dsf_mesh_point_dist_forward(const float* verts, const float* points, const int32_t* faces,
                                const int32_t* part_first, const int64_t* seg, int B, int V, int P,
                                int n_parts, float* dists, int32_t* idxs, dsf_stream_t stream);

/* grad_verts (B,V,3) zeroed then accumulated; grad_points (B,P,3) written (may be NULL). */
int dsf_mesh_point_dist_backward(const float* verts, const float* points, const int32_t* faces,
                                 const int32_t* idxs, const float* grad_dists, int B, int V, int P,
                                 float* grad_verts, float* grad_points, dsf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K6/K7  Sphere hand model: radii/centres, collision loss, point-cloud part labels.
 * Replaces MANO_SMPL.get_sphere_radius (:271-317), calculate_coll (:373-386), seg_pcl (:404-426).
 * ---------------------------------------------------------------------------------- */
typedef struct dsf_sphere_model {
    const uint8_t* jreg_mask;   /* (21,778) 1 where J_regressor[v][j] > 0   (:275) */
    const float* coll_mask;     /* (66,66)                                   (:240-269) */
    float t_finger[3];          /* linspace(0,1,4)[:-1]                      (:231) */
    float t_palm[4];            /* linspace(0,1,6)[1:-1]                     (:236) */
} dsf_sphere_model;

/* joints (B,21,3), mesh (B,V>=778,3) -> centres (B,66,3), radii (B,66); topk_idx (B,21,10) int32
 * (the 10 nearest owned vertices per joint, needed by the backward; may be NULL). */
int dsf_sphere_set(const dsf_sphere_model* sm, const float* joints, const float* mesh, int B, int V,
                   float* centres, float* radii, int32_t* topk_idx, dsf_stream_t stream);
/* seg_pcl's sphere set, /root/reference/render_model/mano_layer.py:404-413: centres from `joints_centres` (the pixel branch's
 * skeleton), radii from `joints_radii` + mesh (the MANO skeleton) -- the reference calls get_sphere_radius twice and keeps half of
 * each result; here one launch (one top-10 selection per joint). */
int dsf_sphere_mixed(const dsf_sphere_model* sm, const float* joints_centres, const float* joints_radii, const float* mesh, int B,
                     int V, float* centres, float* radii, dsf_stream_t stream);

/* loss_rows (B,66): gated row sums sum_j err_ij (the caller takes the mean, = calculate_coll).
 * Also writes centres/radii/topk_idx for the backward. */
int dsf_collision_forward(const dsf_sphere_model* sm, const float* joints, const float* mesh, int B, int V,
                          float* loss_rows, float* centres, float* radii, int32_t* topk_idx,
                          dsf_stream_t stream);

/* grad_rows (B,66) -> grad_joints (B,21,3), grad_mesh (B,V,3) (both written). */
int dsf_collision_backward(const dsf_sphere_model* sm, const float* joints, const float* mesh,
                           const float* centres, const float* radii, const int32_t* topk_idx,
                           const float* grad_rows, int B, int V, float* grad_joints, float* grad_mesh,
                           dsf_stream_t stream);

/* centres (B,66,3) from the pixel-branch joints, radii (B,66) from the MANO joints (:407-408);
 * pcl (B,P,3) -> labels (B,P) int64 in 0..15. */
int dsf_seg_pcl(const float* centres, const float* radii, const float* pcl, int B, int P,
                int64_t* labels, dsf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K8/K9/K10  image-side elementwise kernels.
 * ---------------------------------------------------------------------------------- */
/* uvd (B,N,3) crop-normalised -> xyz (B,N,3); mat (B,3,3) = torch.inverse(M).
 * normalise=1: uvd_nl2xyznl_tensor, 0: uvd_nl2xyz_tensor (data/render_loader.py:1044-1073). */
int dsf_uvd_to_xyz(const float* uvd, const float* center, const float* minv, const float* cube,
                   const dsf_camera* cam, int B, int N, int img_size, int normalise, float* xyz,
                   dsf_stream_t stream);
/* xyz_nl2uvdnl_tensor (data/render_loader.py:1075-1088) == Render.JointTrans for world input
 * (render_model/mano_layer.py:1301-1309) when world=1 (xyz already in mm). */
int dsf_xyz_to_uvd(const float* xyz, const float* center, const float* M, const float* cube,
                   const dsf_camera* cam, int B, int N, int img_size, int world, float* uvd,
                   dsf_stream_t stream);
/* backward of the two transforms w.r.t. their point input */
int dsf_uvd_to_xyz_backward(const float* uvd, const float* center, const float* minv, const float* cube,
                            const dsf_camera* cam, const float* grad_xyz, int B, int N, int img_size,
                            int normalise, float* grad_uvd, dsf_stream_t stream);
int dsf_xyz_to_uvd_backward(const float* xyz, const float* center, const float* M, const float* cube,
                            const dsf_camera* cam, const float* grad_uvd, int B, int N, int img_size,
                            int world, float* grad_xyz, dsf_stream_t stream);

/* crop_hand (data/render_loader.py:1209-1227) fused with uvdImg2xyzImg (:1190-1201):
 * img (B,1,S,S), joints_nl (B,J,3) -> img_hand (B,1,S,S); xyz_nl (B,S*S,3) normalised point
 * image (may be NULL); keep (B,S*S) uint8 inside-box mask (may be NULL; d img_hand/d img). */
int dsf_crop_hand(const float* img, const float* joints_nl, const float* center, const float* minv,
                  const float* cube, const dsf_camera* cam, int B, int J, int S, float offset_xy,
                  float offset_z, float thickness, float* img_hand, float* xyz_nl, uint8_t* keep,
                  dsf_stream_t stream);

/* Img2pcl (data/render_loader.py:1121-1156): valid = img <= 0.99 in scan order, converted with
 * uvd_nl2xyznl; resampled to exactly n_sample points: floor(n_sample/count) whole copies followed
 * by (n_sample mod count) distinct points (count >= n_sample: n_sample distinct points), zeros if
 * empty.  The reference draws with torch.multinomial; here the draw is the explicit input
 * rand_keys (B,S*S) uint32 (the distinct points are the valid pixels with the smallest keys --
 * ties by scan order -- emitted in scan order), so tests can inject the same draw.
 * workspace: (B, 2*S*S) uint32.  counts (B) int32 out. */
int dsf_img2pcl(const float* img, const float* center, const float* minv, const float* cube,
                const dsf_camera* cam, const uint32_t* rand_keys, int B, int S, int n_sample,
                float* pcl, int32_t* counts, uint32_t* workspace, dsf_stream_t stream);

/* SmoothL1Loss with delta (metric/losses.py:6-30): loss[0] = scale * sum_i h(x_i - y_i),
 * h(z) = 0.5 z^2 if |z| < delta else delta (|z| - delta/2); scale = 1/n for size_average (mean over the last
 * dim, then mean over the rest), 1/last_dim for the sum variant.  x, y: n floats walked in memory order (any two
 * tensors with identical dense strides).  workspace: >= 1024 floats (may be NULL when n <= 4096).
 * Deterministic (fixed partials, fixed order).  backward: grad_x = grad_loss[0] * scale * h'(x - y). */
int dsf_huber_mean_forward(const float* x, const float* y, int64_t n, float delta, float scale, float* loss,
                           float* workspace, dsf_stream_t stream);
int dsf_huber_mean_backward(const float* x, const float* y, const float* grad_loss, int64_t n, float delta,
                            float scale, float* grad_x, dsf_stream_t stream);

/* GFM.joint2offset (util/generateFeature.py:14-37 == model/backbone.py:68-91):
 * joints (B,J,3), img (B,1,H,H) -> maps (B,4J,S,S).
 * map_strides: host array {batch, channel, pixel} strides in floats of maps / grad_maps (element (b,c,y,x) at
 * b*s0 + c*s1 + (y*S + x)*s2), NULL = contiguous NCHW {4J*S*S, S*S, 1}; channels-last is {4J*S*S, 1, 4J}, a
 * channel slice of a wider channels-last tensor keeps that tensor's batch / pixel strides. */
int dsf_joint2offset_forward(const float* joints, const float* img, int B, int J, int H, int S,
                             float kernel_size, float* maps, const int64_t* map_strides, dsf_stream_t stream);
int dsf_joint2offset_backward(const float* joints, const float* img, const float* grad_maps, int B, int J,
                              int H, int S, float kernel_size, float* grad_joints, const int64_t* map_strides,
                              dsf_stream_t stream);
/* GFM.offset2joint_softmax (util/generateFeature.py:39-59 == model/backbone.py:45-65):
 * maps (B,4J,S,S), depth (B,1,H,H) -> joints (B,J,3); stats (B,J,2) = softmax max / denominator
 * kept for the backward. */
int dsf_offset2joint_forward(const float* maps, const float* depth, int B, int J, int H, int S,
                             float kernel_size, float scale, float* joints, float* stats,
                             dsf_stream_t stream);
int dsf_offset2joint_backward(const float* maps, const float* depth, const float* joints,
                              const float* stats, const float* grad_joints, int B, int J, int H, int S,
                              float kernel_size, float scale, float* grad_maps, dsf_stream_t stream);
/* The same pair on maps of ANY uniformly strided layout: map_strides = {batch, channel, pixel} strides in floats (NCHW:
 * {4 J S S, S S, 1}; channels-last, the layout the network produces: {4 J S S, 1, 4 J}); grad_maps is written in the maps' own
 * layout.  Round 6: the decode of a channels-last map needs no layout copy in front of it and none behind its backward. */
int dsf_offset2joint_forward_strided(const float* maps, const int64_t* map_strides, const float* depth, int B, int J, int H, int S,
                                     float kernel_size, float scale, float* joints, float* stats, dsf_stream_t stream);
int dsf_offset2joint_backward_strided(const float* maps, const int64_t* map_strides, const float* depth, const float* joints,
                                      const float* stats, const float* grad_joints, int B, int J, int H, int S, float kernel_size,
                                      float scale, float* grad_maps, dsf_stream_t stream);
/* The same decode for a DENSE CHANNELS-LAST map (B, S*S, 4J) -- the layout the network's heads write -- with J <= 32 (else
 * DSF_ERR_UNSUPPORTED: use the strided entry points): workgroups own 128 consecutive pixels of a sample and all joints, so that a
 * pixel's 4J-float record is read contiguously (the (sample, joint) kernels behind the strided entry points touch a cache line per
 * lane on this layout: 84 / 125 us forward / backward at B = 32 against 21 / 23 us on NCHW).  Forward = three small launches
 * (per-chunk maxima, per-chunk sums with the global maximum, an ordered fold: deterministic); backward = one elementwise launch.
 * `workspace`: dsf_offset2joint_cl_workspace_floats(B, S) floats, uninitialised.  Per element the arithmetic is that of
 * dsf_offset2joint_forward / _backward; the forward differs from it by the order of the fp32 sums only.  Added in round 6 (ABI 4). */
int64_t dsf_offset2joint_cl_workspace_floats(int B, int S);
int dsf_offset2joint_forward_cl(const float* maps, const float* depth, int B, int J, int H, int S, float kernel_size, float scale,
                                float* joints, float* stats, float* workspace, dsf_stream_t stream);
int dsf_offset2joint_backward_cl(const float* maps, const float* depth, const float* joints, const float* stats,
                                 const float* grad_joints, int B, int J, int H, int S, float kernel_size, float scale,
                                 float* grad_maps, dsf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K11  fp32 implicit-GEMM convolution on the matrix cores (v_mfma_f32_32x32x2_f32).
 * Replaces the cuDNN/MIOpen convolutions behind nn.Conv2d / nn.ConvTranspose2d of
 * model/resnet.py, model/backbone.py, model/hourglass.py, render_model/transfer.py (MIOpen has
 * no gfx950 database in ROCm 7.2: find mode compiles for hours, immediate mode runs naive kernels).
 * NHWC activations; W is the [KH*KW*Ci][Co] row-major GEMM operand.
 *   Y[b,oy,ox,n] = bias[n] + sum_{kh,kw,c} Xv[b, oy*stride+kh-pad_h, ox*stride+kw-pad_w, c] * W[(kh*KW+kw)*Ci+c][n]
 * where Xv is X upsampled by `dil` with zeros (dil = 1: ordinary convolution; dil > 1 with
 * stride = 1: transposed convolution / backward-data of a strided convolution).
 * ---------------------------------------------------------------------------------- */
int dsf_conv_igemm_forward(const float* X, const float* W, const float* bias, float* Y, int B, int Hi, int Wi,
                           int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int dil, int pad_h,
                           int pad_w, dsf_stream_t stream);
/* Backward-data of a stride-1 convolution read straight from the layer's forward operand W_fwd
 * [KH][KW][Cin][Cout] (taps flipped and the tile transposed in the loader; no re-laid weight copy):
 * dX[b,y,x,ci] = sum_{kh,kw,co} dY[b, y+pad_h-kh, x+pad_w-kw, co] * W_fwd[kh][kw][ci][co];
 * dY is (B, H+2pad-KH+1, W+2pad-KW+1, Cout), dX (B,H,W,Cin).  Needs Cin%4 == Cout%4 == 0, Cout >= 32
 * (else DSF_ERR_UNSUPPORTED: use dsf_conv_igemm_forward with flipped weights). */
int dsf_conv_igemm_bwd_data_s1(const float* dY, const float* W_fwd, float* dX, int B, int H, int W, int Cout,
                               int Cin, int KH, int KW, int pad_h, int pad_w, dsf_stream_t stream);
/* dsf_conv_igemm_forward with the weight operand given transposed and tap-flipped, Wt[KH][KW][Co][Ci] holding tap
 * (kh, kw) at (KH-1-kh, KW-1-kw): the memory of a layer's own parameter seen from its other direction
 * (ConvTranspose2d forward, backward-data of strided Conv2d).  Implemented by the vectorised kernels only
 * (Ci, Co multiples of 4, Ci >= 32; dil 2 needs even Ho, Wo); returns DSF_ERR_UNSUPPORTED otherwise. */
int dsf_conv_igemm_forward_wt(const float* X, const float* Wt, const float* bias, float* Y, int B, int Hi, int Wi,
                              int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int dil, int pad_h,
                              int pad_w, dsf_stream_t stream);

/* dW[(kh*KW+kw)*Ci+c][n] = sum_{b,oy,ox} X[b, oy*stride+kh-pad_h, ox*stride+kw-pad_w, c] * dY[b,oy,ox,n]
 * (accumulated with float atomics across the pixel splits).  accumulate = 0: dW is zeroed by the call;
 * accumulate != 0: the sums are added to what dW holds (a caller that zeroes one pool for all layers, or
 * gradient accumulation). */
int dsf_conv_igemm_wrw(const float* X, const float* dY, float* dW, int B, int Hi, int Wi, int Ci, int Ho, int Wo,
                       int Co, int KH, int KW, int stride, int pad_h, int pad_w, int accumulate, dsf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Depth data path (csrc/data_ops.hip): the test-phase crop of loader.__getitem__
 * (data/render_loader.py:1909-1916) = Crop_Image_deep_pp (:748-810; comToBounds :356-364, getCrop :867-905,
 * cv2.resize INTER_NEAREST) + normalize_img (:738-745) on raw depth frames in HBM.
 * depth (B,Hd,Wd) f32 mm (0 = hole), com (B,3) f64 = centre of mass (u, v, z mm), cube (B,3) f64 mm, fx / fy the focal
 * lengths -> img (B,dsize,dsize) f32 in [-1,1], trans (B,3,3) f64 (may be NULL) = the crop's pixel transform,
 * raw_crop (B,dsize,dsize) f32 (may be NULL) = the crop before normalisation (what the training-phase augmentations
 * take).  Source-pixel selection is bit-exact with the numpy code; normalised values agree to 1 float32 ulp. */
int dsf_depth_crop_normalize(const float* depth, const double* com, const double* cube, double fx, double fy, int B, int Hd,
                             int Wd, int dsize, float* img, double* trans, float* raw_crop, dsf_stream_t stream);
/* the same on the sensors' RAW 16-bit frames (uint16 millimetres: what nyu_reader / icvl_reader / hands17_reader decode,
 * render_loader.py:201-218, before their float32 cast) -- bit-identical results, half the frame traffic, no host cast */
int dsf_depth_crop_normalize_u16(const uint16_t* depth, const double* com, const double* cube, double fx, double fy, int B, int Hd,
                                 int Wd, int dsize, float* img, double* trans, float* raw_crop, dsf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Direct convolution for 1-channel inputs (csrc/conv_c1.hip): the 5x5 ResNet stem (model/backbone.py:196-199), the 7x7
 * stride-2 hourglass stem (model/hourglass.py:178) and the generator's first layer (render_model/transfer.py:409).
 * X (B,Hi,Wi) f32, W [K][K][1][Co] (= [K*K][Co]), Y / dY (B,Ho,Wo,Co) NHWC, Co <= 64, square K in {5, 7}, stride 1 or 2,
 * symmetric zero padding `pad`; other shapes return DSF_ERR_UNSUPPORTED (dsf_conv_c1_supported tells beforehand) and go
 * through dsf_conv_igemm_*.  fp32 FMAs, lane = output channel.  dsf_conv_c1_wrw: dW [K*K][Co] (accumulate as in
 * dsf_conv_igemm_wrw), workspace = dsf_conv_c1_workspace_bytes(K, K) bytes of per-workgroup partial sums; deterministic.
 * ---------------------------------------------------------------------------------- */
int dsf_conv_c1_supported(int Co, int KH, int KW, int stride);
/* ONE output channel (the generator's last layer, ReflectionPad2d(3) + Conv2d(64, 1, 7): render_model/transfer.py:441-442; as an
 * implicit GEMM it would pad N from 1 to 64 columns): X (B,Hi,Wi,Ci) NHWC, W [K][K][Ci] (the kernel layout [K][K][Ci][1]),
 * Y (B,Ho,Wo); square K in {3, 5, 7}, stride 1, Ci % 8 == 0, zero padding `pad`; other shapes DSF_ERR_UNSUPPORTED.  fp32 FMAs,
 * lane = output pixel.  Added in round 5 (ABI version 3 since round 6). */
int dsf_conv_co1_forward(const float* X, const float* W, const float* bias, float* Y, int B, int Hi, int Wi, int Ci, int Ho, int Wo,
                         int K, int pad, dsf_stream_t stream);
int64_t dsf_conv_c1_workspace_bytes(int KH, int KW);
int dsf_conv_c1_forward(const float* X, const float* W, const float* bias, float* Y, int B, int Hi, int Wi, int Ho, int Wo, int Co,
                        int K, int stride, int pad, dsf_stream_t stream);
/* dsf_conv_c1_forward (bias may be NULL) that also ADDS the per-channel sum and sum of squares of Y, as doubles, into the zeroed block
 * acc [acc_rows][2][Co] (dsf_bn_acc_rows() rows): the batch statistics of the BatchNorm behind the stem convolution, consumed by
 * dsf_bn_forward_acc / dsf_bn_relu_pool_forward with acc_filled = 1 (ABI 5).  Deterministic mode: DSF_ERR_UNSUPPORTED. */
int dsf_conv_c1_forward_bn_acc(const float* X, const float* W, const float* bias, float* Y, int B, int Hi, int Wi, int Ho, int Wo, int Co, int K, int stride,
                               int pad, double* acc, int acc_rows, dsf_stream_t stream);
/* Backward of [1-channel convolution -> BatchNorm (-> ReLU) (-> MaxPool2d)] behind the sums pass (ABI 5): the BatchNorm backward's apply
 * arithmetic and the convolution's dW in ONE launch -- the gradient of the convolution output (134 MB at B = 32) is never written.
 * X (B,Hi,Wi) the convolution input, Y (B,Ho,Wo,Co) its saved output (the BatchNorm input); grad = the gradient of the BatchNorm
 * output (pool_k = 0, argmax NULL) or of the POOLED output with the argmax bytes of dsf_bn_relu_pool_forward / dsf_maxpool_forward
 * (pool (3, 2, 1) or (2, 2, 0); others DSF_ERR_UNSUPPORTED); relu: the mask is recomputed from Y (dsf_bn_backward's mode 2).
 * acc: the accumulation rows a sums-only pass filled (dsf_bn_backward_acc / dsf_bn_relu_pool_backward with grad_x = NULL).  dW
 * [K*K + 1][Co] is overwritten -- row K*K is the convolution's BIAS gradient (the per-channel sum of the output gradient);
 * grad_gamma / grad_beta as dsf_bn_backward_acc_pair (accumulate_affine).  dW equals dsf_conv_c1_wrw on the
 * gradient dsf_bn_backward_acc / dsf_bn_relu_pool_backward would have written, bit for bit (rows 0 .. K*K - 1).  workspace as
 * dsf_conv_c1_wrw. */
int dsf_conv_c1_wrw_bn(const float* X, const float* Y, const float* grad, const uint8_t* argmax, const float* gamma, const float* beta,
                       const float* save_mean, const float* save_invstd, const double* acc, int acc_rows, int relu, int pool_k, int pool_stride,
                       int pool_pad, float* dW, float* grad_gamma, float* grad_beta, int accumulate_affine, float* workspace, int B, int Hi,
                       int Wi, int Ho, int Wo, int Co, int K, int stride, int pad, dsf_stream_t stream);
int dsf_conv_c1_wrw(const float* X, const float* dY, float* dW, float* workspace, int B, int Hi, int Wi, int Ho, int Wo, int Co,
                    int K, int stride, int pad, int accumulate, dsf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * fp32 convolution on the bf16 matrix cores by exact operand splitting (csrc/conv_x6.hip).  Same callers as
 * dsf_conv_igemm_forward (model/backbone.py:200-233, model/resnet.py, model/hourglass.py, render_model/transfer.py).
 * Every fp32 operand is the exact sum of three bf16 values; six bf16 MFMAs with fp32 accumulation reproduce the fp32
 * product to 2^-26 relative, so results carry the same (accumulation-order) error as the fp32-MFMA kernels -- measured
 * against float64 in tests/test_gpu_conv.py -- at up to 16/6 of their rate.
 *
 * The weight operand is an IMAGE made once per weight update by dsf_conv_x6_split_weights:
 *   mode 0: W [KH][KW][Ci][Co] as the forward operand (reduction Ci, outputs Co);
 *   mode 1: the same memory as the stride-1 backward-data operand (reduction Co, outputs Ci, taps flipped).
 * dsf_conv_x6_image_bytes(KH, KW, Ck, Cn): size of an image with reduction width Ck and Cn outputs.
 * dsf_conv_x6_forward: Y[b,oy,ox,:] = bias + sum_taps X[b, oy*stride - pad_h + kh, ox*stride - pad_w + kw, :] . image
 *   (NHWC); Ci / Co are the image's reduction / output widths, Ci % 4 == 0.  dil = 2 (stride 1, even Ho and Wo): X is read
 *   as if upsampled by 2 with zeros -- ConvTranspose2d forward and the backward-data pass of stride-2 convolutions, both
 *   with a mode-1 image.  k_splits <= 0: chosen by the launcher (split-K partial sums meet in Y with float atomics).
 * ---------------------------------------------------------------------------------- */
int64_t dsf_conv_x6_image_bytes(int KH, int KW, int Ck, int Cn);
int dsf_conv_x6_split_weights(const float* W, void* image, int KH, int KW, int Ci, int Co, int mode, dsf_stream_t stream);
/* All images of a network in one launch (after an optimizer step).  jobs: device array of (n_jobs + 1) x 8 int64:
 * row j = {W address, image address, KH, KW, Ci, Co, mode, first granule of job j}, where a job has
 * dsf_conv_x6_image_granules(KH, KW, Ck, Cn) granules (one thread each) and row n_jobs holds the total in column 7. */
int64_t dsf_conv_x6_image_granules(int KH, int KW, int Ck, int Cn);
int dsf_conv_x6_split_weights_multi(const int64_t* jobs, int n_jobs, int64_t total_granules, dsf_stream_t stream);
int dsf_conv_x6_forward(const float* X, const void* image, const float* bias, float* Y, int B, int Hi, int Wi, int Ci, int Ho,
                        int Wo, int Co, int KH, int KW, int stride, int dil, int pad_h, int pad_w, int k_splits,
                        dsf_stream_t stream);
/* dsf_conv_x6_forward_splits: the K splits dsf_conv_x6_forward (k_splits <= 0) chooses for this shape; 1 = an unsplit launch
 * that stores Y.  A split launch first zero-fills Y -- one more launch in front of every small layer.
 * dsf_conv_x6_forward_into: the split launch WITHOUT that fill: it ADDS the convolution (+ bias) into a Y the caller has
 * initialised -- zeros taken from one pooled fill per step (dsf_amd/nn_conv.py: zero_pool), or a residual.  k_splits >= 2 as
 * reported by dsf_conv_x6_forward_splits, else DSF_ERR_UNSUPPORTED (also in deterministic mode, which never splits). */
int dsf_conv_x6_forward_splits(int B, int Ho, int Wo, int Ci, int Co, int KH, int KW, int dil);
int dsf_conv_x6_forward_into(const float* X, const void* image, const float* bias, float* Y, int B, int Hi, int Wi, int Ci, int Ho,
                             int Wo, int Co, int KH, int KW, int stride, int dil, int pad_h, int pad_w, int k_splits,
                             dsf_stream_t stream);
/* What dsf_conv_x6_forward (k_splits <= 0) launches for this layer, with the full geometry (dsf_conv_x6_forward_splits knows only
 * the output size and answers for the general kernels).  *variant: 0 = both operands staged through LDS, 1 = weight fragments
 * read straight from the image, 2 = the same with the input staged once per 16-channel chunk as a patch with halo that all nine
 * taps read (3 x 3, stride 1, pad 1, maps 64 / 32 / 16 / 8 wide; bit-identical to the others when unsplit; DSF_X6_PATCH=0
 * switches it off).  *k_splits: the K splits of that launch -- what to hand dsf_conv_x6_forward_into.  Either pointer may be
 * NULL.  Added in round 5 (ABI version 3 since round 6). */
int dsf_conv_x6_forward_plan(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int dil, int pad_h,
                             int pad_w, int* variant, int* k_splits);
/* Measurement aid for bench.py: launches a bare v_mfma_f32_32x32x16_bf16 loop (no memory traffic) on `workgroups` x 4 waves,
 * `iters` x 24 MFMAs each, operands = 16 KiB of bf16 pairs; returns the number of MFMAs issued (each 2*32*32*16 flop), -1 on error.
 * Timed by the caller: the matrix-pipe rate the chip sustains at the clock it holds under that load. */
int64_t dsf_mfma_bf16_probe(const void* operands, float* out, int workgroups, int iters, dsf_stream_t stream);
/* Backward-weights twin of dsf_conv_igemm_wrw (same arguments and accumulate semantics; Ci % 4 == 0 and Co % 4 == 0):
 * both operands are split on the fly and transposed by the LDS read (ds_read_b64_tr_b16); the pixel reduction is cut into
 * splits that meet in dW by float atomics. */
int dsf_conv_x6_wrw(const float* X, const float* dY, float* dW, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                    int KW, int stride, int pad_h, int pad_w, int accumulate, dsf_stream_t stream);
/* dsf_conv_x6_wrw that also returns the bias gradient from the same launch: dbias[co] += sum over all pixels of dY[.][co] (the
 * workgroups that stage a dY tile for the first K tile add its column sums; float atomics, so dbias must be zeroed -- or hold
 * what it accumulates into -- like dW under accumulate != 0).  Replaces the separate column-sum launches behind every biased
 * convolution (dsf_col_sum: 2-3 launches).  DSF_ERR_UNSUPPORTED -- nothing launched: use dsf_conv_x6_wrw_ws + dsf_col_sum -- in
 * deterministic mode, for layers whose pixels are cut into more than 64 splits (more adders per address bring nothing) and for the
 * large 3 x 3 layers that take the row-staged kernel.  Added in round 5 (ABI version 3 since round 6). */
int dsf_conv_x6_wrw_bias(const float* X, const float* dY, float* dW, float* dbias, int B, int Hi, int Wi, int Ci, int Ho, int Wo,
                         int Co, int KH, int KW, int stride, int pad_h, int pad_w, int accumulate, dsf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Fused training-mode BatchNorm2d (+ residual add) (+ ReLU) on NHWC activations, x viewed as (M, C).
 * Replaces nn.BatchNorm2d + `out += identity` + nn.ReLU of model/resnet.py:38-55, 82-98 and the
 * conv-bn-relu sequences of model/backbone.py:16-42 (3 + 1 + 1 forward and 3 + 1 backward kernels in
 * PyTorch) by three launches each way (per-workgroup partial sums, combine + finalise, one streaming pass).
 * C must be a power of two in [4, 1024] (else DSF_ERR_UNSUPPORTED and the caller keeps torch's kernels).
 * workspace: dsf_bn_workspace_bytes(C) bytes, 8-byte aligned scratch; one buffer can serve every layer of a stream
 * because calls on a stream are ordered.  The reduction is deterministic (fixed partials, fixed order, double combine).
 * running_mean / running_var (may be NULL) are updated in place with `momentum` (unbiased variance).
 * Maps of up to 1024 rows (M) take ONE launch per pass on every BatchNorm entry point of this header -- a workgroup per channel
 * quad holds its column in registers; fixed-order sums, the workspace / accumulation rows stay untouched (DSF_BN_SMALL=0: off).
 * ---------------------------------------------------------------------------------- */
int64_t dsf_bn_workspace_bytes(int C);
int dsf_bn_forward(const float* x, const float* residual, const float* gamma, const float* beta, int64_t M,
                   int C, float eps, float momentum, int relu, float* running_mean, float* running_var,
                   float* y, float* save_mean, float* save_invstd, double* workspace, dsf_stream_t stream);
int dsf_bn_apply(const float* x, const float* residual, const float* gamma, const float* beta,
                 const float* mean, const float* invstd, int64_t M, int C, int relu, float* y,
                 dsf_stream_t stream);
/* grad_x (M,C) written; grad_residual (M,C) = relu-masked grad_y, may be NULL; grad_gamma / grad_beta (C).
 * relu: 0 none; 1 ReLU mask from the saved output y (needed when a residual was added); 2 mask recomputed from x,
 * gamma, beta with the forward's own expression (no residual: y is neither read nor needs to be kept, may be NULL). */
int dsf_bn_backward(const float* x, const float* grad_y, const float* y, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_invstd, int64_t M, int C, int relu,
                    float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta,
                    double* workspace, dsf_stream_t stream);
/* The same pass for a gradient that arrives as TWO tensors, g = grad_y + grad_y2 (grad_y2 may be NULL): the output of a residual
 * block feeds the next block's first convolution AND its identity path (model/resnet.py:39-55, 78-98: `out += identity`), so
 * autograd would sum the two gradients with an elementwise pass of its own (2 reads + 1 write of the activation) before this
 * backward could run.  Here the sums pass adds them on the fly; when grad_residual is wanted it also WRITES the (masked) g
 * there and the apply pass reads x and that tensor only.  Same arithmetic as torch's add followed by dsf_bn_backward.
 * accumulate_affine != 0: grad_gamma / grad_beta are ADDED to what the buffers hold -- the second application of one layer in a
 * backward pass (train_render.py:628-703 runs the network on the synthetic and on the real batch before one `backward()`), whose
 * contribution autograd would otherwise add with a launch per parameter. */
int dsf_bn_backward_pair(const float* x, const float* grad_y, const float* grad_y2, const float* y, const float* gamma,
                         const float* beta, const float* save_mean, const float* save_invstd, int64_t M, int C, int relu,
                         float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta, int accumulate_affine,
                         double* workspace, dsf_stream_t stream);

/* out[c] = sum_m x[m][c] of a row-major (M, C) matrix (bias gradient of an NHWC convolution output:
 * the `gy.sum((0,2,3))` of nn.Conv2d's backward).  workspace: dsf_col_sum_workspace_bytes(C) bytes of 16-byte
 * aligned scratch (partial rows; deterministic two-launch reduction), or NULL (single launch with float atomics; ONE workgroup in
 * deterministic mode).  Inputs of up to 2^20 elements with C % 4 == 0 take one launch with a fixed-order fold either way. */
int64_t dsf_col_sum_workspace_bytes(int C);
int dsf_col_sum(const float* x, int64_t M, int C, float* out, float* workspace, dsf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * AdamW step of every parameter tensor in one launch (torch.optim.AdamW as constructed at train_render.py:131-139;
 * arithmetic of torch/optim/adamw.py, amsgrad off): ptrs (T,4) device array of {param, grad, exp_avg, exp_avg_sq}
 * addresses (fp32, one dense layout per tensor, walked in memory order), sizes (T) element counts, and the static
 * chunk table: chunk c covers elements [chunk_index[c] * E, +E) of tensor chunk_tensor[c], E = dsf_adamw_chunk_elems().
 * bias_correction{1,2} = 1 - beta^step are computed by the caller (no device-side step counter, no sync); scalars are
 * doubles so that derived factors (1 - beta2, lr / bias_correction1, ...) are rounded to fp32 once, as torch does.
 * ---------------------------------------------------------------------------------- */
int dsf_adamw_chunk_elems(void);
int dsf_adamw_multi(const uint64_t* ptrs, const int64_t* sizes, const int32_t* chunk_tensor,
                    const int32_t* chunk_index, int n_chunks, double lr, double beta1, double beta2, double eps,
                    double weight_decay, double bias_correction1, double bias_correction2, dsf_stream_t stream);

/* ----------------------------------------------------------------------------------
 * Convolution + BatchNorm statistics in one pass (conv -> BatchNorm pairs of model/resnet.py, model/backbone.py).
 * dsf_conv_x6_forward_bn = dsf_conv_x6_forward without a bias that ALSO writes, per 64 / 128 / 256-row output tile, one
 * partial row [2][Co] (per-channel sum and sum of squares of the tile) into bn_stats -- when the launch it chooses can
 * (unsplit reduction, weight-direct kernel); *bn_rows (host) = the number of rows written, 0 = none (the caller then runs
 * the ordinary dsf_bn_forward).  bn_stats: dsf_conv_x6_bn_stats_rows(B, Ho, Wo) * 2 * Co floats.
 * dsf_bn_forward_from_stats = dsf_bn_forward with its statistics pass replaced by those rows (finalise + apply).
 * ---------------------------------------------------------------------------------- */
int dsf_conv_x6_bn_stats_rows(int B, int Ho, int Wo);
int dsf_conv_x6_forward_bn(const float* X, const void* image, float* Y, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                           int KH, int KW, int stride, int dil, int pad_h, int pad_w, float* bn_stats, int* bn_rows,
                           dsf_stream_t stream);

/* Convolution with a fused per-channel output epilogue, Y = act((conv + bias) * scale[c] + shift[c] (+ residual)) -- the
 * evaluation-mode sequence conv -> BatchNorm (frozen statistics: scale = gamma / sqrt(var + eps), shift = beta - mean * scale)
 * (-> + identity) (-> ReLU) of model/resnet.py:18-98 in one launch (torch runs 2-4 kernels).  residual (or NULL): NHWC, the
 * shape of Y.  *applied: 1 = the epilogue ran; 0 = this shape's launch cannot carry it (split reduction / staged kernel): Y is
 * the plain convolution (+ bias) and the caller finishes with dsf_bn_apply. */
int dsf_conv_x6_forward_affine(const float* X, const void* image, const float* bias, float* Y, int B, int Hi, int Wi, int Ci,
                               int Ho, int Wo, int Co, int KH, int KW, int stride, int dil, int pad_h, int pad_w,
                               const float* scale, const float* shift, const float* residual, int relu, int* applied,
                               dsf_stream_t stream);
int dsf_bn_forward_from_stats(const float* x, const float* residual, const float* gamma, const float* beta, int64_t M, int C,
                              float eps, float momentum, int relu, float* running_mean, float* running_var, float* y,
                              float* save_mean, float* save_invstd, const float* part, int rows, dsf_stream_t stream);

/* ----------------------------------------------------------------------------------
 * The same BatchNorm passes (reference model/resnet.py:18-98: conv -> BatchNorm2d (+ skip)(+ ReLU), training mode) WITHOUT the
 * finalise launches -- the default mode.  `acc` is a caller-ZEROED block of dsf_bn_acc_rows() rows [row][2][C]
 * DOUBLES: the statistics pass (dsf_bn_forward_acc's own reduction when acc_filled == 0, or the producing convolution's
 * epilogue, dsf_conv_x6_forward_bn_acc, *filled = 1) adds per-workgroup float sums into row (workgroup mod rows) with
 * global_atomic_add_f64, and the apply kernel folds those rows in its own prologue (double, ascending), writes save_mean / save_invstd and
 * updates the running statistics: forward = 1 launch after a convolution that filled the rows (else 2), backward = 2
 * (dsf_bn_backward_acc: `acc` = a second zeroed block), against 2-3 and 3 on the ordered-partials path above.  The two paths
 * differ by the order of double additions only (~1e-16 on the sums).  All three return DSF_ERR_UNSUPPORTED in deterministic mode (callers then use
 * dsf_bn_forward / dsf_bn_forward_from_stats / dsf_bn_backward, which are bit-reproducible).
 * ---------------------------------------------------------------------------------- */
int dsf_bn_acc_rows(void);
/* Cross-replica BatchNorm (torch.nn.SyncBatchNorm semantics; the reference trains on one GPU, SURVEY 8e offers it for
 * data-parallel runs): the statistics of a layer cross the ranks ONCE per pass as 2C + 1 doubles instead of torch's
 * per-layer gather of means / invstds / counts.  Forward: dsf_bn_local_sums (sums[0..C) = sum x, sums[C..2C) = sum x^2 of
 * this replica -- from the producing convolution's epilogue rows `part` (dsf_conv_x6_forward_bn) or, part == NULL, from its own
 * reduction pass over x) -> the caller writes its row count to sums[2C] and all-reduces the 2C + 1 doubles ->
 * dsf_bn_forward_from_sums (mean / invstd / running statistics from the GLOBAL sums, then the apply pass).  Backward:
 * dsf_bn_backward_sums (sum g, sum g xhat of this replica; grad_gamma / grad_beta are the replica's own share, averaged
 * later with every other gradient) -> all-reduce of 2C doubles -> dsf_bn_backward_apply with `count` = the device address
 * of the global element count (sums[2C] of the forward).  No host synchronisation anywhere. */
/* Inference-time pieces of the frozen Consis-CycleGAN generator (reference render_model/transfer.py:393-448, the network
 * Trainer.Pretrain / FinetuneStage push every synthetic image through, train_render.py:428-435, 633-639), NHWC:
 * dsf_instnorm_forward = nn.InstanceNorm2d(affine=False, track_running_stats=False)(x) (+ residual) (+ ReLU); x, y, residual
 * (B, HW, C) f32, acc = B x 2C doubles of caller-zeroed scratch (per-sample sums); biased variance, eps as torch.
 * dsf_reflect_pad_nhwc = nn.ReflectionPad2d(pad): x (B,H,W,C) -> y (B,H+2pad,W+2pad,C), pad < H, W. */
int dsf_instnorm_forward(const float* x, const float* residual, int B, int64_t HW, int C, float eps, int relu, float* y,
                         double* acc, dsf_stream_t stream);
int dsf_reflect_pad_nhwc(const float* x, float* y, int B, int H, int W, int C, int pad, dsf_stream_t stream);
int dsf_bn_local_sums(const float* x, int64_t M, int C, const float* part, int rows, double* sums, double* workspace,
                      dsf_stream_t stream);
int dsf_bn_forward_from_sums(const float* x, const float* residual, const float* gamma, const float* beta, int64_t M, int C,
                             float eps, float momentum, int relu, float* running_mean, float* running_var, float* y,
                             float* save_mean, float* save_invstd, const double* sums, dsf_stream_t stream);
int dsf_bn_backward_sums(const float* x, const float* grad_y, const float* y, const float* gamma, const float* beta,
                         const float* save_mean, const float* save_invstd, int64_t M, int C, int relu, double* sums,
                         float* grad_gamma, float* grad_beta, double* workspace, dsf_stream_t stream);
int dsf_bn_backward_apply(const float* x, const float* grad_y, const float* y, const float* gamma, const float* beta,
                          const float* save_mean, const float* save_invstd, const double* sums, const double* count, int64_t M,
                          int C, int relu, float* grad_x, float* grad_residual, dsf_stream_t stream);
int dsf_conv_x6_forward_bn_acc(const float* X, const void* image, float* Y, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                               int KH, int KW, int stride, int dil, int pad_h, int pad_w, double* acc, int acc_rows, int* filled,
                               dsf_stream_t stream);
int dsf_bn_forward_acc(const float* x, const float* residual, const float* gamma, const float* beta, int64_t M, int C, float eps,
                       float momentum, int relu, float* running_mean, float* running_var, float* y, float* save_mean,
                       float* save_invstd, double* acc, int acc_filled, dsf_stream_t stream);
/* (dsf_bn_backward_acc with grad_x = NULL, ABI 5: the sums pass only -- the rows in `acc` then feed dsf_conv_c1_wrw_bn; M > 1024 rows) */
int dsf_bn_backward_acc(const float* x, const float* grad_y, const float* y, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd, int64_t M, int C, int relu, float* grad_x,
                        float* grad_residual, float* grad_gamma, float* grad_beta, double* acc, dsf_stream_t stream);
/* dsf_bn_backward_pair on the accumulation rows (no finalise launch) */
int dsf_bn_backward_acc_pair(const float* x, const float* grad_y, const float* grad_y2, const float* y, const float* gamma,
                             const float* beta, const float* save_mean, const float* save_invstd, int64_t M, int C, int relu,
                             float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta, int accumulate_affine,
                             double* acc, dsf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * The backbone stem's BatchNorm -> ReLU -> MaxPool2d(k, stride, pad) (reference model/backbone.py:200-204: nn.BatchNorm2d(64),
 * nn.ReLU, nn.MaxPool2d(3, 2, 1) on the B x 64 x 128 x 128 map) as one training-mode layer (ABI 5; csrc/norm.hip).
 * Forward: batch statistics of x [B, Hi, Wi, C] into the zeroed accumulation block `acc` (acc_filled = 1: they are there already,
 * left by dsf_conv_c1_forward_bn_acc / dsf_conv_x6_forward_bn_acc -- no statistics pass), then ONE apply pass that writes the pooled
 * output y [B, Ho, Wo, C], the 1-byte window position of each maximum (dsf_maxpool_forward's rule) and save_mean / save_invstd, and
 * updates the running statistics; the full-resolution normalised map is never written.  Backward: both BatchNorm backward passes
 * gather the gradient of a full-resolution element from the pooled gradient grad_y (dsf_maxpool_backward's terms in its order: the
 * same bits) and recompute the ReLU mask from x; `acc` is a second zeroed block; grad_x = NULL: the sums pass only (the rows in `acc`
 * then feed dsf_conv_c1_wrw_bn).  Results equal dsf_bn_forward_acc(relu = 1) +
 * dsf_maxpool_forward resp. dsf_maxpool_backward + dsf_bn_backward_acc(relu = 2) bit for bit (grad_x, grad_gamma, grad_beta up to the
 * order of the double-precision atomic sums, as for those entries).  k in {2, 3}, k <= 2 stride, 2 pad <= k, C % 4 == 0 with C / 4 dividing 256 or a multiple of it,
 * B Hi Wi C / 4 < 2^31; anything else (and the deterministic mode) DSF_ERR_UNSUPPORTED: run the separate layers.
 * ------------------------------------------------------------------------------------ */
int dsf_bn_relu_pool_forward(const float* x, const float* gamma, const float* beta, int B, int Hi, int Wi, int C, int k, int stride,
                             int pad, float eps, float momentum, float* running_mean, float* running_var, float* y, uint8_t* argmax,
                             float* save_mean, float* save_invstd, double* acc, int acc_filled, dsf_stream_t stream);
int dsf_bn_relu_pool_backward(const float* x, const float* grad_y, const uint8_t* argmax, const float* gamma, const float* beta,
                              const float* save_mean, const float* save_invstd, int B, int Hi, int Wi, int C, int k, int stride, int pad,
                              float* grad_x, float* grad_gamma, float* grad_beta, int accumulate_affine, double* acc,
                              dsf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Loss-side glue of the trainer steps, one launch (pair) per term (round 6; csrc/step_ops.hip).  The reference writes each of
 * these as a chain of 10-25 elementwise / reduce torch operators (forward and backward); on one stream that is launch latency.
 *
 * dsf_m2d_*: model-to-data depth term, /root/reference/train_render.py:728-732 (also :556-558): per sample over the P pixels of a
 *   crop, u = (real < thresh) | (synth < thresh); per[b] = sum |real - synth| u / (sum u + 1e-8); loss[0] = mean_b per[b] * scale.
 *   sums (B,4) = {sum |d| u, sum u, sum |d| a, sum a} with a = the AND of the two masks: the agreement test of the M2P gate
 *   (:786-789) reads them.  Gradient w.r.t. synth only.
 * dsf_cube_points_*: Render.render's point transforms, /root/reference/render_model/mano_layer.py:1078-1092: world = p * cube / 2 +
 *   center and norm = (world - center) / cube * 2 for the vertex (B,NV,3) and the joint (B,NJ,3) tensor in one launch each way
 *   (any of the four incoming gradients may be NULL).
 * dsf_view_rotate: RotationPoints (mano_layer.py:874-884) through batch_rodrigues / quat2mat (:773-805): rot (B,3) axis-angle or
 *   (B,4) quaternion (NULL: none), points rotated about center (B,3); recentre != 0 first moves the points so that the mean of the
 *   joints sits at center (Render.forward :995-1003).  dsf_cube_normalise: (p - center) / cube * 2 for both tensors (:1033-1034).
 *   Forward only (the synthetic branch renders without gradients).
 * dsf_m2p_*: the M2P term, /root/reference/train_render.py:590-603 / 787-801: Huber (delta) between the pixel branch's and the MANO
 *   branch's joints (B,21,3) over the rows (b, j) with sample_ok[b] (uint8) and part_dist[b][part(j)] < part_thresh (joint 0 always,
 *   joints 16..20 follow parts 2, 5, 8, 11, 14), mean over the selected rows x weight, 0 when no row with a positive index is
 *   selected (the reference's host test); aux = {rows selected, that flag}.  Gradient w.r.t. juvd_pix only.
 * dsf_part_mean_*: the per-part masked means of JointICPLoss / FingerICPLoss, /root/reference/metric/meshLoss.py:389-394: out[b][k] =
 *   sum of dis over the points labelled k + 1, divided by (the number of those with dis > 0) + 1e-8, 0 when there are none;
 *   valid (B,n_parts) = those counts (kept for the backward pass).  n_parts <= 16.
 * dsf_mano_reg_*: the two regularisers of Pretrain, /root/reference/train_render.py:463-464, on the packed parameter rows (B,W):
 *   out[0] = mean(p[:, beta_col:beta_col+10]^2) * w_beta, out[1] = mean(|min(p[:, scale_col], 0)|) * w_scale; the backward
 *   writes EVERY column of grad_paras (zeros outside the eleven).
 * All reductions run in a fixed order (deterministic).
 * ---------------------------------------------------------------------------------- */
int dsf_m2d_forward(const float* real, const float* synth, int B, int P, float thresh, float scale, float* sums, float* per,
                    float* loss, dsf_stream_t stream);
int dsf_m2d_backward(const float* real, const float* synth, const float* sums, const float* grad_loss, int B, int P, float thresh,
                     float scale, float* grad_synth, dsf_stream_t stream);
int dsf_cube_points_forward(const float* verts, const float* joints, const float* center, const float* cube, int B, int NV, int NJ,
                            float* verts_world, float* joints_world, float* verts_norm, float* joints_norm, dsf_stream_t stream);
int dsf_cube_points_backward(const float* g_verts_world, const float* g_joints_world, const float* g_verts_norm,
                             const float* g_joints_norm, const float* cube, int B, int NV, int NJ, float* g_verts, float* g_joints,
                             dsf_stream_t stream);
int dsf_view_rotate(const float* verts, const float* joints, const float* center, const float* rot, int rot_dim, int recentre, int B,
                    int NV, int NJ, float* verts_out, float* joints_out, dsf_stream_t stream);
int dsf_cube_normalise(const float* verts, const float* joints, const float* center, const float* cube, int B, int NV, int NJ,
                       float* verts_norm, float* joints_norm, dsf_stream_t stream);
int dsf_m2p_forward(const float* juvd_pix, const float* juvd_mano, const unsigned char* sample_ok, const float* part_dist, int B,
                    float part_thresh, float delta, float weight, float* out, float* aux, dsf_stream_t stream);
int dsf_m2p_backward(const float* juvd_pix, const float* juvd_mano, const unsigned char* sample_ok, const float* part_dist,
                     const float* aux, const float* grad_out, int B, float part_thresh, float delta, float weight, float* grad_pix,
                     dsf_stream_t stream);
/* The MANO regression head, /root/reference/model/backbone.py:225-226 (`nn.AdaptiveAvgPool2d(1)`, flatten, `nn.Linear(C, 62)`) on a
 * channels-last feature map x (B, HW, C): pooled (B,C) = mean over the pixels (kept for the backward pass), out (B,O) = pooled W^T +
 * bias.  Backward: grad_x (B,HW,C) = (grad_out W) / HW broadcast over the pixels, grad_weight (O,C), grad_bias (O) (any of the
 * three may be NULL; grad_bias rides with grad_weight).  C <= 2048, O <= 64.  Fixed-order sums. */
int dsf_pool_linear_forward(const float* x, const float* weight, const float* bias, int B, int HW, int C, int O, float* pooled,
                            float* out, dsf_stream_t stream);
int dsf_pool_linear_backward(const float* grad_out, const float* pooled, const float* weight, int B, int HW, int C, int O,
                             float* grad_x, float* grad_weight, float* grad_bias, dsf_stream_t stream);
/* torch.cat((a, b, c, d), dim = 1) of channels-last maps as one launch (the stage-2 input of reference model/backbone.py:256; ABI 5):
 * out (pixels, ca + cb + cc + cd) from a (pixels, ca) ... d (pixels, cd); trailing sources may be absent (count 0, pointer NULL);
 * channel counts are multiples of 4, pixels x total / 4 < 2^32, else DSF_ERR_UNSUPPORTED.  A copy: bit-exact. */
int dsf_cat_channels_nhwc(const float* a, int ca, const float* b, int cb, const float* c, int cc, const float* d, int cd, float* out,
                          int64_t pixels, dsf_stream_t stream);
int dsf_part_mean_forward(const float* dis, const int64_t* seg, int B, int P, int n_parts, float* out, float* valid,
                          dsf_stream_t stream);
int dsf_part_mean_backward(const float* grad_out, const int64_t* seg, const float* valid, int B, int P, int n_parts, float* grad_dis,
                           dsf_stream_t stream);
int dsf_mano_reg_forward(const float* paras, int B, int W, int beta_col, int scale_col, float w_beta, float w_scale, float* out,
                         dsf_stream_t stream);
int dsf_mano_reg_backward(const float* paras, const float* grad_out, int B, int W, int beta_col, int scale_col, float w_beta,
                          float w_scale, float* grad_paras, dsf_stream_t stream);

/* ----------------------------------------------------------------------------------
 * Training-phase augmentation of cropped frames: `loader.augmentCrop`
 * (/root/reference/data/render_loader.py:653-695 = rotateHand :458-497 / moveCoM :427-456 / scaleHand :499-527 through
 * recropHand :403-424, then normalize_img :738-745) for a batch, replacing the cv2 / numpy DataLoader workers.
 *   crop (B,S,S) f32: the un-normalised crops (dsf_depth_crop_normalize's raw_crop); joints (B,J,3) f32 mm relative to
 *   the centre; com (B,3) f64 image coordinates (u, v, z); cube (B,3) f64 mm; M (B,3,3) f64 the crop transform;
 *   mode (B) int32 index into the reference's aug_modes ['rot','com','sc','none'] (:1816); off (B,3) f64 mm, rot (B) f64
 *   degrees, sc (B) f64: the `rand_augment` draws (:625-650) as explicit inputs; (fx, fy, fu, fv), flip: camera.
 *   -> img (B,1,S,S) f32 normalised to [-1,1], joints_out (B,J,3) f32, cube_out / com_out (B,3) f64, M_out (B,3,3) f64.
 * cv2.warpPerspective / getRotationMatrix2D / warpAffine (INTER_NEAREST, BORDER_CONSTANT) follow OpenCV's published rules.
 * ---------------------------------------------------------------------------------- */
int dsf_depth_augment_crop(const float* crop, const float* joints, const double* com, const double* cube, const double* M,
                           const int32_t* mode, const double* off, const double* rot, const double* sc, double fx, double fy,
                           double fu, double fv, int flip, int B, int S, int J, float* img, float* joints_out,
                           double* cube_out, double* com_out, double* M_out, dsf_stream_t stream);

/* ----------------------------------------------------------------------------------
 * Deterministic mode (SURVEY 5.2, 8b "deterministic segmented reduce (flag)"; pytorch3d's backward kernels use float
 * atomicAdd and are not reproducible run to run).  dsf_set_deterministic(1) -- or DSF_DETERMINISTIC=1 in the environment
 * when the library is loaded -- makes every launcher of this library bit-reproducible: the raster / point-face / collision
 * backward kernels accumulate in 64-bit fixed point (order-independent), the forward-type convolutions do not split
 * their reduction, backward-weights writes one partial tile per pixel split and adds them in ascending order
 * (dsf_conv_x6_wrw_ws with dsf_conv_x6_wrw_workspace_bytes(...) bytes of scratch; dsf_conv_x6_wrw refuses in this mode;
 * accumulate != 0 starts the ordered sum from what dW holds, as in the default mode).
 * Returns the previous setting.  Off (default): float atomics, results agree to ~1e-7 relative.
 * ---------------------------------------------------------------------------------- */
int dsf_set_deterministic(int on);
int dsf_get_deterministic(void);
int64_t dsf_conv_x6_wrw_workspace_bytes(int B, int Ho, int Wo, int Ci, int Co, int KH, int KW);
int dsf_conv_x6_wrw_ws(const float* X, const float* dY, float* dW, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                       int KW, int stride, int pad_h, int pad_w, int accumulate, float* workspace, dsf_stream_t stream);

/* ----------------------------------------------------------------------------------
 * NHWC max pooling (nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the backbone stem, reference
 * model/backbone.py:200-204; nn.MaxPool2d(2, 2) of model/hourglass.py:131).  x (B,Hi,Wi,C) -> y (B,Ho,Wo,C), C % 4 == 0.
 * argmax (B,Ho,Wo,C) uint8: window position kh * k + kw of the maximum (torch's rule: first maximum in scan order, NaN
 * wins), consumed by the backward, which is a deterministic gather (no atomics, grad_x fully written).
 * ---------------------------------------------------------------------------------- */
int dsf_maxpool_forward(const float* x, float* y, uint8_t* argmax, int B, int Hi, int Wi, int C, int Ho, int Wo, int k,
                        int stride, int pad, dsf_stream_t stream);
int dsf_maxpool_backward(const float* grad_y, const uint8_t* argmax, float* grad_x, int B, int Hi, int Wi, int C, int Ho,
                         int Wo, int k, int stride, int pad, dsf_stream_t stream);

/* ----------------------------------------------------------------------------------
 * Self-intersection volume of watertight hand parts: replaces trimesh's `voxelized(pitch)` +
 * `contains(points)` inside the reference's `self_intersection` (/root/reference/eval_coll.py:611-626; also
 * util/intersect.py:102-107) for a BATCH of meshes.
 *   verts (B,V,3) f32: the vertex pool of every sample (the 779 MANO vertices + the cap vertices the reference
 *       appends, eval_coll.py:348-366); faces (n_faces,3) int32 into that pool, the faces of part i being rows
 *       [part_first[i], part_first[i+1]) (part_first has n_parts + 1 entries); pairs (n_pairs,2) int32 = the (s, t) part
 *       pairs to evaluate (the reference: t > s, not parent / child); max_part_faces = the largest part's face count.
 *   pitch: voxel size (the reference uses 2, then 1 for the colliding meshes); grid: cells per axis of a part's occupancy
 *       mask, a multiple of 32 with (part extent / pitch) + 3 <= grid.
 *   workspace: dsf_part_volume_workspace_bytes(B, n_parts, grid) bytes (16-byte aligned).
 *   count (B) uint64: sum over the pairs of #{surface voxels of part t inside part s}  (volume = count * pitch^3);
 *   pair_count (B, n_pairs) int32 or NULL; status (1) int32: 0 ok, bit 0 = a part does not fit `grid`, bit 1 = a face
 *       needs more than trimesh's 10 subdivision rounds (the reference raises there).  All outputs are integers:
 *       bit-exact against oracle/volume_ref.py.  No allocation, no sync (the caller reads `status` when it reads `count`).
 * ---------------------------------------------------------------------------------- */
int64_t dsf_part_volume_workspace_bytes(int B, int n_parts, int grid);
int dsf_part_intersection_volume(const float* verts, const int32_t* faces, const int32_t* part_first, const int32_t* pairs,
                                 int B, int V, int n_parts, int n_faces, int n_pairs, int max_part_faces, double pitch, int grid,
                                 void* workspace, unsigned long long* count, int32_t* pair_count, int32_t* status,
                                 dsf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DSF_HIP_H */
