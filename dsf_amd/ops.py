"""torch-facing wrappers of the C ABI (include/dsf_hip.h): allocation of outputs,
raw-pointer hand-off on the current HIP stream, and autograd registration.
PyTorch is plumbing here (device memory, streams, autograd graph); all
arithmetic happens in the HIP kernels.  No CPU path exists: CPU tensors raise.
"""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L
from ._lib import F, I, I64, ptr, f32, check, stream_ptr

SAVE_FLOATS = 5248            # DSF_MANO_SAVE_FLOATS
BWD_SCRATCH_FLOATS = 2560     # DSF_MANO_BWD_SCRATCH_FLOATS


def _empty(shape, ref, dtype=torch.float32):
    return torch.empty(shape, device=ref.device, dtype=dtype)


# --------------------------------------------------------------------------------------------
# K5 MANO
# --------------------------------------------------------------------------------------------
class ManoFunction(Function):
    """MANO_SMPL.forward / get_mano_vertices (render_model/mano_layer.py:573-693)."""

    @staticmethod
    def forward(ctx, model, beta, theta, rot, cam, k1, k2):
        beta, theta, rot = f32(beta), f32(theta), f32(rot)
        cam = f32(cam) if cam is not None else None
        B, ncomp, rot_dim = beta.shape[0], theta.shape[1], rot.shape[1]
        verts = _empty((B, 779, 3), beta)
        joints = _empty((B, 21, 3), beta)
        Rs = _empty((B, 15, 3, 3), beta)
        save = _empty((B, SAVE_FLOATS), beta)          # staging between the two forward launches + state of the backward pass
        check(L.lib().dsf_mano_forward(ctypes.byref(model.c_struct), ptr(beta), ptr(theta), ptr(rot), ptr(cam), I(B),
                                       I(ncomp), I(rot_dim), I(0), F(k1), F(k2), ptr(verts), ptr(joints), ptr(Rs),
                                       ptr(save), stream_ptr()), "dsf_mano_forward")
        ctx.model, ctx.k = model, (k1, k2)
        ctx.has_cam = cam is not None
        ctx.save_for_backward(theta, rot, cam, save)
        ctx.mark_non_differentiable(Rs)
        return verts, joints, Rs

    @staticmethod
    @once_differentiable
    def backward(ctx, g_verts, g_joints, _g_rs):
        theta, rot, cam, save = ctx.saved_tensors
        B, ncomp, rot_dim = theta.shape[0], theta.shape[1], rot.shape[1]
        g_verts = f32(g_verts) if g_verts is not None else None
        g_joints = f32(g_joints) if g_joints is not None else None
        g_beta = _empty((B, 10), theta)
        g_theta = _empty((B, ncomp), theta)
        g_rot = _empty((B, rot_dim), theta)
        g_cam = _empty((B, 4), theta) if ctx.has_cam else None
        k1, k2 = ctx.k
        check(L.lib().dsf_mano_backward(ctypes.byref(ctx.model.c_struct), ptr(theta), ptr(rot), ptr(cam), ptr(save),
                                        ptr(g_verts), ptr(g_joints), I(B), I(ncomp), I(rot_dim), I(0), F(k1), F(k2),
                                        ptr(g_beta), ptr(g_theta), ptr(g_rot), ptr(g_cam),
                                        ptr(_empty((B, BWD_SCRATCH_FLOATS), theta)), stream_ptr()), "dsf_mano_backward")
        return None, g_beta, g_theta, g_rot, g_cam, None, None


class ManoPackedFunction(Function):
    """ManoFunction on the network's packed parameter rows (B, 3|4 + 45 + 10 + 4) = [rot | theta | beta | cam]
    (Render._split's layout): the kernels read the four fields as column offsets of one row and write one gradient row,
    so the ~12 slice / contiguous / zero-fill / add kernels per call of the sliced formulation disappear."""

    @staticmethod
    def forward(ctx, model, paras, k1, k2):
        paras = f32(paras)
        B, W = paras.shape
        rot_dim = W - 59
        assert rot_dim in (3, 4)
        verts = _empty((B, 779, 3), paras)
        joints = _empty((B, 21, 3), paras)
        save = _empty((B, SAVE_FLOATS), paras)
        col = lambda c: ctypes.c_void_p((paras.data_ptr() if B else L._dummy(paras.device).data_ptr()) + 4 * c)
        check(L.lib().dsf_mano_forward(ctypes.byref(model.c_struct), col(rot_dim + 45), col(rot_dim), col(0), col(rot_dim + 55),
                                       I(B), I(45), I(rot_dim), I(W), F(k1), F(k2), ptr(verts), ptr(joints), ptr(None),
                                       ptr(save), stream_ptr()), "dsf_mano_forward")
        ctx.model, ctx.k = model, (k1, k2)
        ctx.save_for_backward(paras, save)
        return verts, joints

    @staticmethod
    @once_differentiable
    def backward(ctx, g_verts, g_joints):
        paras, save = ctx.saved_tensors
        B, W = paras.shape
        rot_dim = W - 59
        g_verts = f32(g_verts) if g_verts is not None else None
        g_joints = f32(g_joints) if g_joints is not None else None
        g = _empty((B, W), paras)                                        # every column is written by the kernel
        base = lambda t, c: ctypes.c_void_p((t.data_ptr() if B else L._dummy(t.device).data_ptr()) + 4 * c)
        k1, k2 = ctx.k
        check(L.lib().dsf_mano_backward(ctypes.byref(ctx.model.c_struct), base(paras, rot_dim), base(paras, 0),
                                        base(paras, rot_dim + 55), ptr(save), ptr(g_verts), ptr(g_joints), I(B), I(45),
                                        I(rot_dim), I(W), F(k1), F(k2), base(g, rot_dim + 45), base(g, rot_dim), base(g, 0),
                                        base(g, rot_dim + 55), ptr(_empty((B, BWD_SCRATCH_FLOATS), paras)), stream_ptr()),
              "dsf_mano_backward")
        return None, g, None, None


# --------------------------------------------------------------------------------------------
# K1/K2 rasteriser (pytorch3d._C contract) and the fused crop renderer
# --------------------------------------------------------------------------------------------
def project_face_verts(verts, faces_i32, cam):
    verts = f32(verts)
    N, V, _ = verts.shape
    Fn = faces_i32.shape[0]
    out = _empty((N * Fn, 3, 3), verts)
    check(L.lib().dsf_project_face_verts(ptr(verts), ptr(faces_i32), ctypes.byref(cam), I(N), I(V), I(Fn), ptr(out),
                                         stream_ptr()), "dsf_project_face_verts")
    return out


class RasterizeMeshesFunction(Function):
    """pytorch3d._C.rasterize_meshes(+_backward) for the reference's settings (mano_layer.py:946-951)."""

    @staticmethod
    def forward(ctx, face_verts, mesh_to_face_first_idx, num_faces_per_mesh, image_size, blur_radius=0.0,
                faces_per_pixel=1, bin_size=None, max_faces_per_bin=None, perspective_correct=False,
                clip_barycentric_coords=False, cull_backfaces=False):
        face_verts = f32(face_verts)
        N = mesh_to_face_first_idx.shape[0]
        S = int(image_size)
        p2f = _empty((N, S, S, 1), face_verts, torch.int64)
        zbuf = _empty((N, S, S, 1), face_verts)
        bary = _empty((N, S, S, 1, 3), face_verts)
        dists = _empty((N, S, S, 1), face_verts)
        ws = _empty((max(N, 1), 4), face_verts)
        check(L.lib().dsf_rasterize_meshes(ptr(face_verts), ptr(mesh_to_face_first_idx.contiguous()),
                                           ptr(num_faces_per_mesh.contiguous()), I(N), I64(face_verts.shape[0]), I(S),
                                           F(blur_radius), I(faces_per_pixel), I(int(perspective_correct)),
                                           I(int(clip_barycentric_coords)), I(int(cull_backfaces)), ptr(p2f),
                                           ptr(zbuf), ptr(bary), ptr(dists), ptr(ws), stream_ptr()),
              "dsf_rasterize_meshes")
        ctx.save_for_backward(face_verts, p2f)
        ctx.S = S
        ctx.mark_non_differentiable(p2f)
        return p2f, zbuf, bary, dists

    @staticmethod
    @once_differentiable
    def backward(ctx, _g_p2f, g_zbuf, g_bary, g_dists):
        face_verts, p2f = ctx.saved_tensors
        # only zbuf is consumed by the reference (mano_layer.py:1023); bary/dists grads must be zero
        g = _empty(face_verts.shape, face_verts)
        check(L.lib().dsf_rasterize_meshes_backward(ptr(face_verts), ptr(p2f), ptr(f32(g_zbuf)), ptr(None), ptr(None),
                                                    I(p2f.shape[0]), I64(face_verts.shape[0]), I(ctx.S), ptr(g),
                                                    stream_ptr()), "dsf_rasterize_meshes_backward")
        return (g,) + (None,) * 10


# ---- per-batch constants: computed once per input tensor, not once per call site --------------------------------------------------
# The crop geometry of a batch (centre, M, bounds from ``comToBounds`` / ``Offset2Trans``) and ``torch.inverse(M)`` depend on the
# batch's centre / cube / M INPUTS only, yet the reference recomputes them inside every utility it calls (train_render.py:693-808
# goes through ``crop_hand`` / ``Img2pcl`` / ``uvd_nl2xyznl_tensor`` / ``render`` a dozen times per step with one M: 12 LAPACK-style
# inversions = 60 rocsolver launches + their copies per config-5 step until round 6).  ``_memo`` keys a value on the identity
# (address, shape, dtype) AND version counter of the tensors it was computed from and HOLDS those tensors, so an address cannot come
# back as another tensor while the entry lives and an in-place write (``copy_`` into a static input buffer) makes a new entry.
# Same routine on the same bits: the values are those of the uncached calls.  Never inside a stream capture (the kernels must be in
# the graph: GraphedStep refreshes its static inputs in place) and never for tensors that carry gradients.
_MEMO = {}


def _memo(name, keys, fn, extra=()):
    if torch.cuda.is_current_stream_capturing() or any(k.requires_grad for k in keys):
        return fn()
    sig = tuple((k.data_ptr(), k._version, tuple(k.shape), k.dtype) for k in keys) + tuple(extra)
    entries = _MEMO.setdefault(name, [])
    for e in entries:
        if e[0] == sig:
            return e[2]
    val = fn()
    entries.append((sig, tuple(keys), val))
    if len(entries) > 8:
        entries.pop(0)
    return val


def memo_clear():
    _MEMO.clear()


def inverse3x3(M):
    """``torch.inverse(M)`` of a batch's (B,3,3) crop transforms (the reference's own routine: its LAPACK-style rounding decides
    exact-.5 ties of the nearest-neighbour crop), once per M tensor.  No host synchronisation."""
    M = M.reshape(-1, 3, 3).float()
    return _memo("inv", (M,), lambda: torch.linalg.inv_ex(M)[0].contiguous())     # (LAPACK-style output is column-major: one copy here, not one per consumer)


def column(t, c):
    """``t[:, c].contiguous()`` once per tensor (the depth column of a batch's centres / cubes, read by every crop launch)"""
    return _memo("col%d" % c, (t,), lambda: t[:, c].contiguous())


def crop_setup(center3d, cube, cam, crop=128, want_closed_inverse=False):
    center3d, cube = f32(center3d), f32(cube)

    def run():
        B = center3d.shape[0]
        c2 = _empty((B, 3), center3d)
        M = _empty((B, 3, 3), center3d)
        bounds = _empty((B, 4), center3d, torch.int32)
        minv = _empty((B, 3, 3), center3d) if want_closed_inverse else None
        check(L.lib().dsf_crop_setup(ptr(center3d), ptr(cube), ctypes.byref(cam), I(B), I(crop), ptr(c2), ptr(M),
                                     ptr(bounds), ptr(minv), stream_ptr()), "dsf_crop_setup")
        return c2, M, bounds, minv
    return _memo("crop_setup", (center3d, cube), run,
                 (cam.fx, cam.fy, cam.px, cam.py, cam.img_w, cam.img_h, int(crop), bool(want_closed_inverse)))


class RenderCropFunction(Function):
    """verts (B,V,3) world mm -> normalised 128x128 depth crop (Render.render leaf, mano_layer.py:1082-1092)."""

    @staticmethod
    def forward(ctx, verts, faces_i32, minv, rowmap, center_z, cube_z, cam, raster_size, crop):
        verts, minv = f32(verts), f32(minv)
        center_z = f32(center_z) if center_z is not None else None
        cube_z = f32(cube_z) if cube_z is not None else None
        B, V, _ = verts.shape
        img = _empty((B, 1, crop, crop), verts)
        p2f = _empty((B, crop, crop), verts, torch.int32)
        check(L.lib().dsf_render_crop_forward(ptr(verts), ptr(faces_i32), ptr(minv), ptr(rowmap), ptr(center_z),
                                              ptr(cube_z), ctypes.byref(cam), I(B), I(V), I(faces_i32.shape[0]),
                                              I(raster_size), I(crop), ptr(img), ptr(p2f), stream_ptr()),
              "dsf_render_crop_forward")
        ctx.save_for_backward(verts, faces_i32, minv, rowmap, center_z, cube_z, p2f)
        ctx.cam, ctx.dims = cam, (raster_size, crop)
        ctx.mark_non_differentiable(p2f)
        return img, p2f

    @staticmethod
    @once_differentiable
    def backward(ctx, g_img, _g_p2f):
        verts, faces_i32, minv, rowmap, center_z, cube_z, p2f = ctx.saved_tensors
        B, V, _ = verts.shape
        raster_size, crop = ctx.dims
        g = _empty(verts.shape, verts)
        check(L.lib().dsf_render_crop_backward(ptr(verts), ptr(faces_i32), ptr(minv), ptr(rowmap), ptr(center_z),
                                               ptr(cube_z), ctypes.byref(ctx.cam), ptr(p2f), ptr(f32(g_img)), I(B),
                                               I(V), I(faces_i32.shape[0]), I(raster_size), I(crop), ptr(g),
                                               stream_ptr()), "dsf_render_crop_backward")
        return (g,) + (None,) * 8


# --------------------------------------------------------------------------------------------
# K3/K4 point-face distance
# --------------------------------------------------------------------------------------------
class PointFaceDistance(Function):
    """pytorch3d._C.point_face_dist_forward/backward (metric/meshLoss.py:21-70)."""

    @staticmethod
    def forward(ctx, points, points_first_idx, tris, tris_first_idx, max_points):
        points, tris = f32(points), f32(tris)
        P, T, N = points.shape[0], tris.shape[0], points_first_idx.shape[0]
        dists = _empty((P,), points)
        idxs = _empty((P,), points, torch.int64)
        check(L.lib().dsf_point_face_dist_forward(ptr(points), ptr(points_first_idx.contiguous()), ptr(tris),
                                                  ptr(tris_first_idx.contiguous()), I(N), I64(P), I64(T),
                                                  I64(int(max_points)), ptr(dists), ptr(idxs), stream_ptr()),
              "dsf_point_face_dist_forward")
        ctx.save_for_backward(points, tris, idxs)
        return dists

    @staticmethod
    @once_differentiable
    def backward(ctx, g_dists):
        points, tris, idxs = ctx.saved_tensors
        gp = _empty(points.shape, points)
        gt = _empty(tris.shape, tris)
        check(L.lib().dsf_point_face_dist_backward(ptr(points), ptr(tris), ptr(idxs), ptr(f32(g_dists)),
                                                   I64(points.shape[0]), I64(tris.shape[0]), ptr(gp), ptr(gt),
                                                   stream_ptr()), "dsf_point_face_dist_backward")
        return gp, None, gt, None, None


class MeshPointDistance(Function):
    """Fused batched point->mesh(-part) distance behind ICPLoss/JointICPLoss (metric/meshLoss.py:347-395)."""

    @staticmethod
    def forward(ctx, verts, points, faces_cat, part_first, seg, n_parts):
        verts, points = f32(verts), f32(points)
        B, V, _ = verts.shape
        P = points.shape[1]
        dists = _empty((B, P), verts)
        idxs = _empty((B, P), verts, torch.int32)
        seg_c = seg.contiguous() if seg is not None else None
        if seg_c is not None and seg_c.dtype != torch.int64:
            seg_c = seg_c.long()
        check(L.lib().dsf_mesh_point_dist_forward(ptr(verts), ptr(points), ptr(faces_cat), ptr(part_first), ptr(seg_c),
                                                  I(B), I(V), I(P), I(n_parts), ptr(dists), ptr(idxs), stream_ptr()),
              "dsf_mesh_point_dist_forward")
        ctx.save_for_backward(verts, points, faces_cat, idxs)
        ctx.need_points = ctx.needs_input_grad[1]
        ctx.mark_non_differentiable(idxs)
        return dists, idxs

    @staticmethod
    @once_differentiable
    def backward(ctx, g_dists, _g_idx):
        verts, points, faces_cat, idxs = ctx.saved_tensors
        B, V, _ = verts.shape
        gv = _empty(verts.shape, verts)
        gp = _empty(points.shape, points) if ctx.need_points else None
        check(L.lib().dsf_mesh_point_dist_backward(ptr(verts), ptr(points), ptr(faces_cat), ptr(idxs),
                                                   ptr(f32(g_dists)), I(B), I(V), I(points.shape[1]), ptr(gv), ptr(gp),
                                                   stream_ptr()), "dsf_mesh_point_dist_backward")
        return gv, gp, None, None, None, None


# --------------------------------------------------------------------------------------------
# K6/K7 spheres, collision, segmentation
# --------------------------------------------------------------------------------------------
def sphere_set(sphere_model, joints, mesh):
    joints, mesh = f32(joints), f32(mesh)
    B, V = joints.shape[0], mesh.shape[1]
    c = _empty((B, 66, 3), joints)
    r = _empty((B, 66), joints)
    check(L.lib().dsf_sphere_set(ctypes.byref(sphere_model), ptr(joints), ptr(mesh), I(B), I(V), ptr(c), ptr(r),
                                 ptr(None), stream_ptr()), "dsf_sphere_set")
    return c, r


def sphere_mixed(sphere_model, joints_centres, joints_radii, mesh):
    """seg_pcl's sphere set (mano_layer.py:404-413): centres from one skeleton, radii from the other, one launch; once per
    (joints, joints, mesh) triple (the trainer labels two clouds against the same spheres, train_render.py:695-700)."""
    jc, jr, mesh = f32(joints_centres), f32(joints_radii), f32(mesh)

    def run():
        B, V = jc.shape[0], mesh.shape[1]
        c = _empty((B, 66, 3), jc)
        r = _empty((B, 66), jc)
        check(L.lib().dsf_sphere_mixed(ctypes.byref(sphere_model), ptr(jc), ptr(jr), ptr(mesh), I(B), I(V), ptr(c), ptr(r), stream_ptr()),
              "dsf_sphere_mixed")
        return c, r
    return _memo("sphere_mixed", (jc, jr, mesh), run)


class CollisionRows(Function):
    """Gated per-sphere row sums of calculate_coll (mano_layer.py:373-386)."""

    @staticmethod
    def forward(ctx, sphere_model, joints, mesh):
        joints, mesh = f32(joints), f32(mesh)
        B, V = joints.shape[0], mesh.shape[1]
        rows = _empty((B, 66), joints)
        c = _empty((B, 66, 3), joints)
        r = _empty((B, 66), joints)
        topk = _empty((B, 16, 10), joints, torch.int32)
        check(L.lib().dsf_collision_forward(ctypes.byref(sphere_model), ptr(joints), ptr(mesh), I(B), I(V), ptr(rows),
                                            ptr(c), ptr(r), ptr(topk), stream_ptr()), "dsf_collision_forward")
        ctx.sm = sphere_model
        ctx.save_for_backward(joints, mesh, c, r, topk)
        return rows

    @staticmethod
    @once_differentiable
    def backward(ctx, g_rows):
        joints, mesh, c, r, topk = ctx.saved_tensors
        B, V = joints.shape[0], mesh.shape[1]
        gj = _empty(joints.shape, joints)
        gm = _empty(mesh.shape, mesh)
        check(L.lib().dsf_collision_backward(ctypes.byref(ctx.sm), ptr(joints), ptr(mesh), ptr(c), ptr(r), ptr(topk),
                                             ptr(f32(g_rows)), I(B), I(V), ptr(gj), ptr(gm), stream_ptr()),
              "dsf_collision_backward")
        return None, gj, gm


def seg_pcl(centres, radii, pcl):
    centres, radii, pcl = f32(centres), f32(radii), f32(pcl)
    B, P = pcl.shape[0], pcl.shape[1]
    labels = _empty((B, P), pcl, torch.int64)
    check(L.lib().dsf_seg_pcl(ptr(centres), ptr(radii), ptr(pcl), I(B), I(P), ptr(labels), stream_ptr()), "dsf_seg_pcl")
    return labels


# --------------------------------------------------------------------------------------------
# K8/K9/K10 image-side ops
# --------------------------------------------------------------------------------------------
class UvdToXyz(Function):
    @staticmethod
    def forward(ctx, uvd, center, minv, cube, cam, img_size, normalise):
        uvd, center, minv, cube = f32(uvd), f32(center), f32(minv), f32(cube)
        B, N, _ = uvd.shape
        out = _empty(uvd.shape, uvd)
        check(L.lib().dsf_uvd_to_xyz(ptr(uvd), ptr(center), ptr(minv), ptr(cube), ctypes.byref(cam), I(B), I(N),
                                     I(img_size), I(int(normalise)), ptr(out), stream_ptr()), "dsf_uvd_to_xyz")
        ctx.save_for_backward(uvd, center, minv, cube)
        ctx.args = (cam, img_size, int(normalise))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        uvd, center, minv, cube = ctx.saved_tensors
        cam, img_size, normalise = ctx.args
        B, N, _ = uvd.shape
        out = _empty(uvd.shape, uvd)
        check(L.lib().dsf_uvd_to_xyz_backward(ptr(uvd), ptr(center), ptr(minv), ptr(cube), ctypes.byref(cam),
                                              ptr(f32(g)), I(B), I(N), I(img_size), I(normalise), ptr(out),
                                              stream_ptr()), "dsf_uvd_to_xyz_backward")
        return out, None, None, None, None, None, None


class XyzToUvd(Function):
    @staticmethod
    def forward(ctx, xyz, center, M, cube, cam, img_size, world):
        xyz, center, M, cube = f32(xyz), f32(center), f32(M), f32(cube)
        B, N, _ = xyz.shape
        out = _empty(xyz.shape, xyz)
        check(L.lib().dsf_xyz_to_uvd(ptr(xyz), ptr(center), ptr(M), ptr(cube), ctypes.byref(cam), I(B), I(N),
                                     I(img_size), I(int(world)), ptr(out), stream_ptr()), "dsf_xyz_to_uvd")
        ctx.save_for_backward(xyz, center, M, cube)
        ctx.args = (cam, img_size, int(world))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xyz, center, M, cube = ctx.saved_tensors
        cam, img_size, world = ctx.args
        B, N, _ = xyz.shape
        out = _empty(xyz.shape, xyz)
        check(L.lib().dsf_xyz_to_uvd_backward(ptr(xyz), ptr(center), ptr(M), ptr(cube), ctypes.byref(cam), ptr(f32(g)),
                                              I(B), I(N), I(img_size), I(world), ptr(out), stream_ptr()),
              "dsf_xyz_to_uvd_backward")
        return out, None, None, None, None, None, None


class CropHand(Function):
    """crop_hand (data/render_loader.py:1209-1227); also returns the normalised point image."""

    @staticmethod
    def forward(ctx, img, joints_nl, center, minv, cube, cam, offsetxy, offsetz, thickness):
        img, joints_nl, center, minv, cube = f32(img), f32(joints_nl), f32(center), f32(minv), f32(cube)
        B, _, S, _ = img.shape
        out = _empty(img.shape, img)
        xyz = _empty((B, S * S, 3), img)
        keep = _empty((B, 1, S, S), img, torch.uint8)
        check(L.lib().dsf_crop_hand(ptr(img), ptr(joints_nl), ptr(center), ptr(minv), ptr(cube), ctypes.byref(cam),
                                    I(B), I(joints_nl.shape[1]), I(S), F(offsetxy), F(offsetz), F(thickness), ptr(out),
                                    ptr(xyz), ptr(keep), stream_ptr()), "dsf_crop_hand")
        ctx.save_for_backward(keep)
        ctx.mark_non_differentiable(xyz, keep)
        return out, xyz, keep

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _gx, _gk):
        keep, = ctx.saved_tensors
        return (g * keep.to(g.dtype),) + (None,) * 8


def img2pcl(img, center, minv, cube, cam, n_sample, rand_keys=None):
    """Img2pcl (data/render_loader.py:1121-1156) with the random draw as an explicit input."""
    img, center, minv, cube = f32(img), f32(center), f32(minv), f32(cube)
    B, _, S, _ = img.shape
    if rand_keys is None:
        rand_keys = torch.randint(0, 2 ** 31 - 1, (B, S * S), device=img.device, dtype=torch.int32)
    rand_keys = rand_keys.contiguous()
    pcl = _empty((B, n_sample, 3), img)
    counts = _empty((B,), img, torch.int32)
    ws = _empty((B, 2 * S * S), img, torch.int32)
    check(L.lib().dsf_img2pcl(ptr(img), ptr(center), ptr(minv), ptr(cube), ctypes.byref(cam), ptr(rand_keys), I(B), I(S),
                              I(n_sample), ptr(pcl), ptr(counts), ptr(ws), stream_ptr()), "dsf_img2pcl")
    return pcl, counts


def depth_crop_normalize(depth, com, cube, paras, dsize=128, want_raw=False):
    """Test-phase data path of loader.__getitem__ (data/render_loader.py:1909-1916): Crop_Image_deep_pp + normalize_img on
    raw depth frames.  depth (B,Hd,Wd) mm -- uint16 / int16 (the sensors' raw frames, read as they are) or anything float32 can
    hold --, com (B,3) (u, v, z) and cube (B,3) / (3,) are used as float64 (the reference's
    numpy arithmetic) -> img (B,1,dsize,dsize) f32, trans (B,3,3) f64 [, raw crop (B,dsize,dsize)]."""
    if not depth.is_cuda:
        raise RuntimeError("dsf_amd ops run on the GPU only (got a %s tensor); there is no CPU path" % depth.device)
    # raw sensor frames stay 16-bit (torch.uint16, or int16 holding the same bits); anything else is read as float32
    raw16 = depth.dtype in (torch.uint16, torch.int16)
    depth = depth.contiguous() if raw16 else depth.float().contiguous()
    B, Hd, Wd = depth.shape
    com = torch.as_tensor(com, dtype=torch.float64, device=depth.device).reshape(B, 3).contiguous()
    cube = torch.as_tensor(cube, dtype=torch.float64, device=depth.device)
    cube = (cube.reshape(1, 3).expand(B, 3) if cube.numel() == 3 else cube.reshape(B, 3)).contiguous()
    img = _empty((B, 1, dsize, dsize), depth)
    trans = torch.empty((B, 3, 3), device=depth.device, dtype=torch.float64)
    raw = _empty((B, dsize, dsize), depth) if want_raw else None
    D = ctypes.c_double
    fn = L.lib().dsf_depth_crop_normalize_u16 if raw16 else L.lib().dsf_depth_crop_normalize
    check(fn(ptr(depth), ptr(com), ptr(cube), D(paras[0]), D(paras[1]), I(B), I(Hd), I(Wd),
             I(dsize), ptr(img), ptr(trans), ptr(raw), stream_ptr()), "dsf_depth_crop_normalize")
    return (img, trans, raw) if want_raw else (img, trans)


def depth_augment_crop(crop, joints, com, cube, M, mode, off, rot, sc, paras, flip=1):
    """Training-phase ``augmentCrop`` (data/render_loader.py:653-695) for a batch of raw crops on the device.
    crop (B,S,S) f32 (un-normalised), joints (B,J,3) f32, com / cube (B,3), M (B,3,3), mode (B,) ints into
    ['rot','com','sc','none'], off (B,3) mm, rot (B,) degrees, sc (B,) -> img (B,1,S,S), joints, cube, com, M (f64)."""
    if not crop.is_cuda:
        raise RuntimeError("dsf_amd ops run on the GPU only (got a %s tensor); there is no CPU path" % crop.device)
    dev = crop.device
    crop = crop.float().contiguous()
    B, S, _ = crop.shape
    joints = joints.to(dev).float().contiguous()
    J = joints.shape[1]
    d64 = lambda t, shape: torch.as_tensor(t, dtype=torch.float64, device=dev).reshape(shape).contiguous()
    com, cube, M, off, rot, sc = d64(com, (B, 3)), d64(cube, (B, 3)), d64(M, (B, 3, 3)), d64(off, (B, 3)), d64(rot, (B,)), d64(sc, (B,))
    mode = torch.as_tensor(mode, dtype=torch.int32, device=dev).reshape(B).contiguous()
    img = _empty((B, 1, S, S), crop)
    jout = _empty((B, J, 3), crop)
    cube_o, com_o = torch.empty((B, 3), device=dev, dtype=torch.float64), torch.empty((B, 3), device=dev, dtype=torch.float64)
    M_o = torch.empty((B, 3, 3), device=dev, dtype=torch.float64)
    D = ctypes.c_double
    check(L.lib().dsf_depth_augment_crop(ptr(crop), ptr(joints), ptr(com), ptr(cube), ptr(M), ptr(mode), ptr(off), ptr(rot), ptr(sc),
                                         D(paras[0]), D(paras[1]), D(paras[2]), D(paras[3]), I(int(flip)), I(B), I(S), I(J), ptr(img),
                                         ptr(jout), ptr(cube_o), ptr(com_o), ptr(M_o), stream_ptr()), "dsf_depth_augment_crop")
    return img, jout, cube_o, com_o, M_o


def _map_strides(t):
    """(batch, channel, pixel) strides of a (B,C,S,S) tensor whose pixels are uniformly strided (NCHW, channels-last,
    or a channel slice of either) as a ctypes int64[3]; None if the layout is anything else."""
    sb, sc, sy, sx = t.stride()
    if sy != t.shape[-1] * sx:
        return None
    return (ctypes.c_int64 * 3)(sb, sc, sx)


class Joint2Offset(Function):
    """GFM.joint2offset (util/generateFeature.py:14-37).  ``channels_last``: the (B,4J,S,S) result is written
    with channels-last strides (same values; it then concatenates / compares with convolution outputs copy-free)."""

    @staticmethod
    def forward(ctx, joints, img, kernel_size, S, channels_last=False):
        joints, img = f32(joints), f32(img)
        B = img.shape[0]
        joints = joints.reshape(B, -1, 3) if B else joints.reshape(0, joints.shape[-2] if joints.dim() >= 2 else 0, 3)
        J, H = joints.shape[1], img.shape[-1]
        maps = torch.empty((B, 4 * J, S, S), device=img.device, dtype=torch.float32,
                           memory_format=torch.channels_last if channels_last else torch.contiguous_format)
        check(L.lib().dsf_joint2offset_forward(ptr(joints), ptr(img), I(B), I(J), I(H), I(S), F(kernel_size),
                                               L.addr(maps), _map_strides(maps), stream_ptr()),
              "dsf_joint2offset_forward")
        ctx.save_for_backward(joints, img)
        ctx.args = (kernel_size, S)
        return maps

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        joints, img = ctx.saved_tensors
        ks, S = ctx.args
        B, J, _ = joints.shape
        gj = _empty(joints.shape, joints)
        if g.dtype != torch.float32:
            g = g.float()
        st = _map_strides(g)
        if st is None:
            g = g.contiguous()
            st = _map_strides(g)
        check(L.lib().dsf_joint2offset_backward(ptr(joints), ptr(img), L.addr(g), I(B), I(J),
                                                I(img.shape[-1]), I(S), F(ks), ptr(gj), st, stream_ptr()),
              "dsf_joint2offset_backward")
        return gj, None, None, None, None


class HuberMean(Function):
    """SmoothL1Loss with delta (metric/losses.py:6-30) as one fused reduction + one fused backward; gradient w.r.t.
    ``x`` only.  x and y are walked in memory order, so they must share one dense layout (arranged by the caller)."""

    @staticmethod
    def forward(ctx, x, y, delta, scale):
        assert x.shape == y.shape and x.stride() == y.stride()
        n = x.numel()
        loss = torch.empty((), device=x.device, dtype=torch.float32)
        ws = torch.empty(1024, device=x.device, dtype=torch.float32) if n > 4096 else None
        check(L.lib().dsf_huber_mean_forward(L.addr(x), L.addr(y), I64(n), F(delta),
                                             F(scale), ptr(loss), ptr(ws), stream_ptr()), "dsf_huber_mean_forward")
        ctx.save_for_backward(x, y)
        ctx.args = (delta, scale)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        delta, scale = ctx.args
        gx = torch.empty_like(x)                      # preserve_format: same dense strides as x
        assert gx.stride() == x.stride()
        check(L.lib().dsf_huber_mean_backward(L.addr(x), L.addr(y),
                                              ptr(f32(g)), I64(x.numel()), F(delta), F(scale),
                                              L.addr(gx), stream_ptr()), "dsf_huber_mean_backward")
        return gx, None, None, None


def _dense(t):
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


def huber_mean(x, y, delta=0.01, size_average=True, weight=1.0):
    """mean_{rows}(mean_{last dim} h(x - y)) (or the sum over rows) on the fused kernels; falls back to None when the
    inputs are not plain fp32 GPU tensors or y needs a gradient (caller then uses the composite formula)."""
    if not (x.is_cuda and y.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32) or y.requires_grad:
        return None
    if x.numel() == 0:
        return None
    if not _dense(x):
        x = x.contiguous()
    if y.stride() != x.stride():
        y = y.contiguous(memory_format=torch.channels_last) if (x.dim() == 4 and not x.is_contiguous()) else y.contiguous()
        if y.stride() != x.stride():              # size-1 dims can make strides ambiguous: settle on plain contiguous
            x, y = x.contiguous(), y.contiguous()
    scale = (1.0 / x.numel() if size_average else 1.0 / x.shape[-1]) * weight
    return HuberMean.apply(x, y, float(delta), float(scale))


DECODE_CL = [os.environ.get("DSF_DECODE_CL", "1") == "1"]      # 0: channels-last maps through the (sample, joint) kernels too (A/B aid)


class Offset2Joint(Function):
    """GFM.offset2joint_softmax (util/generateFeature.py:39-59).  The maps are read in the layout they come in (NCHW or the
    network's channels-last) and their gradient is written in that layout: no layout copy either way."""

    @staticmethod
    def forward(ctx, maps, depth, kernel_size, scale):
        maps, depth = (maps if maps.dtype == torch.float32 else maps.float()), f32(depth)
        st = _map_strides(maps)
        if st is None or st[0] != maps[0].numel():      # an exotic layout: settle on plain contiguous
            maps = maps.contiguous()
            st = _map_strides(maps)
        B, C, S, _ = maps.shape
        J, H = C // 4, depth.shape[-1]
        joints = _empty((B, J, 3), maps)
        stats = _empty((B, J, 2), maps)
        # a dense channels-last map (what the heads write): pixel-chunk workgroups that read a pixel's record contiguously
        # (dsf_offset2joint_*_cl); any other uniformly strided layout: the (sample, joint) kernels
        ctx.cl = bool(DECODE_CL[0] and B > 0 and J <= 32 and C == 4 * J and tuple(st) == (C * S * S, 1, C))
        if ctx.cl:
            ws = _empty((int(L.lib().dsf_offset2joint_cl_workspace_floats(I(B), I(S))),), maps)
            check(L.lib().dsf_offset2joint_forward_cl(L.addr(maps), ptr(depth), I(B), I(J), I(H), I(S), F(kernel_size), F(scale), ptr(joints),
                                                      ptr(stats), ptr(ws), stream_ptr()), "dsf_offset2joint_forward_cl")
        else:
            check(L.lib().dsf_offset2joint_forward_strided(L.addr(maps), st, ptr(depth), I(B), I(J), I(H), I(S), F(kernel_size), F(scale),
                                                           ptr(joints), ptr(stats), stream_ptr()), "dsf_offset2joint_forward_strided")
        ctx.save_for_backward(maps, depth, joints, stats)
        ctx.args = (kernel_size, scale)
        return joints

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        maps, depth, joints, stats = ctx.saved_tensors
        ks, scale = ctx.args
        B, C, S, _ = maps.shape
        gm = torch.empty_like(maps)                       # preserve_format: the maps' own dense layout
        assert gm.stride() == maps.stride()
        if ctx.cl:
            check(L.lib().dsf_offset2joint_backward_cl(L.addr(maps), ptr(depth), ptr(joints), ptr(stats), ptr(f32(g)), I(B), I(C // 4),
                                                       I(depth.shape[-1]), I(S), F(ks), F(scale), L.addr(gm), stream_ptr()),
                  "dsf_offset2joint_backward_cl")
            return gm, None, None, None
        check(L.lib().dsf_offset2joint_backward_strided(L.addr(maps), _map_strides(maps), ptr(depth), ptr(joints), ptr(stats), ptr(f32(g)), I(B),
                                                        I(C // 4), I(depth.shape[-1]), I(S), F(ks), F(scale), L.addr(gm),
                                                        stream_ptr()), "dsf_offset2joint_backward_strided")
        return gm, None, None, None


# --------------------------------------------------------------------------------------------
# Loss-side glue of the trainer steps as fused launches (csrc/step_ops.hip; include/dsf_hip.h "Loss-side glue")
# --------------------------------------------------------------------------------------------
class M2dLoss(Function):
    """``m2d_loss`` (train_render.py:728-732) -> (loss, sums (B,4), per (B,)); gradient w.r.t. ``synth`` only."""

    @staticmethod
    def forward(ctx, real, synth, thresh, scale):
        B = real.shape[0]
        P = real.numel() // max(B, 1)
        sums = _empty((B, 4), real)
        per = _empty((B,), real)
        loss = torch.zeros((), device=real.device, dtype=torch.float32) if B == 0 else _empty((), real)
        check(L.lib().dsf_m2d_forward(ptr(real), ptr(synth), I(B), I(P), F(thresh), F(scale), ptr(sums), ptr(per), ptr(loss), stream_ptr()),
              "dsf_m2d_forward")
        ctx.save_for_backward(real, synth, sums)
        ctx.args = (thresh, scale)
        ctx.mark_non_differentiable(sums, per)
        return loss, sums, per

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _gs, _gp):
        real, synth, sums = ctx.saved_tensors
        thresh, scale = ctx.args
        B = real.shape[0]
        gs = torch.empty_like(synth)
        check(L.lib().dsf_m2d_backward(ptr(real), ptr(synth), ptr(sums), ptr(f32(g)), I(B), I(real.numel() // max(B, 1)), F(thresh), F(scale),
                                       ptr(gs), stream_ptr()), "dsf_m2d_backward")
        return None, gs, None, None


def m2d(real, synth, thresh=0.99, scale=0.1):
    """-> (loss, sums, per) of the fused model-to-data term, or None when the fused path does not apply (``real`` needs a
    gradient, layouts differ, not fp32)."""
    if not (real.is_cuda and real.dtype == torch.float32 and synth.dtype == torch.float32 and real.shape == synth.shape and
            real.is_contiguous() and synth.is_contiguous() and not real.requires_grad and real.dim() >= 2):
        return None
    return M2dLoss.apply(real, synth, float(thresh), float(scale))


class CubePoints(Function):
    """(verts_n, joints_n, center, cube) -> (verts_world, joints_world, verts_norm, joints_norm): Render.render's
    ``p * cube / 2 + center`` and ``(p - center) / cube * 2`` (mano_layer.py:1078-1092); gradients w.r.t. the points only."""

    @staticmethod
    def forward(ctx, verts, joints, center, cube):
        verts, joints, center, cube = f32(verts), f32(joints), f32(center), f32(cube)
        B, NV, NJ = verts.shape[0], verts.shape[1], joints.shape[1]
        vw, vn = torch.empty_like(verts), torch.empty_like(verts)
        jw, jn = torch.empty_like(joints), torch.empty_like(joints)
        check(L.lib().dsf_cube_points_forward(ptr(verts), ptr(joints), ptr(center), ptr(cube), I(B), I(NV), I(NJ), ptr(vw), ptr(jw), ptr(vn), ptr(jn),
                                              stream_ptr()), "dsf_cube_points_forward")
        ctx.save_for_backward(cube)
        ctx.dims = (B, NV, NJ)
        ctx.set_materialize_grads(False)
        return vw, jw, vn, jn

    @staticmethod
    @once_differentiable
    def backward(ctx, gvw, gjw, gvn, gjn):
        cube, = ctx.saved_tensors
        B, NV, NJ = ctx.dims
        c = lambda t: None if t is None else f32(t)
        gvw, gjw, gvn, gjn = c(gvw), c(gjw), c(gvn), c(gjn)
        gv = _empty((B, NV, 3), cube)
        gj = _empty((B, NJ, 3), cube)
        check(L.lib().dsf_cube_points_backward(ptr(gvw), ptr(gjw), ptr(gvn), ptr(gjn), ptr(cube), I(B), I(NV), I(NJ), ptr(gv), ptr(gj), stream_ptr()),
              "dsf_cube_points_backward")
        return gv, gj, None, None


def view_rotate(verts, joints, center, rot=None, recentre=False):
    """RotationPoints (mano_layer.py:874-884) in one launch, inference only -> (verts, joints) rotated about ``center`` by ``rot``
    ((B,3) axis-angle or (B,4) quaternion; None: no rotation); ``recentre``: the points are first moved so that the mean of the
    joints sits at ``center`` (Render.forward :995-1003)."""
    verts, joints, center = f32(verts), f32(joints), f32(center)
    rot = f32(rot) if rot is not None else None
    ov, oj = torch.empty_like(verts), torch.empty_like(joints)
    check(L.lib().dsf_view_rotate(ptr(verts), ptr(joints), ptr(center), ptr(rot), I(rot.shape[-1] if rot is not None else 0), I(int(recentre)),
                                  I(verts.shape[0]), I(verts.shape[1]), I(joints.shape[1]), ptr(ov), ptr(oj), stream_ptr()), "dsf_view_rotate")
    return ov, oj


def cube_normalise(verts, joints, center, cube):
    """((verts - center) / cube * 2, the same for the joints) in one launch, inference only (Render.forward :1033-1034)"""
    verts, joints, center, cube = f32(verts), f32(joints), f32(center), f32(cube)
    vn, jn = torch.empty_like(verts), torch.empty_like(joints)
    check(L.lib().dsf_cube_normalise(ptr(verts), ptr(joints), ptr(center), ptr(cube), I(verts.shape[0]), I(verts.shape[1]), I(joints.shape[1]),
                                     ptr(vn), ptr(jn), stream_ptr()), "dsf_cube_normalise")
    return vn, jn


class M2P(Function):
    """the M2P term (train_render.py:590-603 / 787-801) in one launch each way: juvd_pix, juvd_mano (B,21,3), sample_ok (B) bool,
    part_dist (B,15) -> scalar (weight folded in); gradient w.r.t. juvd_pix only."""

    @staticmethod
    def forward(ctx, juvd_pix, juvd_mano, sample_ok, part_dist, weight):
        a, b = f32(juvd_pix), f32(juvd_mano)
        ok = sample_ok.to(torch.uint8).contiguous()
        pd = f32(part_dist)
        B = a.shape[0]
        out, aux = _empty((), a), _empty((2,), a)
        check(L.lib().dsf_m2p_forward(ptr(a), ptr(b), ptr(ok), ptr(pd), I(B), F(1e-3), F(0.01), F(weight), ptr(out), ptr(aux), stream_ptr()),
              "dsf_m2p_forward")
        ctx.save_for_backward(a, b, ok, pd, aux)
        ctx.weight = weight
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, b, ok, pd, aux = ctx.saved_tensors
        ga = torch.empty_like(a)
        check(L.lib().dsf_m2p_backward(ptr(a), ptr(b), ptr(ok), ptr(pd), ptr(aux), ptr(f32(g)), I(a.shape[0]), F(1e-3), F(0.01), F(ctx.weight),
                                       ptr(ga), stream_ptr()), "dsf_m2p_backward")
        return ga, None, None, None, None


class PartMean(Function):
    """per-part masked mean of point distances (metric/meshLoss.py:389-394): dis (B,P), seg (B,P) int64 labels -> (B,n_parts)"""

    @staticmethod
    def forward(ctx, dis, seg, n_parts):
        dis = f32(dis)
        seg = seg.contiguous()
        B, P = dis.shape
        out = _empty((B, n_parts), dis)
        valid = _empty((B, n_parts), dis)
        check(L.lib().dsf_part_mean_forward(ptr(dis), ptr(seg), I(B), I(P), I(n_parts), ptr(out), ptr(valid), stream_ptr()), "dsf_part_mean_forward")
        ctx.save_for_backward(seg, valid)
        ctx.n = n_parts
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        seg, valid = ctx.saved_tensors
        B, P = seg.shape
        gd = _empty((B, P), valid)
        check(L.lib().dsf_part_mean_backward(ptr(f32(g)), ptr(seg), ptr(valid), I(B), I(P), I(ctx.n), ptr(gd), stream_ptr()), "dsf_part_mean_backward")
        return gd, None, None


class ManoReg(Function):
    """the two MANO regularisers of Pretrain (train_render.py:463-464) on the packed rows (B,W) -> (2,) [beta term, scale term]"""

    @staticmethod
    def forward(ctx, paras, beta_col, scale_col, w_beta, w_scale):
        paras = f32(paras)
        B, W = paras.shape
        out = _empty((2,), paras)
        check(L.lib().dsf_mano_reg_forward(ptr(paras), I(B), I(W), I(beta_col), I(scale_col), F(w_beta), F(w_scale), ptr(out), stream_ptr()),
              "dsf_mano_reg_forward")
        ctx.save_for_backward(paras)
        ctx.args = (beta_col, scale_col, w_beta, w_scale)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        paras, = ctx.saved_tensors
        B, W = paras.shape
        bc, sc, wb, ws_ = ctx.args
        gp = torch.empty_like(paras)
        check(L.lib().dsf_mano_reg_backward(ptr(paras), ptr(f32(g)), I(B), I(W), I(bc), I(sc), F(wb), F(ws_), ptr(gp), stream_ptr()),
              "dsf_mano_reg_backward")
        return gp, None, None, None, None


class PoolLinear(Function):
    """``Linear(AdaptiveAvgPool2d(1)(x).flatten(1))`` (model/backbone.py:225-226: the MANO regression head) on a channels-last feature
    map: one launch forward, two backward (csrc/step_ops.hip); the input gradient comes back channels-last."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous(memory_format=torch.channels_last)
        w = f32(weight)
        b = f32(bias) if bias is not None else None
        B, C, H, W_ = x.shape
        O = w.shape[0]
        pooled = _empty((B, C), x)
        out = _empty((B, O), x)
        check(L.lib().dsf_pool_linear_forward(L.addr(x), ptr(w), ptr(b), I(B), I(H * W_), I(C), I(O), ptr(pooled), ptr(out), stream_ptr()),
              "dsf_pool_linear_forward")
        ctx.save_for_backward(pooled, w)
        ctx.dims = (B, C, H, W_, bias is not None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        pooled, w = ctx.saved_tensors
        B, C, H, W_, has_bias = ctx.dims
        O = w.shape[0]
        g = f32(g)
        gx = torch.empty((B, C, H, W_), device=g.device, dtype=torch.float32, memory_format=torch.channels_last) if ctx.needs_input_grad[0] else None
        gw = _empty((O, C), g) if ctx.needs_input_grad[1] else None
        gb = _empty((O,), g) if (has_bias and ctx.needs_input_grad[2] and gw is not None) else None
        check(L.lib().dsf_pool_linear_backward(ptr(g), ptr(pooled), ptr(w), I(B), I(H * W_), I(C), I(O), L.addr(gx) if gx is not None else None,
                                               ptr(gw), ptr(gb), stream_ptr()), "dsf_pool_linear_backward")
        if has_bias and ctx.needs_input_grad[2] and gb is None:
            gb = g.sum(0)
        return gx, gw, gb


class CatChannels(Function):
    """``torch.cat(maps, dim=1)`` of up to four channels-last fp32 maps in one launch (csrc/step_ops.hip: dsf_cat_channels_nhwc; the
    stage-2 input of reference model/backbone.py:256); the gradients are the channel slices of the incoming one, as torch's"""

    @staticmethod
    def forward(ctx, *maps):
        maps = [m.contiguous(memory_format=torch.channels_last) for m in maps]
        B, _, H, W_ = maps[0].shape
        cs = [m.shape[1] for m in maps]
        out = torch.empty((B, sum(cs), H, W_), device=maps[0].device, dtype=torch.float32, memory_format=torch.channels_last)
        a = [(L.addr(m), I(c)) for m, c in zip(maps, cs)] + [(None, I(0))] * (4 - len(maps))
        check(L.lib().dsf_cat_channels_nhwc(a[0][0], a[0][1], a[1][0], a[1][1], a[2][0], a[2][1], a[3][0], a[3][1], L.addr(out),
                                            ctypes.c_int64(B * H * W_), stream_ptr()), "dsf_cat_channels_nhwc")
        ctx.cs = cs
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        outs, o = [], 0
        for i, c in enumerate(ctx.cs):
            outs.append(g.narrow(1, o, c) if ctx.needs_input_grad[i] else None)
            o += c
        return tuple(outs)


CAT_FUSED = [os.environ.get("DSF_CAT", "1") == "1"]


def cat_channels(maps):
    """torch.cat(maps, dim=1); one launch for 2-4 channels-last fp32 GPU maps whose channel counts are multiples of 4 (DSF_CAT=0: torch's)"""
    m0 = maps[0]
    if (CAT_FUSED[0] and 2 <= len(maps) <= 4 and all(m.is_cuda and m.dim() == 4 and m.dtype == torch.float32 and m.shape[1] % 4 == 0 and
                                                      m.shape[0] == m0.shape[0] and m.shape[2:] == m0.shape[2:] for m in maps)
            and m0.numel() and sum(m.numel() for m in maps) // 4 < 2 ** 32):
        return CatChannels.apply(*maps)
    return torch.cat(tuple(maps), dim=1)


def pool_linear(x, linear):
    """the fused head when it applies (GPU, fp32, C <= 2048, <= 64 outputs), else None"""
    if x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[1] <= 2048 and linear.out_features <= 64 and x.shape[0] > 0 \
            and linear.weight.dtype == torch.float32:
        return PoolLinear.apply(x, linear.weight, linear.bias)
    return None
