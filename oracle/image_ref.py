"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by dsf_amd/).

CPU fp32 restatement of the image-side pieces of the hot path: crop-matrix
construction, the two nearest-neighbour index maps (resize 640->480 rows and
the per-sample crop warp), depth normalisation, joint uvd transforms, the
``loader`` tensor utilities, GFM offset maps and the Huber loss.

Index-producing code is numpy float32 with one IEEE op per expression so that
the HIP kernels can match it bit for bit; differentiable pieces are torch-CPU so
autograd provides the gradient oracle.  Pinned against the reference by
tests/test_oracle_golden.py (golden vectors made by importing the reference).
"""
import numpy as np
import torch

F = np.float32
CAM_NYU = (588.03, 587.07, 320.0, 240.0)


# --------------------------------------------------------------------------------------
# crop matrix (Render.points3DToImg / comToBounds / Offset2Trans, mano_layer.py:1318-1324,
# 1133-1169)
# --------------------------------------------------------------------------------------
def project_points(xyz, cam=CAM_NYU):
    """points3DToImg, mano_layer.py:1318-1324 (eps only on the u denominator)."""
    fx, fy, fu, fv = (F(c) for c in cam)
    xyz = np.asarray(xyz, dtype=F)
    u = xyz[..., 0] * fx / (xyz[..., 2] + F(1e-8)) + fu
    v = xyz[..., 1] * fy / xyz[..., 2] + fv
    return np.stack([u, v, xyz[..., 2]], -1).astype(F)


def crop_bounds(center2d, cube, cam=CAM_NYU):
    """comToBounds, mano_layer.py:1133-1141 -> int32 (xs, xe, ys, ye), f32 (zs, ze)."""
    fx, fy = F(cam[0]), F(cam[1])
    c = np.asarray(center2d, dtype=F)
    s = np.asarray(cube, dtype=F)
    half = F(2.0)
    xs = np.floor((c[:, 0] * c[:, 2] / fx - s[:, 0] / half) / c[:, 2] * fx + F(0.5)).astype(np.int32)
    xe = np.floor((c[:, 0] * c[:, 2] / fx + s[:, 0] / half) / c[:, 2] * fx + F(0.5)).astype(np.int32)
    ys = np.floor((c[:, 1] * c[:, 2] / fy - s[:, 1] / half) / c[:, 2] * fy + F(0.5)).astype(np.int32)
    ye = np.floor((c[:, 1] * c[:, 2] / fy + s[:, 1] / half) / c[:, 2] * fy + F(0.5)).astype(np.int32)
    return xs, xe, ys, ye, c[:, 2] - s[:, 2] / half, c[:, 2] + s[:, 2] / half


def crop_matrix(xs, xe, ys, ye, crop=128):
    """Offset2Trans, mano_layer.py:1143-1169: M = off . scale . trans (B,3,3) f32."""
    wb = (xe - xs).astype(np.int32)
    hb = (ye - ys).astype(np.int32)
    wide = wb > hb
    with np.errstate(divide="ignore", invalid="ignore"):
        sz0 = np.where(wide, crop, ((wb * crop).astype(F) / hb.astype(F)).astype(np.int32))
        sz1 = np.where(wide, ((hb * crop).astype(F) / wb.astype(F)).astype(np.int32), crop)
        s = np.where(wide, F(crop) / wb.astype(F), F(crop) / hb.astype(F)).astype(F)
    ox = np.floor(F(crop / 2.0) - sz0.astype(F) / F(2.0)).astype(np.int32).astype(F)
    oy = np.floor(F(crop / 2.0) - sz1.astype(F) / F(2.0)).astype(np.int32).astype(F)
    B = wb.shape[0]
    M = np.zeros((B, 3, 3), dtype=F)
    M[:, 0, 0] = s
    M[:, 1, 1] = s
    M[:, 2, 2] = 1
    M[:, 0, 2] = s * (-xs).astype(F) + ox
    M[:, 1, 2] = s * (-ys).astype(F) + oy
    return M


# --------------------------------------------------------------------------------------
# nearest index maps (Render.resize :1233-1242, affine_grid/warpPerspective :1244-1260)
# --------------------------------------------------------------------------------------
def warp_source_index(Minv, crop=128, w_src=640, h_src=480):
    """For every crop pixel the flat index (row*w_src+col) of the 480x640 pixel
    that grid_sample(nearest, align_corners=False, zeros) reads, -1 = padding.
    ``Minv`` is the reference's ``torch.inverse(M)`` (an explicit input: its low
    bits are LAPACK noise that decides exact .5 ties)."""
    Minv = np.asarray(Minv, dtype=F)
    x = np.arange(crop, dtype=F)[None, None, :]
    y = np.arange(crop, dtype=F)[None, :, None]
    m = lambda r, c: Minv[:, r, c][:, None, None]
    sx = (m(0, 0) * x + m(0, 1) * y) + m(0, 2)            # torch CPU matmul: mul, mul, add, add
    sy = (m(1, 0) * x + m(1, 1) * y) + m(1, 2)
    gx = (sx / F(w_src)) * F(2) - F(1)
    gy = (sy / F(h_src)) * F(2) - F(1)
    ix = np.rint((gx + F(1)) * F(w_src / 2.0) - F(0.5))
    iy = np.rint((gy + F(1)) * F(h_src / 2.0) - F(0.5))
    ok = (ix >= 0) & (ix < w_src) & (iy >= 0) & (iy < h_src)
    return np.where(ok, iy * w_src + ix, -1).astype(np.int32)


def normalize_depth(img, center_z, cube_z):
    """normalize_img, mano_layer.py:1289-1299 (img: (B,...) metric depth)."""
    img = np.asarray(img, dtype=F)
    shp = (-1,) + (1,) * (img.ndim - 1)
    cz = np.asarray(center_z, dtype=F).reshape(shp)
    half = (np.asarray(cube_z, dtype=F) / F(2.0)).reshape(shp)
    zmin, zmax = cz - half, cz + half
    out = np.where((img == -1) | (img == 0), zmax, img)
    out = np.where(out > zmax, zmax, out)
    out = np.where(out < zmin, zmin, out)
    return ((out - cz) / half).astype(F)


def joint_trans(xyz, M, center2d, cube, cam=CAM_NYU, crop=128):
    """JointTrans, mano_layer.py:1301-1309."""
    uvd = project_points(xyz, cam)
    M = np.asarray(M, dtype=F)
    u = (M[:, None, 0, 0] * uvd[..., 0] + M[:, None, 0, 1] * uvd[..., 1]) + M[:, None, 0, 2]
    v = (M[:, None, 1, 0] * uvd[..., 0] + M[:, None, 1, 1] * uvd[..., 1]) + M[:, None, 1, 2]
    d = (uvd[..., 2] - np.asarray(center2d, dtype=F)[:, None, 2]) / (np.asarray(cube, dtype=F)[:, None, 2] / F(2.0))
    return np.stack([u / F(crop) * F(2) - F(1), v / F(crop) * F(2) - F(1), d], -1).astype(F)


# --------------------------------------------------------------------------------------
# loader tensor utilities (data/render_loader.py:1044-1088, 1113-1118, 1190-1227)
# --------------------------------------------------------------------------------------
def uvd_to_xyz(uvd, center, Minv, cube, cam=CAM_NYU, img_size=128, normalise=True):
    """uvd_nl2xyznl_tensor / uvd_nl2xyz_tensor (render_loader.py:1044-1073), flip=1."""
    fx, fy, fu, fv = (F(c) for c in cam)
    uvd = np.asarray(uvd, dtype=F)
    center = np.asarray(center, dtype=F)[:, None, :]
    cube = np.asarray(cube, dtype=F)[:, None, :]
    Minv = np.asarray(Minv, dtype=F)
    uu = (uvd[..., 0] + F(1)) * F(img_size / 2)
    vv = (uvd[..., 1] + F(1)) * F(img_size / 2)
    d = uvd[..., 2] * (cube[..., 2] / F(2.0)) + center[..., 2]
    u = (Minv[:, None, 0, 0] * uu + Minv[:, None, 0, 1] * vv) + Minv[:, None, 0, 2]
    v = (Minv[:, None, 1, 0] * uu + Minv[:, None, 1, 1] * vv) + Minv[:, None, 1, 2]
    xyz = np.stack([(u - fu) * d / fx, (v - fv) * d / fy, d], -1).astype(F)
    if normalise:
        xyz = (xyz - center) / (cube / F(2.0))
    return xyz.astype(F)


def xyz_to_uvd(xyz_nl, center, M, cube, cam=CAM_NYU, img_size=128):
    """xyz_nl2uvdnl_tensor (render_loader.py:1075-1088)."""
    center = np.asarray(center, dtype=F)[:, None, :]
    cube = np.asarray(cube, dtype=F)[:, None, :]
    w = np.asarray(xyz_nl, dtype=F) * cube / F(2.0) + center
    uvd = project_points(w, cam)
    M = np.asarray(M, dtype=F)
    u = (M[:, None, 0, 0] * uvd[..., 0] + M[:, None, 0, 1] * uvd[..., 1]) + M[:, None, 0, 2]
    v = (M[:, None, 1, 0] * uvd[..., 0] + M[:, None, 1, 1] * uvd[..., 1]) + M[:, None, 1, 2]
    d = (uvd[..., 2] - center[..., 2]) / (cube[..., 2] / F(2))
    return np.stack([u / F(img_size) * F(2.0) - F(1), v / F(img_size) * F(2.0) - F(1), d], -1).astype(F)


def _pixel_grid_aligned(S):
    """2i/(S-1)-1 grid of uvdImg2xyzImg / Img2pcl (render_loader.py:1126-1128, 1194-1196);
    channel 0 varies along columns, channel 1 along rows."""
    g = (F(2.0) * np.arange(S, dtype=F) / F(S - 1.0) - F(1.0)).astype(F)
    return np.broadcast_to(g[None, :], (S, S)), np.broadcast_to(g[:, None], (S, S))


def depth_image_to_xyz(img, center, Minv, cube, cam=CAM_NYU):
    """uvdImg2xyzImg (render_loader.py:1190-1201): returns (xyz_mm, xyz_normalised), each (B,3,S,S)."""
    img = np.asarray(img, dtype=F)
    B, _, S, _ = img.shape
    gu, gv = _pixel_grid_aligned(S)
    uvd = np.stack([np.broadcast_to(gu, (B, S, S)), np.broadcast_to(gv, (B, S, S)), img[:, 0]], -1).reshape(B, -1, 3)
    a = uvd_to_xyz(uvd, center, Minv, cube, cam, S, normalise=False)
    b = uvd_to_xyz(uvd, center, Minv, cube, cam, S, normalise=True)
    back = lambda t: t.reshape(B, S, S, 3).transpose(0, 3, 1, 2)
    return back(a), back(b)


def crop_hand_keep(img, joints_nl, center, Minv, cube, cam=CAM_NYU, offsetxy=25.0, offsetz=20.0, thick=20.0):
    """Pixel mask of crop_hand (render_loader.py:1209-1226): inside the joints' bounding box grown by the offsets."""
    img = np.asarray(img, dtype=F)
    sk = np.asarray(joints_nl, dtype=F) * np.asarray(cube, dtype=F)[:, None, :] / F(2) + np.asarray(center, dtype=F)[:, None, :]
    lo = sk.min(1)
    hi = sk.max(1)
    lo = lo - np.array([offsetxy, offsetxy, offsetz], dtype=F)
    hi = hi + np.array([offsetxy, offsetxy, offsetz], dtype=F)
    lo[:, 2] = lo[:, 2] - F(thick)
    xyz, _ = depth_image_to_xyz(img, center, Minv, cube, cam)
    keep = np.ones(img[:, 0].shape, dtype=bool)
    for a in range(3):
        keep &= (xyz[:, a] > lo[:, a, None, None]) & (xyz[:, a] < hi[:, a, None, None])
    return keep[:, None]


def crop_hand(img, joints_nl, center, Minv, cube, cam=CAM_NYU, offsetxy=25.0, offsetz=20.0, thick=20.0):
    """crop_hand (render_loader.py:1209-1227)."""
    img = np.asarray(img, dtype=F)
    keep = crop_hand_keep(img, joints_nl, center, Minv, cube, cam, offsetxy, offsetz, thick)
    return np.where(keep, img, F(1.0)).astype(F)


def image_to_points_candidates(img, center, Minv, cube, cam=CAM_NYU):
    """Deterministic part of Img2pcl (render_loader.py:1121-1139): per sample the
    list of valid (img <= 0.99) pixels in scan order, converted to normalised xyz."""
    img = np.asarray(img, dtype=F)
    B, _, S, _ = img.shape
    gu, gv = _pixel_grid_aligned(S)
    out = []
    for b in range(B):
        m = img[b, 0] <= F(0.99)
        uvd = np.stack([gu[m], gv[m], img[b, 0][m]], -1)[None]
        out.append(uvd_to_xyz(uvd, center[b:b + 1], Minv[b:b + 1], cube[b:b + 1], cam, 128)[0] if m.any()
                   else np.zeros((0, 3), dtype=F))
    return out


# --------------------------------------------------------------------------------------
# GFM offset maps (util/generateFeature.py:14-59 == model/backbone.py:45-91), Huber loss
# --------------------------------------------------------------------------------------
def _centre_grid(S):
    g = 2.0 * (torch.arange(S).float() + 0.5) / S - 1.0
    return g.view(1, 1, 1, S).expand(1, 1, S, S), g.view(1, 1, S, 1).expand(1, 1, S, S)   # u (cols), v (rows)


def joints_to_offset_maps(joints, img, kernel=0.8, S=64):
    """joint2offset (generateFeature.py:14-37). joints (B,J,3) torch, img (B,1,H,W) -> (B,4J,S,S)."""
    B, J, _ = joints.shape
    dep = torch.nn.functional.interpolate(img, size=[S, S])
    gu, gv = _centre_grid(S)
    coords = torch.cat([gu.expand(B, 1, S, S), gv.expand(B, 1, S, S), dep], 1)          # (B,3,S,S)
    off = joints.view(B, J, 3, 1, 1) - coords.unsqueeze(1)
    dist = torch.sqrt((off * off).sum(2) + 1e-8)
    unit = off / dist.unsqueeze(2)
    heat = (kernel - dist) / kernel
    mask = (heat >= 0).float() * (dep < 0.99).float()
    return torch.cat([(unit * mask.unsqueeze(2)).view(B, 3 * J, S, S), heat * mask], 1)


def offset_maps_to_joints(maps, depth, kernel=0.8, scale=30):
    """offset2joint_softmax (generateFeature.py:39-59)."""
    B, C, S, _ = maps.shape
    J = C // 4
    if depth.shape[-1] != S:
        depth = torch.nn.functional.interpolate(depth, size=[S, S])
    gu, gv = _centre_grid(S)
    coords = torch.cat([gu.expand(B, 1, S, S), gv.expand(B, 1, S, S), depth], 1).view(B, 1, 3, -1)
    mask = (depth < 0.99).float().view(B, 1, 1, -1)
    unit = maps[:, :3 * J].reshape(B, J, 3, -1) * mask
    heat = maps[:, 3 * J:].reshape(B, J, -1) * mask.view(B, 1, -1)
    wgt = torch.softmax(heat * scale, dim=-1)
    dist = kernel - heat * kernel
    return ((unit * dist.unsqueeze(2) + coords) * wgt.unsqueeze(2)).sum(-1)


def huber(x, y):
    """SmoothL1Loss(size_average=True), metric/losses.py:6-30 (delta 0.01)."""
    z = (x - y).float()
    a = z.abs()
    per = torch.where(a < 0.01, 0.5 * z * z, 0.01 * (a - 0.005))
    return per.mean(-1).mean()


def masked_depth_l1(real, synth):
    """depth_loss(smooth=False), render_model/render_loss.py:9-21."""
    m = (real < 0.99) & (synth < 0.99)
    return (real - synth)[m].abs().mean()
