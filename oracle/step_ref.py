"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by dsf_amd/).

CPU restatement of one whole training step of BASELINE config 2 (the reference's
``Pretrain`` loss list, train_render.py:444-466, plus ``Render.render`` + the m2d depth term,
:719-732), composed from the oracle pieces: hand_ref (MANO, autograd), p3d_ref.c (640x640
rasteriser forward / backward), image_ref (crop chain, GFM, Huber).  Used (a) as the end-to-end
loss / gradient checker of the HIP step and (b) as bench.py's ``cpu_baseline`` leg (kind "port").

The convolutional trunk is plain torch.nn (device-agnostic plumbing shared with the product);
every geometry op on this path is the oracle's, on the host cores.
"""
import numpy as np
import torch

from . import hand_ref as H
from . import image_ref as I
from . import p3d

CAM = (588.03, 587.07, 320.0, 240.0)


def resize_rowmap():
    """640 -> 480 row table of Render.resize from torch's own ops (mano_layer.py:1233-1242)."""
    import torch.nn.functional as Fn
    idx = torch.arange(640, dtype=torch.float32).view(1, 1, 640, 1).expand(1, 1, 640, 640).contiguous()
    grid = Fn.affine_grid(torch.tensor([[[1.0, 0, 0], [0, 1.0, 0]]]), (1, 1, 480, 640), align_corners=False)
    return Fn.grid_sample(idx, grid, mode="nearest", align_corners=False)[0, 0, :, 0].numpy().astype(np.int64)


class _RasterCrop(torch.autograd.Function):
    """face_verts (B*F,3,3) NDC -> normalised crop (B,1,128,128): A.2 forward, A.3 backward."""

    @staticmethod
    def forward(ctx, fv, B, nf, Minv, rowmap, center_z, cube_z):
        fvn = fv.detach().numpy()
        first, cnt = np.arange(B) * nf, np.full(B, nf)
        p2f, zbuf, _, _ = p3d.rasterize_meshes(fvn, first, cnt, 640, want_bary=False)
        depth = np.where(zbuf <= 0, 0, zbuf).astype(np.float32)
        src = I.warp_source_index(Minv)                                       # (B,128,128) flat idx into 480x640
        ok = src >= 0
        s = np.maximum(src, 0)
        ry, rx = rowmap[s // 640], s % 640
        bidx = np.arange(B)[:, None, None]
        crop = np.where(ok, depth[bidx, ry, rx], 0).astype(np.float32)
        img = I.normalize_depth(crop[:, None], center_z, cube_z)
        half = (cube_z / 2.0).astype(np.float32)
        zmin, zmax = center_z - half, center_z + half
        live = ok & (crop > 0) & (crop <= zmax[:, None, None]) & (crop >= zmin[:, None, None])
        ctx.saved = (fvn, p2f, ry, rx, live, half, B)
        return torch.from_numpy(img)

    @staticmethod
    def backward(ctx, g):
        fvn, p2f, ry, rx, live, half, B = ctx.saved
        g = g.numpy()[:, 0]
        gz = np.zeros((B, 640, 640), dtype=np.float32)
        for b in range(B):
            ii, jj = np.nonzero(live[b])
            np.add.at(gz[b], (ry[b, ii, jj], rx[b, ii, jj]), g[b, ii, jj] / half[b])
        return torch.from_numpy(p3d.rasterize_backward_zbuf(fvn, p2f, gz)), None, None, None, None, None, None


class OracleRender:
    def __init__(self, mano_dict):
        self.hm = H.HandModel(mano_dict)
        self.rowmap = resize_rowmap()
        self.faces = self.hm.faces
        self.depth_range = [500, 1200]

    def get_mesh_xyz(self, p):
        v, j = H.mano_vertices(self.hm, p[:, :3], p[:, 3:48], p[:, 48:58], p[:, 58:62], 1 / 125)
        return j, v

    def _crop_geometry(self, center, cube):
        cn, cb = center.detach().numpy(), cube.detach().numpy()
        c2 = I.project_points(cn)
        xs, xe, ys, ye, _, _ = I.crop_bounds(c2, cb)
        M = I.crop_matrix(xs, xe, ys, ye)
        return c2, cb, M, torch.inverse(torch.from_numpy(M)).numpy()

    def _depth_crop(self, vw, center, cube):
        """world verts (B,779,3) -> normalised crop: rasterise 640^2, bg 0, resize, warp, normalise (mano_layer.py:1082-1092)."""
        B = vw.shape[0]
        c2, cb, M, Minv = self._crop_geometry(center, cube)
        hw, hh = 320.0, 240.0                                                       # A.1 (px' = py' = 0 for NYU)
        X, Y, Z = vw.unbind(-1)
        pv = torch.stack([(-X * np.float32(CAM[0] / hw)) / Z, (-Y * np.float32(CAM[1] / hh)) / Z, Z], -1)
        fv = pv[:, self.faces].reshape(-1, 3, 3)
        img = _RasterCrop.apply(fv, B, self.faces.shape[0], Minv, self.rowmap, c2[:, 2], cb[:, 2])
        return img, torch.from_numpy(M)

    @staticmethod
    def joint_trans(pw, Mt, center, cube):
        """JointTrans (mano_layer.py:1301-1309), differentiable torch restatement; pw world mm."""
        u = pw[..., 0] * CAM[0] / (pw[..., 2] + 1e-8) + CAM[2]
        w = pw[..., 1] * CAM[1] / pw[..., 2] + CAM[3]
        uu = (Mt[:, None, 0, 0] * u + Mt[:, None, 0, 1] * w) + Mt[:, None, 0, 2]
        vv = (Mt[:, None, 1, 0] * u + Mt[:, None, 1, 1] * w) + Mt[:, None, 1, 2]
        d = (pw[..., 2] - center[:, None, 2]) / (cube[:, None, 2] / 2.0)
        return torch.stack([uu / 128 * 2 - 1, vv / 128 * 2 - 1, d], -1)

    def render(self, p, center, cube):
        """Render.render (mano_layer.py:1071-1097) -> img, joint_uvd, joint_xyz, mesh_xyz."""
        j, v = self.get_mesh_xyz(p)
        vw = v * cube.unsqueeze(1) / 2 + center.unsqueeze(1)
        jw = j * cube.unsqueeze(1) / 2 + center.unsqueeze(1)
        img, Mt = self._depth_crop(vw, center, cube)
        return img, self.joint_trans(jw, Mt, center, cube), j, v

    def forward(self, p, center3d, cube, aug_view=None, aug_shape=None, aug_center=None, aug_size=None, mask_draws=None):
        """Render.forward (mano_layer.py:983-1039) -> the 8-tuple; ``mask_draws`` = (joint ids, uvd offsets, radii) of
        mask_img (:1326-1340) as explicit inputs."""
        beta = p[:, 48:58] + aug_shape if aug_shape is not None else p[:, 48:58]
        v, j = H.mano_vertices(self.hm, p[:, :3], p[:, 3:48], beta, p[:, 58:62])                 # :1001 (mm, no global scale)
        c = j.mean(dim=1, keepdim=True)                                                            # :1002-1004
        v, j = v - c, j - c
        v, j = v + center3d.unsqueeze(1), j + center3d.unsqueeze(1)                                # :1010-1011
        if aug_view is not None:                                                                   # RotationPoints :874-884
            Rm = H.rodrigues(aug_view).unsqueeze(1)
            c3 = center3d.unsqueeze(1)
            v = torch.matmul(Rm, (v - c3).unsqueeze(-1)).squeeze(-1) + c3
            j = torch.matmul(Rm, (j - c3).unsqueeze(-1)).squeeze(-1) + c3
        if aug_center is not None:
            center3d = center3d + aug_center
        if aug_size is not None:
            cube = cube * aug_size
        img, Mt = self._depth_crop(v, center3d, cube)
        juvd = self.joint_trans(j, Mt, center3d, cube)
        vuvd = self.joint_trans(v, Mt, center3d, cube)
        jxyz = (j - center3d.unsqueeze(1)) / cube.unsqueeze(1) * 2
        vxyz = (v - center3d.unsqueeze(1)) / cube.unsqueeze(1) * 2
        if mask_draws is not None:
            img = mask_image(img, juvd, *mask_draws)
        return img, juvd, vuvd, jxyz, vxyz, center3d, cube, Mt


def mask_image(img, juvd, joint_id, offset, radius):
    """mask_img (mano_layer.py:1326-1340) with its draws given: occluding spheres in (u, v, d) space -> background 1."""
    B = img.shape[0]
    S = img.shape[-1]
    g = 2 * (torch.arange(S).float() + 0.5) / S - 1.0
    xx, yy = torch.meshgrid(g, g, indexing="xy")                                                    # :968-971
    pix = torch.cat((torch.stack((xx, yy), -1).reshape(1, -1, 2).expand(B, -1, -1), img.reshape(B, -1, 1)), -1)
    centre = juvd[:, joint_id, :] + offset
    dis = torch.sqrt(((pix.unsqueeze(1) - centre.unsqueeze(2)) ** 2).sum(-1))
    hit = (dis < radius.unsqueeze(-1)).float().sum(1).gt(0)
    return torch.where(hit.view(B, 1, S, S), torch.ones_like(img), img)


class _PointFace(torch.autograd.Function):
    """_PointFaceDistance (metric/meshLoss.py:21-70) on the C restatement of pytorch3d's op (Appendix A.4)."""

    @staticmethod
    def forward(ctx, points, pfirst, tris, tfirst):
        d, i = p3d.point_face_dist_forward(points.detach().numpy(), pfirst, tris.detach().numpy(), tfirst)
        ctx.saved = (points.detach().numpy(), tris.detach().numpy(), i)
        return torch.from_numpy(d)

    @staticmethod
    def backward(ctx, g):
        pts, tris, i = ctx.saved
        gp, gt = p3d.point_face_dist_backward(pts, tris, i, g.contiguous().numpy())
        return torch.from_numpy(gp), None, torch.from_numpy(gt), None


def icp_loss(mesh, pcl, faces):
    """ICPLoss (meshLoss.py:347-353): mean squared point-to-mesh distance per sample."""
    B, P, _ = pcl.shape
    F_ = faces.shape[0]
    tris = mesh[:, faces.long()].reshape(-1, 3, 3)
    d = _PointFace.apply(pcl.reshape(-1, 3), np.arange(B) * P, tris, np.arange(B) * F_)
    return d.view(B, P).mean(-1)


def joint_icp_loss(mesh, pcl, faces_list, seg):
    """JointICPLoss (meshLoss.py:377-395) in the reference's own form: every point against every part, then the
    ``seg == part`` selection and the mean over the selected points with a positive distance."""
    B, P, _ = pcl.shape
    out = []
    for k, fk in enumerate(faces_list):
        fk = fk.long()
        tris = mesh[:, fk].reshape(-1, 3, 3)
        d = _PointFace.apply(pcl.reshape(-1, 3), np.arange(B) * P, tris, np.arange(B) * fk.shape[0]).view(B, P)
        d = torch.where(seg.eq(k + 1), d, torch.zeros_like(d))
        valid = d.gt(0).sum(-1)
        loss = d.sum(-1) / (valid + 1e-8)
        out.append(torch.where(valid.eq(0), torch.zeros_like(loss), loss))
    return torch.stack(out, dim=-1)


def net_forward(net, img, render, center, cube):
    """MANO_OCR_stage.forward (model/backbone.py:284-323) with the oracle's bridge."""
    c0 = net.pre(img)
    _, feat, pix, mano = net._run_trunk(c0, '')
    if not net.refine:
        return [[pix, mano]]
    mano_img, mano_uvd, _, _ = render.render(mano, center, cube)
    remap = I.joints_to_offset_maps(mano_uvd, mano_img, 0.8, 64)
    # (one dtype and device for the concatenation: the tests also run this with a float64 trunk, or with the torch twin on
    #  the GPU, around the fp32 CPU geometry)
    _, _, pix2, mano2 = net._run_trunk(net.fusion(torch.cat([t.to(device=c0.device, dtype=c0.dtype) for t in (c0, feat, pix, remap)], dim=1)), '_s2')
    return [[pix, mano], [pix2, mano2]]


def m2d(real, synth):
    """train_render.py:728-732."""
    union = ((real < 0.99) | (synth < 0.99)).float()
    per = ((real - synth).abs() * union).sum(-1).sum(-1) / (union.sum(-1).sum(-1) + 1e-8)
    return per.mean() * 0.1


def make_targets(render, params, center, cube, noise=None):
    with torch.no_grad():
        img, juvd, jxyz, mesh = render.render(params, center, cube)
        if noise is not None:
            img = torch.where(img < 0.99, (img + noise).clamp(-1, 0.98), img)
    return {"img": img, "joint_uvd": juvd, "joint_xyz": jxyz, "mesh_xyz": mesh, "center": center, "cube": cube}


def step_loss(net, render, tgt, coord_weight=100.0, deconv_weight=1.0, model_weight=1.0):
    img, center, cube = tgt["img"], tgt["center"], tgt["cube"]
    outs = net_forward(net, img, render, center, cube)
    total = 0
    for pix, mano in outs:
        pix_gt = I.joints_to_offset_maps(tgt["joint_uvd"], img, 0.8, pix.shape[-1])
        juvd = I.offset_maps_to_joints(pix, img, 0.8)
        jxyz, mesh = render.get_mesh_xyz(mano)
        total = total + I.huber(pix, pix_gt) * deconv_weight + I.huber(juvd, tgt["joint_uvd"]) * coord_weight \
            + I.huber(jxyz, tgt["joint_xyz"]) * coord_weight + I.huber(mesh, tgt["mesh_xyz"]) * coord_weight \
            + torch.mean(mano[:, 48:58] ** 2) * coord_weight * 10 + torch.mean(torch.clamp(mano[:, 58], max=0.0).abs()) * 0.1
    img_pd, _, _, _ = render.render(outs[-1][1], center, cube)
    return total + m2d(img, img_pd) * model_weight


def timed_steps(mano_dict, B=2, steps=3, warmup=1, backbone="ResNet_stage_18", seed=0):
    """cpu_baseline leg: full step (fwd + bwd + AdamW) on the host cores; returns (images/s, seconds, n)."""
    import time
    from dsf_amd.model.backbone import MANO_OCR_stage
    from . import nets
    from dsf_amd.train_step import synthetic_batch
    torch.manual_seed(seed)
    net = nets.build(MANO_OCR_stage, backbone, 21, True)         # plain torch.nn twin of the product net
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=0.01)
    render = OracleRender(mano_dict)
    p, c, cube = synthetic_batch(B, "cpu", seed)
    tgt = make_targets(render, p, c, cube)
    t0 = None
    for it in range(warmup + steps):
        if it == warmup:
            t0 = time.perf_counter()
        opt.zero_grad()
        loss = step_loss(net, render, tgt)
        loss.backward()
        opt.step()
    dt = time.perf_counter() - t0
    return B * steps / dt, dt, B * steps


# ------------------------------------------------------------------------------------------------
# the other trainer steps (BASELINE configs 3, 4, 5): CPU compositions with every draw an explicit input
# ------------------------------------------------------------------------------------------------
def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def crop_hand(img, joints_nl, center, Minv, cube):
    """loader.crop_hand (render_loader.py:1209-1227) -> (cropped image, carrying torch.where's gradient; keep mask)."""
    keep = _t(I.crop_hand_keep(img.detach().numpy(), joints_nl.detach().numpy(), center.numpy(), Minv, cube.numpy()))
    return torch.where(keep, img, torch.ones_like(img)), keep


def image_to_points(img, center, Minv, cube, keys, n=2048):
    """loader.Img2pcl (render_loader.py:1121-1156) with the multinomial draw replaced by injected keys: the
    valid pixels (img <= 0.99) in scan order, tiled floor(n / count) times, plus the ``n - count * mult`` pixels with the
    smallest keys (ties -> scan order), listed in scan order; zeros for an empty image."""
    imgn = img.detach().numpy()
    cand = I.image_to_points_candidates(imgn, center.numpy(), Minv, cube.numpy())
    keys = np.asarray(keys)
    B = imgn.shape[0]
    out = np.zeros((B, n, 3), dtype=np.float32)
    for b in range(B):
        valid = np.nonzero(imgn[b, 0].reshape(-1) <= np.float32(0.99))[0]
        m = valid.shape[0]
        if m == 0:
            continue
        mult = n // m
        rem = n - mult * m
        pick = np.sort(np.lexsort((np.arange(m), keys[b, valid].astype(np.uint32)))[:rem])
        out[b] = np.concatenate([cand[b]] * mult + [cand[b][pick]])
    return _t(out)


def depth_image_points(img, center, Minv, cube):
    """normalised xyz of every pixel (uvdImg2xyzImg, render_loader.py:1190-1201) as (B, S*S, 3)."""
    _, nl = I.depth_image_to_xyz(img.detach().numpy(), center.numpy(), Minv, cube.numpy())
    B = nl.shape[0]
    return _t(nl.reshape(B, 3, -1).transpose(0, 2, 1))


def real_targets(hm, img_src, crop_r, jxyz_pix, jxyz_mano, mesh_mano, center, Minv, cube, d):
    """Part labels + the two point clouds of the real image (train_render.py:561-575 / 693-701)."""
    B = crop_r.shape[0]
    pts = depth_image_points(crop_r, center, Minv, cube)
    seg_img = H.segment_points(hm, jxyz_pix.detach(), jxyz_mano.detach(), mesh_mano.detach(), pts)
    seg_img = torch.where(crop_r.detach().lt(0.99).reshape(B, -1), seg_img, torch.zeros_like(seg_img)).reshape(B, 1, 128, 128)
    joint_img = torch.where(seg_img.gt(0), img_src.detach(), torch.ones_like(img_src))
    joint_pcl = image_to_points(joint_img, center, Minv, cube, d["keys_joint"].numpy())
    segment = H.segment_points(hm, jxyz_pix.detach(), jxyz_mano.detach(), mesh_mano.detach(), joint_pcl)
    pcl = image_to_points(crop_r, center, Minv, cube, d["keys_pcl"].numpy())
    return joint_pcl, segment, pcl


def m2p_loss(juvd_pix, juvd_mano, mano_ok, pd2m_j, coord_weight):
    """M2P in the reference's literal form (train_render.py:590-603 / 787-801): nonzero() + index_select and the host
    branch on the SUM of the selected indices."""
    B = juvd_pix.shape[0]
    jm = pd2m_j.lt(1e-3)
    jm = torch.cat((torch.ones(B, 1), jm.float(), jm[:, [2, 5, 8, 11, 14]].float()), dim=-1).gt(0)
    rows = (mano_ok.unsqueeze(-1) & jm).detach().view(-1).gt(0).nonzero().squeeze()
    a = torch.index_select(juvd_mano.reshape(-1, 3), 0, rows.reshape(-1))
    b = torch.index_select(juvd_pix.reshape(-1, 3), 0, rows.reshape(-1))
    if rows.sum() == 0:
        return torch.zeros(())
    return I.huber(b, a.detach()) * coord_weight


def synth_pass(render, transfer, p, cube, d, mask=True):
    """RenderNet(model_para, None, cube, augment...) + frozen transfer generator (train_render.py:428-435 / 633-639)."""
    with torch.no_grad():
        md = (d["mask_joint_id"], d["mask_offset"], d["mask_radius"]) if (mask and "mask_joint_id" in d) else None
        img, juvd, _, jxyz, vxyz, center, cube_s, M = render.forward(p, d["center0"], cube, d["aug_view"], d["aug_shape"],
                                                                    d["aug_center"], d["aug_size"], md)
        img_t = transfer(img) if transfer is not None else img
    return {"img": img, "img_t": img_t, "joint_uvd": juvd, "joint_xyz": jxyz, "mesh_xyz": vxyz, "center": center, "cube": cube_s}


def supervised_terms(render, outs, s, cfg, coll=False, beta_scale=False, first_only=False, hm=None):
    """Per-stage pixel-branch + MANO-branch supervised losses shared by Pretrain (:444-466), Finetune (:512-527) and
    FinetuneStage (:645-667)."""
    total = 0
    for pix, mano in (outs[:1] if first_only else outs):
        pix_gt = I.joints_to_offset_maps(s["joint_uvd"], s["img"], 0.8, pix.shape[-1])
        juvd = I.offset_maps_to_joints(pix, s["img"], 0.8)
        jxyz, mesh = render.get_mesh_xyz(mano)
        total = total + I.huber(pix, pix_gt) * cfg.deconv_weight + I.huber(juvd, s["joint_uvd"]) * cfg.coord_weight \
            + I.huber(jxyz, s["joint_xyz"]) * cfg.coord_weight + I.huber(mesh, s["mesh_xyz"]) * cfg.coord_weight
        if coll:
            total = total + H.collision_loss(hm, jxyz, mesh.detach()) * cfg.coll_weight
        if beta_scale:
            total = total + torch.mean(mano[:, 48:58] ** 2) * cfg.coord_weight * 10 \
                + torch.mean(torch.clamp(mano[:, 58], max=0.0).abs()) * 0.1
    return total


def pretrain_loss(net, render, transfer, p, cube, d, cfg, views=1, mask=True):
    """Trainer.Pretrain (train_render.py:415-488); views > 1 = BASELINE config 4 (every sample from `views` rotations)."""
    if views > 1:
        p, cube = p.repeat_interleave(views, dim=0), cube.repeat_interleave(views, dim=0)
    s = synth_pass(render, transfer, p, cube, d, mask)
    outs = net_forward(net, s["img_t"], render, s["center"], s["cube"])
    return supervised_terms(render, outs, s, cfg, beta_scale=True)


def finetune_loss(net, render, transfer, p, cube, img_r, center_r, cube_r, d, cfg, mask=True):
    """Trainer.Finetune (train_render.py:490-620)."""
    hm = render.hm
    s = synth_pass(render, transfer, p, cube, d, mask)
    outs = net_forward(net, s["img_t"], render, s["center"], s["cube"])
    total = supervised_terms(render, outs, s, cfg, coll=True, first_only=True, hm=hm)
    _, _, M_r, Minv = render._crop_geometry(center_r, cube_r)
    pix_r, mano_r = net_forward(net, img_r, render, center_r, cube_r)[0]
    juvd_r = I.offset_maps_to_joints(pix_r, img_r, 0.8)
    jxyz_r = _t(I.uvd_to_xyz(juvd_r.detach().numpy(), center_r.numpy(), Minv, cube_r.numpy()))
    img_m, mjuvd, mjxyz, mesh = render.render(mano_r, center_r, cube_r)
    coll = H.collision_loss(hm, mjxyz, mesh.detach())
    crop_r, _ = crop_hand(img_r, mjxyz, center_r, Minv, cube_r)
    crop_m, _ = crop_hand(img_m, mjxyz, center_r, Minv, cube_r)
    union = (crop_r.lt(0.99) | crop_m.lt(0.99)).float()
    m2d_ = (torch.abs(crop_r - crop_m).mean(-1).mean(-1) / (union.mean(-1).mean(-1) + 1e-8)).mean()
    joint_pcl, segment, pcl = real_targets(hm, img_r, crop_r, jxyz_r, mjxyz, mesh, center_r, Minv, cube_r, d)
    pd2m_j = joint_icp_loss(mesh, joint_pcl, hm.joint_faces, segment)
    d2m_b = icp_loss(mesh, pcl, hm.faces)
    p2m = I.huber(mjuvd, juvd_r.detach()) * cfg.coord_weight
    both = (crop_r.lt(0.95) & img_m.lt(0.95)).float()
    depth_b = (torch.abs(crop_r - img_m) * both).sum(-1).sum(-1) / both.sum(-1).sum(-1)
    mano_ok = depth_b.lt(0.04).squeeze(-1) & d2m_b.lt(1e-3)
    m2p = m2p_loss(juvd_r, mjuvd, mano_ok, pd2m_j, cfg.coord_weight)
    terms = {"m2d": m2d_, "pd2m": pd2m_j.mean(-1).mean(-1), "P2M": p2m, "coll": coll, "M2P": m2p, "d2m": d2m_b.mean(-1)}
    total = total + p2m + m2d_ * 0.1 * cfg.model_weight + terms["d2m"] * cfg.model_weight + terms["pd2m"] * cfg.partICP_weight \
        + m2p * cfg.M2P_weight + coll * cfg.coll_weight
    return total, terms


def finetune_stage_loss(net, render, transfer, p, cube, img_r, center_r, cube_r, d, cfg, mask=True):
    """Trainer.FinetuneStage (train_render.py:622-823): loss list, detach points (:684, :688-689), thresholds 0.99 / 0.04 /
    1e-3 and joint_add (:791) as in the reference."""
    hm = render.hm
    s = synth_pass(render, transfer, p, cube, d, mask)
    outs = net_forward(net, s["img_t"], render, s["center"], s["cube"])
    total = supervised_terms(render, outs, s, cfg, coll=True, hm=hm)
    _, _, M_r, Minv = render._crop_geometry(center_r, cube_r)
    outs = net_forward(net, img_r, render, center_r, cube_r)
    pix_t, mano_t = outs[1][0].detach(), outs[1][1].detach()
    with torch.no_grad():
        juvd_t = I.offset_maps_to_joints(pix_t, img_r, 0.8)
        jxyz_t = _t(I.uvd_to_xyz(juvd_t.numpy(), center_r.numpy(), Minv, cube_r.numpy()))
        mj_t, mm_t = render.get_mesh_xyz(mano_t)
        crop_r, _ = crop_hand(img_r, mj_t, center_r, Minv, cube_r)
        joint_pcl, segment, pcl = real_targets(hm, crop_r, crop_r, jxyz_t, mj_t, mm_t, center_r, Minv, cube_r, d)
    # stage 1 student (:706-749)
    pix1, mano1 = outs[0]
    juvd1 = I.offset_maps_to_joints(pix1, img_r, 0.8)
    total = total + I.huber(pix1, pix_t) * cfg.deconv_weight + I.huber(juvd1, juvd_t) * cfg.coord_weight
    img1, mjuvd1, mjxyz1, mesh1 = render.render(mano1, center_r, cube_r)
    total = total + I.huber(mjxyz1, jxyz_t) * cfg.coord_weight + I.huber(mesh1, mm_t) * cfg.coord_weight
    total = total + H.collision_loss(hm, mjxyz1, mesh1.detach()) * cfg.coll_weight
    crop1, _ = crop_hand(img1, mj_t, center_r, Minv, cube_r)
    total = total + m2d(crop_r, crop1) * cfg.model_weight
    total = total + icp_loss(mesh1, pcl, hm.faces).mean(-1) * cfg.model_weight
    total = total + joint_icp_loss(mesh1, joint_pcl, hm.joint_faces, segment).mean(-1).mean(-1) * cfg.partICP_weight
    # stage 2 (:752-808)
    pix2, mano2 = outs[1]
    juvd2 = I.offset_maps_to_joints(pix2, img_r, 0.8)
    img2, mjuvd2, mjxyz2, mesh2 = render.render(mano2, center_r, cube_r)
    p2m = I.huber(mjuvd2, juvd_t) * cfg.coord_weight
    coll2 = H.collision_loss(hm, mjxyz2, mesh2.detach())
    crop2, _ = crop_hand(img2, mj_t, center_r, Minv, cube_r)
    union = (crop_r.lt(0.99) | crop2.lt(0.99)).float()
    m2d2 = m2d(crop_r, crop2)
    pd2m_j = joint_icp_loss(mesh2, joint_pcl, hm.joint_faces, segment)
    d2m_b = icp_loss(mesh2, pcl, hm.faces)
    both = (crop_r.lt(0.99) & crop2.lt(0.99)).float()
    depth_b = ((crop_r - crop2).abs() * both).sum(-1).sum(-1) / (union.sum(-1).sum(-1) + 1e-8)
    mano_ok = depth_b.lt(0.04).squeeze(-1) & d2m_b.lt(1e-3)
    m2p = m2p_loss(juvd2, mjuvd2, mano_ok, pd2m_j, cfg.coord_weight)
    total = total + p2m + coll2 * cfg.coll_weight + m2d2 * cfg.model_weight + d2m_b.mean(-1) * cfg.model_weight \
        + pd2m_j.mean(-1).mean(-1) * cfg.partICP_weight + m2p * cfg.M2P_weight
    terms = {"P2M": p2m, "m2d": m2d2, "d2m": d2m_b.mean(-1), "pd2m": pd2m_j.mean(-1).mean(-1), "M2P": m2p, "coll": coll2}
    return total, terms


def mesh_targets(render, p, center, cube, keys_joint, keys_pcl):
    """Targets of the config-3 step (dsf_amd.train_step.MeshLossStep.make_targets): render of the ground-truth
    parameters, its hand crop, part labels and the two sampled point clouds."""
    hm = render.hm
    with torch.no_grad():
        img, juvd, jxyz, mesh = render.render(p, center, cube)
        _, _, M, Minv = render._crop_geometry(center, cube)
        crop, _ = crop_hand(img, jxyz, center, Minv, cube)
        d = {"keys_joint": keys_joint, "keys_pcl": keys_pcl}
        joint_pcl, seg, pcl = real_targets(hm, crop, crop, jxyz, jxyz, mesh, center, Minv, cube, d)
    return {"img": img, "crop": crop, "joint_xyz": jxyz, "mesh_xyz": mesh, "joint_pcl": joint_pcl, "seg": seg, "pcl": pcl,
            "center": center, "cube": cube, "M": _t(M), "Minv": Minv}


def mesh_step_loss(net, render, tgt, cfg):
    """BASELINE config 3 (MeshLossStep.loss): hourglass + MANO head, m2d + part ICP + ICP + collision + supervised terms."""
    hm = render.hm
    _, mano_pd = net(tgt["img"])
    img_pd, juvd, jxyz, mesh = render.render(mano_pd, tgt["center"], tgt["cube"])
    crop_pd, _ = crop_hand(img_pd, tgt["joint_xyz"], tgt["center"], tgt["Minv"], tgt["cube"])
    l_m2d = m2d(tgt["crop"], crop_pd) * cfg.model_weight
    l_part = joint_icp_loss(mesh, tgt["joint_pcl"], hm.joint_faces, tgt["seg"]).mean(-1).mean(-1) * cfg.partICP_weight
    l_icp = icp_loss(mesh, tgt["pcl"], hm.faces).mean(-1) * cfg.model_weight
    l_coll = H.collision_loss(hm, jxyz, mesh.detach()) * cfg.coll_weight
    l_sup = (I.huber(jxyz, tgt["joint_xyz"]) + I.huber(mesh, tgt["mesh_xyz"])) * cfg.coord_weight
    terms = {"m2d": l_m2d, "pd2m": l_part, "d2m": l_icp, "coll": l_coll, "sup": l_sup}
    return l_m2d + l_part + l_icp + l_coll + l_sup, terms
