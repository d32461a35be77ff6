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

    def get_mesh_xyz(self, p):
        v, j = H.mano_vertices(self.hm, p[:, :3], p[:, 3:48], p[:, 48:58], p[:, 58:62], 1 / 125)
        return j, v

    def render(self, p, center, cube):
        """Render.render (mano_layer.py:1071-1097) -> img, joint_uvd, joint_xyz, mesh_xyz."""
        B = p.shape[0]
        j, v = self.get_mesh_xyz(p)
        vw = v * cube.unsqueeze(1) / 2 + center.unsqueeze(1)
        jw = j * cube.unsqueeze(1) / 2 + center.unsqueeze(1)
        cn, cb = center.numpy(), cube.numpy()
        c2 = I.project_points(cn)
        xs, xe, ys, ye, _, _ = I.crop_bounds(c2, cb)
        M = I.crop_matrix(xs, xe, ys, ye)
        Minv = torch.inverse(torch.from_numpy(M)).numpy()
        hw, hh = 320.0, 240.0                                                       # A.1 (px' = py' = 0 for NYU)
        X, Y, Z = vw.unbind(-1)
        pv = torch.stack([(-X * np.float32(CAM[0] / hw)) / Z, (-Y * np.float32(CAM[1] / hh)) / Z, Z], -1)
        fv = pv[:, self.faces].reshape(-1, 3, 3)
        img = _RasterCrop.apply(fv, B, self.faces.shape[0], Minv, self.rowmap, c2[:, 2], cb[:, 2])
        # JointTrans (mano_layer.py:1301-1309), differentiable torch restatement
        Mt = torch.from_numpy(M)
        u = jw[..., 0] * CAM[0] / (jw[..., 2] + 1e-8) + CAM[2]
        w = jw[..., 1] * CAM[1] / jw[..., 2] + CAM[3]
        uu = (Mt[:, None, 0, 0] * u + Mt[:, None, 0, 1] * w) + Mt[:, None, 0, 2]
        vv = (Mt[:, None, 1, 0] * u + Mt[:, None, 1, 1] * w) + Mt[:, None, 1, 2]
        d = (jw[..., 2] - center[:, None, 2]) / (cube[:, None, 2] / 2.0)
        juvd = torch.stack([uu / 128 * 2 - 1, vv / 128 * 2 - 1, d], -1)
        return img, juvd, j, v


def net_forward(net, img, render, center, cube):
    """MANO_OCR_stage.forward (model/backbone.py:284-323) with the oracle's bridge."""
    c0 = net.pre(img)
    _, feat, pix, mano = net._run_trunk(c0, '')
    if not net.refine:
        return [[pix, mano]]
    mano_img, mano_uvd, _, _ = render.render(mano, center, cube)
    remap = I.joints_to_offset_maps(mano_uvd, mano_img, 0.8, 64)
    _, _, pix2, mano2 = net._run_trunk(net.fusion(torch.cat((c0, feat, pix, remap), dim=1)), '_s2')
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
