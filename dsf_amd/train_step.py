"""Per-batch training steps of the hot path (counterparts of ``Trainer.Pretrain`` /
``Trainer.FinetuneStage``, /root/reference/train_render.py:415-488, 622-823) with every
host synchronisation removed: no ``.item()``, no ``.cpu()``, no data-dependent Python branch,
so a whole step can be enqueued asynchronously (and captured in a HIP graph).

Loss lists, weights, detach points and thresholds follow the reference; each block cites its
lines.  TensorBoard / cv2 drawing / ``xyz2error`` host metrics are out of scope.
"""
import math

import numpy as np
import torch

from . import ops
from .metric.losses import SmoothL1Loss
from .metric.meshLoss import ICPLoss, JointICPLoss
from .render_model.render_loss import m2d_loss
from .util.generateFeature import GFM
from .data.render_loader import loader as TensorUtils


class Config:
    """Loss weights / feature settings of /root/reference/config.py:31-98."""
    stage_num = 2
    deconv_weight = 1
    coord_weight = 100
    model_weight = 1
    partICP_weight = 1
    M2P_weight = 1
    coll_weight = 1
    feature_type = ['offset']
    feature_para = [0.8]
    batch_size = 32
    input_size = 128
    lr = 0.001
    weight_decay = 0.01
    cube_size = [250, 250, 250]
    net = 'ResNet_stage_18'


def synthetic_batch(B, device, seed=0, dtype=torch.float32):
    """Synthetic NYU-shape inputs of SURVEY.md 8(d): 62-d MANO parameters
    [rot3|pose45|shape10|scale1|trans3], centre (U(+-40), U(+-40), U(500,1200)) mm, cube 250^3."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    p = torch.zeros(B, 62)
    p[:, :3] = (torch.rand(B, 3, generator=g) * 2 - 1) * math.pi
    p[:, 3:48] = torch.randn(B, 45, generator=g) * 0.5
    p[:, 48:58] = torch.randn(B, 10, generator=g) * 0.5
    p[:, 58] = 1.0
    center = torch.stack([torch.rand(B, generator=g) * 80 - 40, torch.rand(B, generator=g) * 80 - 40,
                          torch.rand(B, generator=g) * 700 + 500], 1)
    cube = torch.full((B, 3), 250.0)
    return p.to(device, dtype), center.to(device, dtype), cube.to(device, dtype)


def _default_adamw(params, lr, weight_decay):
    """AdamW as the reference builds it (train_render.py:131-139): the one-launch HIP version
    (dsf_amd.optim.FusedAdamW; DSF_FUSED_ADAMW=0 selects torch.optim.AdamW's multi-tensor kernels for A/B runs)."""
    import os
    if os.environ.get("DSF_FUSED_ADAMW", "1") == "1":
        from .optim import FusedAdamW
        return FusedAdamW(params, lr=lr, weight_decay=weight_decay)
    return torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay)


class RenderSupervisedStep:
    """BASELINE config 2: backbone forward (incl. the stage-2 re-render bridge) -> per stage the
    pixel-branch losses and the MANO-branch losses of ``Pretrain`` (train_render.py:444-466), plus
    ``Render.render`` of the final MANO estimate and the model-to-data depth term
    (:719-732) against the target depth image; backward; AdamW step."""

    def __init__(self, net, render, config=Config, optimizer=None, grad_sync=None):
        self.net, self.render, self.cfg = net, render, config
        self.L1 = SmoothL1Loss()
        self.gfm = GFM()
        self.opt = optimizer if optimizer is not None else _default_adamw(net.parameters(), lr=config.lr,
                                                                             weight_decay=config.weight_decay)
        self.grad_sync = grad_sync                    # dsf_amd.parallel.GradAllReducer or None
        self.utils = TensorUtils(img_size=config.input_size)

    @torch.no_grad()
    def make_targets(self, params_gt, center, cube, noise_sigma=0.02, seed=1):
        """'real' depth = render of the ground-truth parameters + Gaussian depth noise on the
        foreground; also the GT joints / mesh / offset maps."""
        img, juvd, jxyz, mesh = self.render.render(params_gt, center, cube)
        g = torch.Generator(device=img.device).manual_seed(seed)
        noise = torch.randn(img.shape, device=img.device, generator=g) * noise_sigma
        img = torch.where(img < 0.99, (img + noise).clamp(-1, 0.98), img)
        return {"img": img.contiguous(), "joint_uvd": juvd, "joint_xyz": jxyz, "mesh_xyz": mesh,
                "center": center, "cube": cube}

    def loss(self, tgt):
        cfg = self.cfg
        img, center, cube = tgt["img"], tgt["center"], tgt["cube"]
        outputs = self.net(img, self.render, center, cube)
        terms = {}
        for s, (pixel_pd, mano_pd) in enumerate(outputs):
            S = pixel_pd.size(-1)
            # pixel-wise branch (:451-456); loss weights are folded into the fused Huber kernels
            pixel_gt = self.gfm.joint2feature(tgt["joint_uvd"], img, cfg.feature_para, S, cfg.feature_type)
            juvd_pd = self.gfm.feature2joint(img, pixel_pd, cfg.feature_type, cfg.feature_para)
            terms["pix%d" % s] = self.L1(pixel_pd, pixel_gt, weight=cfg.deconv_weight)
            terms["coord%d" % s] = self.L1(juvd_pd, tgt["joint_uvd"], weight=cfg.coord_weight)
            # model branch (:459-466)
            jxyz_pd, mesh_pd = self.render.get_mesh_xyz(mano_pd)
            terms["joint%d" % s] = self.L1(jxyz_pd, tgt["joint_xyz"], weight=cfg.coord_weight)
            terms["vert%d" % s] = self.L1(mesh_pd, tgt["mesh_xyz"], weight=cfg.coord_weight)
            terms["beta%d" % s] = torch.mean(torch.pow(mano_pd[:, 48:58], 2)) * (cfg.coord_weight * 10)
            terms["scale%d" % s] = torch.mean(torch.abs(torch.clamp(mano_pd[:, 58], max=0.0))) * 0.1
        # render loss on the final estimate (:719, :728-732, :745)
        img_pd, _, _, _ = self.render.render(outputs[-1][1], center, cube)
        terms["m2d"] = m2d_loss(img, img_pd) * cfg.model_weight
        # one stack + one sum instead of a chain of scalar adds (and their backward kernels)
        return torch.stack(list(terms.values())).sum(), terms

    def __call__(self, tgt):
        from . import nn_conv
        self.opt.zero_grad(set_to_none=True)
        self.render.mano_layer.clear_cache()                 # results of the previous step must not outlive its graph
        loss, terms = self.loss(tgt)
        if not hasattr(self, "_pool_floats"):
            self._pool_floats = nn_conv.weight_grad_floats(self.net) + 64          # fused heads re-lay one merged weight
            self._pool_dev = next(self.net.parameters()).device
        with nn_conv.grad_pool(self._pool_floats if self._pool_dev.type == "cuda" else 0, self._pool_dev):
            loss.backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.opt.step()
        return loss.detach(), terms


class MeshLossStep:
    """BASELINE config 3: pixel network with a MANO head (``PoseNetMANO``) + the self-supervised
    geometry terms of FinetuneStage on a target depth map: m2d (:728-732), whole-hand ICP (:739),
    part-aware ICP (:735) with sphere segmentation (:695-701), sphere collision (:725), plus the
    supervised joint/vertex terms that keep the synthetic problem well-posed."""

    def __init__(self, net, render, config=Config, optimizer=None, grad_sync=None, n_points=2048):
        self.net, self.render, self.cfg, self.n_points = net, render, config, n_points
        self.L1 = SmoothL1Loss()
        self.opt = optimizer if optimizer is not None else _default_adamw(net.parameters(), lr=config.lr,
                                                                             weight_decay=config.weight_decay)
        self.grad_sync = grad_sync
        self.utils = TensorUtils(img_size=config.input_size)

    @torch.no_grad()
    def make_targets(self, params_gt, center, cube, seed=1):
        img, juvd, jxyz, mesh = self.render.render(params_gt, center, cube)
        _, M, _, _ = ops.crop_setup(center, cube, self.render.cam, 128)
        u = self.utils
        crop = u.crop_hand(img, jxyz, center, M, cube)
        mano = self.render.mano_layer
        B = img.size(0)
        g = torch.Generator(device=img.device).manual_seed(seed)
        keys = lambda: torch.randint(0, 2 ** 31 - 1, (B, 128 * 128), device=img.device, dtype=torch.int32, generator=g)
        _, pts = u.crop_hand(crop, jxyz, center, M, cube, return_points=True)
        seg_img = mano.seg_pcl(jxyz, jxyz, mesh, pts)                                    # (:695)
        seg_img = torch.where(crop.lt(0.99).reshape(B, -1), seg_img, torch.zeros_like(seg_img)).reshape(B, 1, 128, 128)
        joint_img = torch.where(seg_img.gt(0), crop, torch.ones_like(crop))              # (:698)
        joint_pcl = u.Img2pcl(joint_img, 128, center, M, cube, self.n_points, rand_keys=keys())   # (:699)
        seg = mano.seg_pcl(jxyz, jxyz, mesh, joint_pcl)                                  # (:700)
        pcl = u.Img2pcl(crop, 128, center, M, cube, self.n_points, rand_keys=keys())     # (:701)
        return {"img": img, "crop": crop, "joint_xyz": jxyz, "mesh_xyz": mesh, "joint_pcl": joint_pcl, "seg": seg,
                "pcl": pcl, "center": center, "cube": cube, "M": M}

    def loss(self, tgt):
        cfg = self.cfg
        mano_layer = self.render.mano_layer
        _, mano_pd = self.net(tgt["img"])
        img_pd, juvd, jxyz, mesh = self.render.render(mano_pd, tgt["center"], tgt["cube"])
        crop_pd = self.utils.crop_hand(img_pd, tgt["joint_xyz"], tgt["center"], tgt["M"], tgt["cube"])
        l_m2d = m2d_loss(tgt["crop"], crop_pd) * cfg.model_weight
        l_part = JointICPLoss(mesh, tgt["joint_pcl"], mano_layer.joint_faces, tgt["seg"]).mean(-1).mean(-1) * cfg.partICP_weight
        l_icp = ICPLoss(mesh, tgt["pcl"], mano_layer.faces).mean(-1) * cfg.model_weight
        l_coll = mano_layer.calculate_coll(jxyz, mesh.detach()) * cfg.coll_weight
        l_sup = (self.L1(jxyz, tgt["joint_xyz"]) + self.L1(mesh, tgt["mesh_xyz"])) * cfg.coord_weight
        terms = {"m2d": l_m2d, "pd2m": l_part, "d2m": l_icp, "coll": l_coll, "sup": l_sup}
        return l_m2d + l_part + l_icp + l_coll + l_sup, terms

    def __call__(self, tgt):
        self.opt.zero_grad(set_to_none=True)
        loss, terms = self.loss(tgt)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.opt.step()
        return loss.detach(), terms


class FinetuneStageStep:
    """Counterpart of ``Trainer.FinetuneStage`` (train_render.py:622-823), the reference's default
    self-boosting step (config.py:36-38): a synthetic supervised pass through the frozen transfer
    generator + a real-image pass where the detached stage-2 outputs teach stage 1 and the geometry
    terms (m2d, ICP, part-aware ICP, collision, P2M, M2P) close the self-supervised loop.

    Differences from the reference, all forced by "no host sync inside the step":
      * random draws are explicit (a ``torch.Generator`` per step instead of global RNG state);
      * the M2P term (:784-801) selects rows with ``nonzero()`` + ``index_select`` and branches on
        ``joint_mano_mask.sum() == 0`` on the host; here it is the algebraically identical masked mean
        (same rows, same divisor, same "sum of selected indices == 0 -> 0" rule), evaluated on device;
      * ``xyz2error`` / TensorBoard / colour LUT host round trips (:654-667, :703, :713-721) are dropped.
    """

    def __init__(self, net, render, transfer_net, config=Config, optimizer=None, grad_sync=None, mask=True):
        self.net, self.render, self.transfer, self.cfg, self.mask = net, render, transfer_net, config, mask
        self.L1 = SmoothL1Loss()
        self.gfm = GFM()
        self.opt = optimizer if optimizer is not None else _default_adamw(net.parameters(), lr=config.lr,
                                                                             weight_decay=config.weight_decay)
        self.grad_sync = grad_sync
        self.utils = TensorUtils(img_size=config.input_size)
        if transfer_net is not None:
            for p in transfer_net.parameters():
                p.requires_grad_(False)
            transfer_net.eval()

    def _keys(self, B, g, dev):
        return torch.randint(0, 2 ** 31 - 1, (B, 128 * 128), device=dev, dtype=torch.int32, generator=g)

    def _masked_huber(self, a, b, row_mask):
        """L1Loss(index_select(a, rows), index_select(b, rows)) with the reference's empty rule (:796-801)."""
        z = (a - b).float()
        az = z.abs()
        per_row = torch.where(az < 0.01, 0.5 * z * z, 0.01 * (az - 0.005)).mean(-1)          # (rows,)
        m = row_mask.to(per_row.dtype)
        n = m.sum()
        idx_sum = (torch.arange(m.numel(), device=m.device, dtype=per_row.dtype) * m).sum()
        val = (per_row * m).sum() / torch.clamp(n, min=1.0)
        return torch.where(idx_sum == 0, torch.zeros_like(val), val)

    def loss(self, model_para, cube, img_r, center_r, cube_r, M_r, generator=None):
        cfg, R, u, gfm, L1 = self.cfg, self.render, self.utils, self.gfm, self.L1
        mano_layer = R.mano_layer
        dev = model_para.device
        B = model_para.size(0)
        g = generator if generator is not None else torch.Generator(device=dev)
        rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
        # ---- synthetic branch (:628-667) ----
        aug_shape = torch.randn(B, 10, device=dev, generator=g) * 3
        aug_center = (rnd(B, 3) - 0.5) * 40
        aug_size = 1 + (rnd(B, 1) - 0.5) * 0.4
        aug_view = rnd(B, 3) * math.pi * 2
        depth = rnd(B, 1) * (R.depth_range[1] - R.depth_range[0]) + R.depth_range[0]
        center0 = torch.cat((torch.zeros(B, 2, device=dev), depth), dim=-1)
        with torch.no_grad():
            draws = None
            if self.mask:
                k = 6                                   # reference: 3..9 occluders drawn on the host (:1328); fixed count keeps the step sync-free
                jid = torch.randperm(21, device=dev, generator=g)[:k]
                draws = (jid, (rnd(B, k, 3) - 0.5) * 0.15 * 2, rnd(B, k) * 0.3)
            img, juvd_gt, _, jxyz_gt, mesh_gt, center_s, cube_s, M_s = R(model_para, center0, cube, augmentView=aug_view,
                                                                        augmentShape=aug_shape, augmentCenter=aug_center,
                                                                        augmentSize=aug_size, mask=False)
            if draws is not None:
                img = R.mask_img(img, juvd_gt, 0.15, 0.3, draws=draws)
            img_t = self.transfer(img) if self.transfer is not None else img
        outputs = self.net(img_t, R, center=center_s, cube=cube_s)
        total = 0
        for pixel_pd, mano_pd in outputs:
            S = pixel_pd.size(-1)
            pixel_gt = gfm.joint2feature(juvd_gt, img, cfg.feature_para, S, cfg.feature_type)
            juvd = gfm.feature2joint(img, pixel_pd, cfg.feature_type, cfg.feature_para)
            total = total + L1(pixel_pd, pixel_gt) * cfg.deconv_weight + L1(juvd, juvd_gt) * cfg.coord_weight
            jx, mx = R.get_mesh_xyz(mano_pd)
            total = total + L1(jx, jxyz_gt) * cfg.coord_weight + L1(mx, mesh_gt) * cfg.coord_weight \
                + mano_layer.calculate_coll(jx, mx.detach()) * cfg.coll_weight
        # ---- real branch: teacher from the detached stage-2 outputs (:671-703) ----
        outputs = self.net(img_r, R, center=center_r, cube=cube_r)
        pix_t, mano_t = outputs[1][0].detach(), outputs[1][1].detach()
        with torch.no_grad():
            juvd_t = gfm.feature2joint(img_r, pix_t, cfg.feature_type, cfg.feature_para)
            jxyz_t = u.uvd_nl2xyznl_tensor(juvd_t, center_r, M_r, cube_r)
            mj_t, mm_t = R.get_mesh_xyz(mano_t)
            crop_r, pts = u.crop_hand(img_r, mj_t, center_r, M_r, cube_r, return_points=True)
            _, pts = u.crop_hand(crop_r, mj_t, center_r, M_r, cube_r, return_points=True)
            seg_img = mano_layer.seg_pcl(jxyz_t, mj_t, mm_t, pts)
            seg_img = torch.where(crop_r.lt(0.99).reshape(B, -1), seg_img, torch.zeros_like(seg_img)).reshape(B, 1, 128, 128)
            joint_img = torch.where(seg_img.gt(0), crop_r, torch.ones_like(img_r))
            joint_pcl = u.Img2pcl(joint_img, 128, center_r, M_r, cube_r, 2048, rand_keys=self._keys(B, g, dev))
            segment = mano_layer.seg_pcl(jxyz_t, mj_t, mm_t, joint_pcl)
            pcl = u.Img2pcl(crop_r, 128, center_r, M_r, cube_r, 2048, rand_keys=self._keys(B, g, dev))
        # ---- stage 1 student (:706-749) ----
        pix1, mano1 = outputs[0]
        juvd1 = gfm.feature2joint(img_r, pix1, cfg.feature_type, cfg.feature_para)
        total = total + L1(pix1, pix_t) * cfg.deconv_weight + L1(juvd1, juvd_t) * cfg.coord_weight
        img1, mjuvd1, mjxyz1, mesh1 = R.render(mano1, center_r, cube_r)
        total = total + L1(mjxyz1, jxyz_t) * cfg.coord_weight + L1(mesh1, mm_t) * cfg.coord_weight
        total = total + mano_layer.calculate_coll(mjxyz1, mesh1.detach()) * cfg.coll_weight
        crop1 = u.crop_hand(img1, mj_t, center_r, M_r, cube_r)
        total = total + m2d_loss(crop_r, crop1) * cfg.model_weight
        total = total + ICPLoss(mesh1, pcl, mano_layer.faces).mean(-1) * cfg.model_weight
        total = total + JointICPLoss(mesh1, joint_pcl, mano_layer.joint_faces, segment).mean(-1).mean(-1) * cfg.partICP_weight
        # ---- stage 2 (:752-808) ----
        pix2, mano2 = outputs[1]
        juvd2 = gfm.feature2joint(img_r, pix2, cfg.feature_type, cfg.feature_para)
        img2, mjuvd2, mjxyz2, mesh2 = R.render(mano2, center_r, cube_r)
        p2m = L1(mjuvd2, juvd_t) * cfg.coord_weight
        coll2 = mano_layer.calculate_coll(mjxyz2, mesh2.detach())
        crop2 = u.crop_hand(img2, mj_t, center_r, M_r, cube_r)
        union = (crop_r.lt(0.99) | crop2.lt(0.99)).float()
        m2d2 = m2d_loss(crop_r, crop2)
        pd2m_j = JointICPLoss(mesh2, joint_pcl, mano_layer.joint_faces, segment)
        d2m_b = ICPLoss(mesh2, pcl, mano_layer.faces)
        both = (crop_r.lt(0.99) & crop2.lt(0.99)).float()
        depth_b = ((crop_r - crop2).abs() * both).sum(-1).sum(-1) / (union.sum(-1).sum(-1) + 1e-8)
        mano_ok = depth_b.lt(0.04).squeeze(-1) & d2m_b.lt(1e-3)                              # (:787-789)
        jm = pd2m_j.lt(1e-3)
        jm = torch.cat((torch.ones(B, 1, device=dev, dtype=torch.bool), jm, jm[:, [2, 5, 8, 11, 14]]), dim=-1)
        rows = (mano_ok.unsqueeze(-1) & jm).detach().reshape(-1)
        m2p = self._masked_huber(juvd2.reshape(-1, 3), mjuvd2.detach().reshape(-1, 3), rows) * cfg.coord_weight
        total = total + p2m + coll2 * cfg.coll_weight + m2d2 * cfg.model_weight + d2m_b.mean(-1) * cfg.model_weight \
            + pd2m_j.mean(-1).mean(-1) * cfg.partICP_weight + m2p * cfg.M2P_weight
        terms = {"P2M": p2m, "m2d": m2d2, "d2m": d2m_b.mean(-1), "pd2m": pd2m_j.mean(-1).mean(-1), "M2P": m2p, "coll": coll2}
        return total, terms

    def __call__(self, model_para, cube, img_r, center_r, cube_r, M_r, generator=None):
        self.opt.zero_grad(set_to_none=True)
        loss, terms = self.loss(model_para, cube, img_r, center_r, cube_r, M_r, generator)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.opt.step()
        return loss.detach(), terms
