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


class RenderSupervisedStep:
    """BASELINE config 2: backbone forward (incl. the stage-2 re-render bridge) -> per stage the
    pixel-branch losses and the MANO-branch losses of ``Pretrain`` (train_render.py:444-466), plus
    ``Render.render`` of the final MANO estimate and the model-to-data depth term
    (:719-732) against the target depth image; backward; AdamW step."""

    def __init__(self, net, render, config=Config, optimizer=None, grad_sync=None):
        self.net, self.render, self.cfg = net, render, config
        self.L1 = SmoothL1Loss()
        self.gfm = GFM()
        self.opt = optimizer if optimizer is not None else torch.optim.AdamW(net.parameters(), lr=config.lr,
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
        total = 0
        terms = {}
        for s, (pixel_pd, mano_pd) in enumerate(outputs):
            S = pixel_pd.size(-1)
            # pixel-wise branch (:451-456)
            pixel_gt = self.gfm.joint2feature(tgt["joint_uvd"], img, cfg.feature_para, S, cfg.feature_type)
            juvd_pd = self.gfm.feature2joint(img, pixel_pd, cfg.feature_type, cfg.feature_para)
            l_pix = self.L1(pixel_pd, pixel_gt) * cfg.deconv_weight
            l_crd = self.L1(juvd_pd, tgt["joint_uvd"]) * cfg.coord_weight
            # model branch (:459-466)
            jxyz_pd, mesh_pd = self.render.get_mesh_xyz(mano_pd)
            l_j = self.L1(jxyz_pd, tgt["joint_xyz"]) * cfg.coord_weight
            l_v = self.L1(mesh_pd, tgt["mesh_xyz"]) * cfg.coord_weight
            l_beta = torch.mean(torch.pow(mano_pd[:, 48:58], 2)) * cfg.coord_weight * 10
            l_scale = torch.mean(torch.abs(torch.clamp(mano_pd[:, 58], max=0.0))) * 0.1
            total = total + l_pix + l_crd + l_j + l_v + l_beta + l_scale
            terms["pix%d" % s], terms["coord%d" % s], terms["joint%d" % s], terms["vert%d" % s] = l_pix, l_crd, l_j, l_v
        # render loss on the final estimate (:719, :728-732, :745)
        img_pd, _, _, _ = self.render.render(outputs[-1][1], center, cube)
        l_m2d = m2d_loss(img, img_pd) * cfg.model_weight
        terms["m2d"] = l_m2d
        return total + l_m2d, terms

    def __call__(self, tgt):
        self.opt.zero_grad(set_to_none=False)
        loss, terms = self.loss(tgt)
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
        self.opt = optimizer if optimizer is not None else torch.optim.AdamW(net.parameters(), lr=config.lr,
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
        self.opt.zero_grad(set_to_none=False)
        loss, terms = self.loss(tgt)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.opt.step()
        return loss.detach(), terms
