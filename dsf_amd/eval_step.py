"""Evaluation path of the reference's trainer (``Trainer.test`` / ``test_iter`` / ``xyz2error``,
train_render.py:316-400, 826-864) on the HIP kernels: eval-mode backbone forward (frozen-statistics BN on
``dsf_bn_apply``), offset-map decoding, crop-uvd -> cube-xyz, MANO joints, and the mean joint error in mm computed on
the device without a host sync per batch (the reference moves every tensor to numpy per call)."""
import numpy as np
import torch

from .data.render_loader import loader as TensorUtils
from .util.generateFeature import GFM

ICVL_BIAS = (20, 22, 13.5, 7.5, 12.5, 12.5, 3, 12.5, 12.5, 8, 16, 12.5, 3, 13, 7.3, 6)


def xyz2error(output, joint, center, cube_size, dataset="nyu", keep_batch=False, keep_joint=False):
    """``Trainer.xyz2error`` (train_render.py:826-851) as device tensors: output / joint (B,J,3) cube-normalised,
    center / cube_size (B,3) mm -> scalar tensor (or (B,) / (B,J))."""
    c, s = center.unsqueeze(1), cube_size.unsqueeze(1)
    a = output.detach() * s / 2 + c
    b = joint.detach() * s / 2 + c
    if dataset == "icvl":
        bias = torch.tensor(ICVL_BIAS, device=a.device, dtype=a.dtype).view(1, 16)
        a = torch.cat([a[..., :2], (a[..., 2] - bias).unsqueeze(-1)], -1)
    d = (a - b).pow(2).sum(-1).sqrt()
    if keep_joint:
        return d
    if keep_batch:
        return d.mean(-1)
    return d[:, 1:].mean() if dataset == "msra" else d.mean()


class EvalStep:
    """``Trainer.test_iter`` (:354-400): per network stage the pixel-branch error and the MANO-branch error."""

    def __init__(self, net, render, config, dataset="nyu"):
        self.net, self.render, self.cfg, self.dataset = net, render, config, dataset
        self.gfm = GFM()
        self.utils = TensorUtils(img_size=config.input_size)
        self.transfer = list(render.mano_layer.transfer)
        from . import nn_conv
        # the scored net is read-only here: its split weight images are kept from one batch to the next (a load_state_dict or an
        # in-place torch write bumps the version counters; raw ``.data`` writes must call nn_conv.weights_changed())
        nn_conv.manage_weights(net.parameters())

    @torch.no_grad()
    def test_iter(self, img, xyz_gt, center, cube, M, writers=None):
        """-> [pixel error, MANO error] per stage (device scalars, mm).  ``writers``: optional dict of open text files
        {'result': [per-output files], 'mesh': f, 'mano': f} written in the reference's format (:384-398)."""
        cfg = self.cfg
        outputs = self.net(img, self.render, center, cube)
        errors = []
        for index, (pixel_pd, mano_para) in enumerate(outputs):
            all_uvd = self.gfm.feature2joint(img, pixel_pd, cfg.feature_type, cfg.feature_para)
            all_xyz = self.utils.uvd_nl2xyznl_tensor(all_uvd, center, M, cube)
            mano_xyz, mano_mesh = self.render.get_mesh_xyz(mano_para)
            joint_xyz = all_xyz[:, self.transfer, :]
            mano_joint_xyz = mano_xyz[:, self.transfer, :]
            n = joint_xyz.size(1)
            errors.append(xyz2error(joint_xyz[:, :n - 1], xyz_gt[:, :n - 1], center, cube, self.dataset))
            errors.append(xyz2error(mano_joint_xyz[:, :n - 1], xyz_gt[:, :n - 1], center, cube, self.dataset))
        if writers is not None:
            B = img.size(0)
            if "mesh" in writers:
                world = mano_mesh * cube.unsqueeze(-2) / 2 + center.unsqueeze(-2)
                np.savetxt(writers["mesh"], world.cpu().numpy().reshape([B, -1]), fmt='%.3f')
            if "mano" in writers:
                np.savetxt(writers["mano"], mano_para.cpu().numpy().reshape([B, -1]), fmt='%.3f')
            if "result" in writers:
                jw = all_xyz * cube.unsqueeze(-2) / 2 + center.unsqueeze(-2)
                mw = mano_xyz * cube.unsqueeze(-2) / 2 + center.unsqueeze(-2)
                np.savetxt(writers["result"][0], self.render.points3DToImg(jw).cpu().numpy().reshape([B, -1]), fmt='%.3f')
                np.savetxt(writers["result"][1], self.render.points3DToImg(mw).cpu().numpy().reshape([B, -1]), fmt='%.3f')
        return errors

    @torch.no_grad()
    @torch.no_grad()
    def test_frames(self, depth, center_uvd, cube, xyz_gt, paras=(588.03, 587.07, 320., 240.), writers=None):
        """``test_iter`` straight from raw depth frames: the reference's test-phase ``loader.__getitem__``
        (data/render_loader.py:1897-1916: ``Crop_Image_deep_pp`` + ``normalize_img``, ``com3D = jointImgTo3D(com2D)``)
        runs on the device first (``dsf_depth_crop_normalize``), so no cropped images cross PCIe.
        depth (B,Hd,Wd) f32 mm, center_uvd (B,3) (u, v, z mm), cube (B,3) mm, xyz_gt as for ``test_iter``."""
        from . import ops
        img, trans = ops.depth_crop_normalize(depth, center_uvd, cube, paras, self.cfg.input_size)
        uvd = torch.as_tensor(center_uvd, dtype=torch.float32, device=img.device).reshape(-1, 3)
        fx, fy, fu, fv = paras
        center = torch.stack(((uvd[:, 0] - fu) * uvd[:, 2] / fx, (uvd[:, 1] - fv) * uvd[:, 2] / fy, uvd[:, 2]), dim=1)   # flip = 1 (NYU)
        cube = torch.as_tensor(cube, dtype=torch.float32, device=img.device).reshape(-1, 3).expand(img.size(0), 3)
        return self.test_iter(img, xyz_gt, center, cube, trans.float(), writers)

    def test(self, batches):
        """``Trainer.test`` (:316-352) over an iterable of (img, xyz_gt, uvd_gt, center, M, cube) device batches:
        -> (mean over outputs of the batch-averaged errors, per-output list).  One host sync at the end."""
        was_training = self.net.training
        self.net.eval()
        total, n = None, 0
        for img, xyz_gt, _uvd_gt, center, M, cube in batches:
            e = torch.stack(self.test_iter(img, xyz_gt, center, cube, M))
            total = e if total is None else total + e
            n += 1
        self.net.train(was_training)
        per_output = (total / max(n, 1)).tolist()
        return sum(per_output) / len(per_output), per_output
