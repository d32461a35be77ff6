"""Counterpart of the GPU-side tensor utilities of the reference's ``data/render_loader.py``
``loader`` class (coordinate transforms, crop_hand, Img2pcl; :336-353, 1044-1227) plus the test-phase depth crop
(``Crop_Image_deep_pp`` + ``normalize_img``, :748-810, 738-745) as one device kernel over a batch of raw frames
(SURVEY 8f row 1).  File readers and the training-phase cv2 augmentations are out of scope."""
import torch

from .. import _lib as L
from .. import ops


class loader:
    """Only the tensor utilities the training step calls; ``paras``/``flip`` as in nyu_loader
    (/root/reference/data/render_loader.py:1808-1812)."""

    def __init__(self, root_dir=None, phase='train', img_size=128, center_type='refine', dataset_name='nyu',
                 paras=(588.03, 587.07, 320., 240.), ori_img_size=(640, 480)):
        self.dataset_name = dataset_name
        self.img_size = img_size
        self.paras = paras
        self.flip = 1
        self.ori_img_size = ori_img_size
        self.cam = L.camera(paras, ori_img_size)

    def crop_normalize(self, depth, com, cube_size, paras=None, want_raw=False):
        """Batched device version of the ``phase == 'test'`` branch of ``__getitem__`` (:1909-1916):
        ``Crop_Image_deep_pp(depth, center_uvd, cube_size, (img_size, img_size), paras)`` followed by
        ``normalize_img(crop.max(), crop, center, cube_size)`` -> data (B,1,S,S) f32, M (B,3,3) f64 [, raw crop]."""
        return ops.depth_crop_normalize(depth, com, cube_size, paras or self.paras, self.img_size, want_raw)

    aug_modes = ['rot', 'com', 'sc', 'none']                      # nyu_loader (:1816)

    def rand_augment(self, B, sigma_com=10., sigma_sc=0.2, rot_range=180., rng=None):
        """``rand_augment`` (:625-650) for a batch, on the host (as the reference: Python's ``random``): mode index,
        off = U(-1,1)^3 * sigma_com mm, rot = U(-rot_range, rot_range) degrees, sc = |1 + U(-1,1) * sigma_sc|
        (defaults: config.py:85 ``augment_para``).  -> numpy arrays, the explicit inputs of ``augmentCrop``."""
        import numpy as np
        rng = rng if rng is not None else np.random.default_rng()
        return (rng.integers(0, len(self.aug_modes), B).astype(np.int32), rng.uniform(-1, 1, (B, 3)) * sigma_com,
                rng.uniform(-rot_range, rot_range, B), np.abs(1. + rng.uniform(-1, 1, B) * sigma_sc))

    def augmentCrop(self, crop, gt3Dcrop, com, cube, M, mode, off, rot, sc, paras=None):
        """Batched device version of ``augmentCrop`` (:653-695) on raw crops (``crop_normalize(..., want_raw=True)[2]``):
        -> (imgD (B,1,S,S) normalised, new_joints3D (B,J,3), cube (B,3), com (B,3), M (B,3,3))."""
        return ops.depth_augment_crop(crop, gt3Dcrop, com, cube, M, mode, off, rot, sc, paras or self.paras, self.flip)

    @staticmethod
    def _b3(x, B):
        return x.reshape(B, 3)

    @staticmethod
    def _inv(M):
        return ops.inverse3x3(M)                 # once per M tensor (ops._memo)

    def uvd_nl2xyznl_tensor(self, uvd, center, m, cube):
        B = uvd.size(0)
        return ops.UvdToXyz.apply(uvd, self._b3(center, B), self._inv(m), self._b3(cube, B), self.cam, self.img_size, True)

    def uvd_nl2xyz_tensor(self, uvd, center, m, cube):
        B = uvd.size(0)
        return ops.UvdToXyz.apply(uvd, self._b3(center, B), self._inv(m), self._b3(cube, B), self.cam, self.img_size, False)

    def xyz_nl2uvdnl_tensor(self, joint_xyz, center, M, cube_size):
        B = joint_xyz.size(0)
        return ops.XyzToUvd.apply(joint_xyz, self._b3(center, B), M.reshape(B, 3, 3), self._b3(cube_size, B), self.cam,
                                  self.img_size, False)

    def pointsImgTo3D(self, point_uvd, flip=None):
        fx, fy, fu, fv = self.paras
        flip = self.flip if flip is None else flip
        x = (point_uvd[..., 0] - fu) * point_uvd[..., 2] / fx
        y = flip * (point_uvd[..., 1] - fv) * point_uvd[..., 2] / fy
        return torch.stack([x, y, point_uvd[..., 2]], -1)

    def points3DToImg(self, joint_xyz, flip=None):
        fx, fy, fu, fv = self.paras
        flip = self.flip if flip is None else flip
        u = joint_xyz[..., 0] * fx / (joint_xyz[..., 2] + 1e-8) + fu
        v = flip * joint_xyz[..., 1] * fy / joint_xyz[..., 2] + fv
        return torch.stack([u, v, joint_xyz[..., 2]], -1)

    def uvdImg2xyzImg(self, uvd_img, center, M, cube):
        """-> (xyz_mm, xyz_normalised), each (B,3,S,S) (reference :1190-1201)."""
        B, _, S, _ = uvd_img.shape
        g = 2.0 * torch.arange(S, device=uvd_img.device).float() / (S - 1.0) - 1.0
        uvd = torch.stack([g.view(1, 1, S).expand(B, S, S), g.view(1, S, 1).expand(B, S, S), uvd_img[:, 0]], -1)
        uvd = uvd.reshape(B, S * S, 3)
        back = lambda t: t.permute(0, 2, 1).reshape(B, 3, S, S)
        return back(self.uvd_nl2xyz_tensor(uvd, center, M, cube)), back(self.uvd_nl2xyznl_tensor(uvd, center, M, cube))

    def crop_hand(self, img, joint, center, M, cube, offsetxy=25, offsetz=20, hand_thickness=20, return_points=False):
        """Keep pixels inside the skeleton's padded 3-D box, others -> 1 (reference :1209-1227)."""
        B = img.size(0)
        out, xyz_nl, _ = ops.CropHand.apply(img, joint.detach(), self._b3(center, B), self._inv(M), self._b3(cube, B),
                                            self.cam, float(offsetxy), float(offsetz), float(hand_thickness))
        return (out, xyz_nl) if return_points else out

    def Img2pcl(self, img, feature_size, center, M, cube, sample_num=1024, rand_keys=None):
        """(B,1,S,S) -> (B,sample_num,3) normalised points (reference :1121-1156); the random draw is
        the explicit ``rand_keys`` input."""
        assert img.size(-1) == feature_size, "Img2pcl resamples at the image resolution only"
        B = img.size(0)
        pcl, _ = ops.img2pcl(img.detach(), self._b3(center, B), self._inv(M), self._b3(cube, B), self.cam,
                             int(sample_num), rand_keys)
        return pcl
