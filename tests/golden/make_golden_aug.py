"""Golden vectors for the training-phase augmentation of the depth data path: runs the REFERENCE's own
`loader.augmentCrop` (data/render_loader.py:653-695 -> moveCoM / rotateHand / scaleHand / recropHand / normalize_img) on the
crops of tests/golden/make_golden_data.py's synthetic frames, one case per (frame, mode).  Build container only:
    python tests/golden/make_golden_aug.py   ->   tests/golden/reference_aug.npz

OpenCV is absent: `cv2.warpPerspective`, `cv2.getRotationMatrix2D`, `cv2.warpAffine` (INTER_NEAREST, BORDER_CONSTANT) and
`cv2.resize` are provided by oracle.data_ref's restatements of OpenCV's published rules -- the only steps of these vectors
that are not the reference's code.  The reference runs here under NumPy 2 (its float32 `joint3DToImg` centre then makes
`comToBounds` float32 arithmetic, where NumPy 1 promoted to float64); the consuming test says which cases that touches.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg
import make_golden_data as mgd


def draws(rng, n):
    """rand_augment (render_loader.py:625-650) draws as explicit inputs: off U(-1,1)^3 * sigma_com, rot U(-180,180), sc"""
    sigma_com, sigma_sc, rot_range = 10., 0.2, 180.                      # config.py:85 augment_para
    return [(rng.uniform(-1, 1, 3) * sigma_com, float(rng.uniform(-rot_range, rot_range)), float(abs(1. + rng.uniform(-1, 1) * sigma_sc)))
            for _ in range(n)]


def main():
    mg._install_stubs()
    sys.path.insert(0, mg.REPO)
    from oracle import data_ref
    import cv2
    cv2.INTER_NEAREST, cv2.BORDER_CONSTANT = 0, 0
    cv2.resize = lambda src, dsize, interpolation=0: data_ref.resize_nearest(src, dsize)
    cv2.warpPerspective = lambda src, M, dsize, flags=0, borderMode=0, borderValue=0.: data_ref.cv_warp_perspective_nn(src, M, dsize, borderValue)
    cv2.getRotationMatrix2D = data_ref.cv_get_rotation_matrix_2d
    cv2.warpAffine = lambda src, M, dsize, flags=0, borderMode=0, borderValue=0.: data_ref.cv_warp_affine_nn(src, M, dsize, borderValue)
    sys.path.insert(0, mg.REF)
    from data import render_loader as rl
    rl.xrange = range                                                    # (Python 2 leftover in rotateHand)
    L = rl.loader("/x", "train", 128, "refine", "nyu")
    L.paras, L.flip = mgd.PARAS, 1
    L.aug_modes = ['rot', 'com', 'sc', 'none']
    rng = np.random.RandomState(11)
    depth, com, cube = mgd.frames(rng, 12)
    rng2 = np.random.RandomState(23)
    dr = draws(rng2, 12)
    joints = (rng2.uniform(-1, 1, (12, 14, 3)) * np.array([90., 90., 60.])).astype(np.float32)      # gt3Dcrop: mm, relative to the centre
    out = {"joints_in": joints, "off": np.stack([d[0] for d in dr]), "rot": np.array([d[1] for d in dr]), "sc": np.array([d[2] for d in dr])}
    imgs, jo, cubes, coms, Ms = [], [], [], [], []
    for i in range(12):
        crop, trans = L.Crop_Image_deep_pp(depth[i].copy(), com[i], cube[i], (128, 128), mgd.PARAS)
        for mode in range(4):
            img, _, j, cb, cm, M, _ = L.augmentCrop(crop.copy(), joints[i].copy(), com[i].copy(), list(cube[i]), trans.copy(), mode,
                                                    dr[i][0].copy(), dr[i][1], dr[i][2], mgd.PARAS)
            imgs.append(np.asarray(img, np.float32)); jo.append(np.asarray(j, np.float32)); cubes.append(np.asarray(cb, np.float64))
            coms.append(np.asarray(cm, np.float64)); Ms.append(np.asarray(M, np.float64))
    out.update(img=np.stack(imgs), joints=np.stack(jo), cube=np.stack(cubes), com=np.stack(coms), M=np.stack(Ms))
    np.savez_compressed(os.path.join(HERE, "reference_aug.npz"), **out)
    print("reference_aug.npz", os.path.getsize(os.path.join(HERE, "reference_aug.npz")) // 1024, "KiB", out["img"].shape)


if __name__ == "__main__":
    main()
