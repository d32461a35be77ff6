"""Data-path oracle (oracle/data_ref.py) against vectors produced by the reference's own loader methods
(tests/golden/reference_data.npz, made by tests/golden/make_golden_data.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "golden"))

from oracle import data_ref                       # noqa: E402
import make_golden_data as mgd                    # noqa: E402  (frames(): the seeded synthetic depth frames)


def _cases():
    g = np.load(os.path.join(HERE, "golden", "reference_data.npz"))
    depth, com, cube = mgd.frames(np.random.RandomState(int(g["seed"])), len(g["com"]))
    assert np.array_equal(com, g["com"]) and np.array_equal(cube, g["cube"])
    return g, depth, com, cube


def test_crop_and_normalize_match_the_reference_run():
    g, depth, com, cube = _cases()
    for i in range(len(com)):
        norm, trans, crop = data_ref.crop_and_normalize(depth[i], com[i], cube[i], (128, 128), mgd.PARAS)
        assert np.array_equal(crop, g["crop"][i])                     # pixel selection: bit-exact
        assert np.array_equal(trans, g["trans"][i])
        assert np.abs(norm - g["norm"][i]).max() < 1e-6               # float32 vs NumPy-2 float64 in-place arithmetic
        assert norm.min() >= -1.0 and norm.max() <= 1.0


def test_resize_nearest_rule():
    src = np.arange(35, dtype=np.float32).reshape(5, 7)
    out = data_ref.resize_nearest(src, (4, 3))                        # (width, height)
    assert out.shape == (3, 4)
    assert np.array_equal(out, src[[0, 1, 3]][:, [0, 1, 3, 5]])
    assert np.array_equal(data_ref.resize_nearest(src, (7, 5)), src)


def _aug_cases():
    import make_golden_aug as mga
    g = np.load(os.path.join(HERE, "golden", "reference_aug.npz"))
    depth, com, cube = mgd.frames(np.random.RandomState(11), 12)
    dr = mga.draws(np.random.RandomState(23), 12)
    assert np.array_equal(np.stack([d[0] for d in dr]), g["off"])
    return g, depth, com, cube


def test_augment_crop_matches_the_reference_run():
    """oracle.data_ref.augment_crop against the reference's own augmentCrop (tests/golden/reference_aug.npz): 12 frames x
    the 4 modes.  'rot' / 'sc' / 'none': image pixels, joints, cube, centre and transform all equal.  'com': the reference
    ran under NumPy 2, whose float32 arithmetic on the moved centre can shift a crop bound by one pixel against the float64
    convention of the oracle -- cases with an equal transform must have equal pixels, and most cases must be such."""
    g, depth, com, cube = _aug_cases()
    same_M = 0
    n_com = 0
    for i in range(12):
        crop, trans = data_ref.crop_image_deep_pp(depth[i], com[i], cube[i], (128, 128), mgd.PARAS)
        for mode in range(4):
            k = i * 4 + mode
            name = data_ref.AUG_MODES[mode]
            img, j, cb, cm, M = data_ref.augment_crop(crop, g["joints_in"][i], com[i], cube[i], trans, name, g["off"][i], float(g["rot"][i]),
                                                     float(g["sc"][i]), mgd.PARAS)
            assert np.allclose(cb, g["cube"][k], rtol=1e-12) and np.allclose(cm, g["com"][k], rtol=1e-6, atol=1e-4), (i, name)
            if name == "com":
                n_com += 1
                if not np.allclose(M, g["M"][k], rtol=1e-9, atol=1e-9):
                    continue                                            # a one-pixel bound difference (NumPy 1 vs 2 promotion)
                same_M += 1
            else:
                assert np.allclose(M, g["M"][k], rtol=1e-12, atol=1e-12), (i, name)
            assert np.abs(j - g["joints"][k]).max() < 2e-3, (i, name, np.abs(j - g["joints"][k]).max())          # mm
            bad = np.abs(img - g["img"][k]) > 1e-5
            assert bad.mean() < (2e-3 if name == "rot" else 2e-4), (i, name, bad.sum())      # (rot: thin structures; else: ties of the last ulp of an inverse)
            assert img.min() >= -1.0 - 1e-6 and img.max() <= 1.0 + 1e-6
    assert same_M >= n_com * 2 // 3, (same_M, n_com)


def test_restated_opencv_warps_on_hand_checkable_cases():
    src = np.arange(36, dtype=np.float32).reshape(6, 6) + 1
    eye = np.eye(3)
    assert np.array_equal(data_ref.cv_warp_perspective_nn(src, eye, (6, 6)), src)
    shift = np.array([[1., 0, 2], [0, 1., 1], [0, 0, 1.]])              # dst(x, y) = src(x - 2, y - 1)
    out = data_ref.cv_warp_perspective_nn(src, shift, (6, 6), border=0.0)
    assert np.array_equal(out[1:, 2:], src[:-1, :-2]) and not out[0].any() and not out[:, :2].any()
    R = data_ref.cv_get_rotation_matrix_2d((3, 3), 90, 1)               # OpenCV: positive angle = counter-clockwise (image coordinates)
    assert np.allclose(R, [[0, 1, 0], [-1, 0, 6]], atol=1e-12)
    rot = data_ref.cv_warp_affine_nn(src, R, (6, 6), border=0.0)
    # R maps (sx, sy) -> (sy, 6 - sx), so dst[y][x] = src[x][6 - y]: a quarter turn; row 0 would read column 6 -> border
    assert not rot[0].any()
    for y in range(1, 6):
        for x in range(6):
            assert rot[y, x] == src[x, 6 - y]
    ident = data_ref.cv_warp_affine_nn(src, np.array([[1., 0, 0], [0, 1., 0]]), (6, 6))
    assert np.array_equal(ident, src)
    # fixed point with round_delta 1/2: a shift of exactly +0.5 reads floor(x - 0.5 + 0.5) = x; a little more reads x - 1
    half = data_ref.cv_warp_affine_nn(src, np.array([[1., 0, 0.5], [0, 1., 0]]), (6, 6))
    assert np.array_equal(half, src)
    more = data_ref.cv_warp_affine_nn(src, np.array([[1., 0, 0.51], [0, 1., 0]]), (6, 6), border=0.0)
    assert np.array_equal(more[:, 1:], src[:, :-1]) and not more[:, 0].any()
