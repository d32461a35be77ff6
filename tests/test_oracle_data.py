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
