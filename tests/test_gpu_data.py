"""Depth data path on the device (dsf_depth_crop_normalize) against the oracle and the reference-generated vectors."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _cases():
    import make_golden_data as mgd
    g = np.load(os.path.join(HERE, "golden", "reference_data.npz"))
    depth, com, cube = mgd.frames(np.random.RandomState(int(g["seed"])), len(g["com"]))
    return mgd, g, depth, com, cube


def test_crop_normalize_matches_reference_vectors():
    from dsf_amd.data.render_loader import loader
    mgd, g, depth, com, cube = _cases()
    L = loader(paras=mgd.PARAS)
    img, trans, raw = L.crop_normalize(torch.tensor(depth).cuda(), com, cube, want_raw=True)
    assert img.shape == (len(com), 1, 128, 128) and trans.dtype == torch.float64
    assert np.array_equal(raw.cpu().numpy(), g["crop"])                           # source-pixel selection: bit-exact
    assert np.array_equal(trans.cpu().numpy(), g["trans"])
    assert np.abs(img[:, 0].cpu().numpy() - g["norm"]).max() < 1e-6


def test_crop_normalize_matches_oracle_on_random_frames():
    from oracle import data_ref
    from dsf_amd import ops
    mgd, _, _, _, _ = _cases()
    depth, com, cube = mgd.frames(np.random.RandomState(123), 24)
    com[5] = [320.4, 239.6, 700.0]; cube[5] = [250.0, 250.0, 250.0]              # (centre of the image)
    img, trans, raw = ops.depth_crop_normalize(torch.tensor(depth).cuda(), com, cube, mgd.PARAS, 128, want_raw=True)
    for i in range(len(com)):
        n, t, c = data_ref.crop_and_normalize(depth[i], com[i], cube[i], (128, 128), mgd.PARAS)
        assert np.array_equal(raw[i].cpu().numpy(), c), i
        assert np.array_equal(trans[i].cpu().numpy(), t), i
        assert np.array_equal(img[i, 0].cpu().numpy(), n), i                       # same float32 operations: bit-exact
    # other output sizes and a shared cube
    img64, t64 = ops.depth_crop_normalize(torch.tensor(depth[:3]).cuda(), com[:3], [250.0, 250.0, 250.0], mgd.PARAS, 64)
    for i in range(3):
        n, t, _ = data_ref.crop_and_normalize(depth[i], com[i], [250.0, 250.0, 250.0], (64, 64), mgd.PARAS)
        assert np.array_equal(img64[i, 0].cpu().numpy(), n) and np.array_equal(t64[i].cpu().numpy(), t)


def test_crop_normalize_empty_batch_and_cpu_tensor():
    from dsf_amd import ops
    img, trans = ops.depth_crop_normalize(torch.zeros(0, 480, 640, device="cuda"), np.zeros((0, 3)), np.zeros((0, 3)),
                                          (588.03, 587.07, 320.0, 240.0))
    assert img.shape == (0, 1, 128, 128) and trans.shape == (0, 3, 3)
    with pytest.raises(RuntimeError):
        ops.depth_crop_normalize(torch.zeros(1, 480, 640), np.zeros((1, 3)), np.zeros((1, 3)), (588.03, 587.07, 320.0, 240.0))
