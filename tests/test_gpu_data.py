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


def test_crop_normalize_reads_raw_uint16_frames():
    """SURVEY 8(f) row 1 names raw 16-bit depth as the input: the sensors' uint16 millimetre frames (what nyu_reader /
    icvl_reader decode before their float32 cast, render_loader.py:201-218) go to the kernel as they are -- same bits out as
    through the float32 frame, against the oracle too."""
    from oracle import data_ref
    from dsf_amd import ops
    mgd, _, _, _, _ = _cases()
    depth, com, cube = mgd.frames(np.random.RandomState(7), 12)
    d16 = np.clip(np.rint(depth), 0, 65535).astype(np.uint16)                     # integer millimetres, as a sensor delivers them
    ref = ops.depth_crop_normalize(torch.tensor(d16.astype(np.float32)).cuda(), com, cube, mgd.PARAS, 128, want_raw=True)
    for dt in (torch.uint16, torch.int16):
        t16 = torch.from_numpy(d16.view(np.int16)).cuda().view(dt)
        got = ops.depth_crop_normalize(t16, com, cube, mgd.PARAS, 128, want_raw=True)
        for a, b in zip(got, ref):
            assert torch.equal(a, b), dt
    for i in range(len(com)):
        n, t, c = data_ref.crop_and_normalize(d16[i].astype(np.float32), com[i], cube[i], (128, 128), mgd.PARAS)
        assert np.array_equal(ref[2][i].cpu().numpy(), c) and np.array_equal(ref[0][i, 0].cpu().numpy(), n), i


def test_crop_normalize_empty_batch_and_cpu_tensor():
    from dsf_amd import ops
    img, trans = ops.depth_crop_normalize(torch.zeros(0, 480, 640, device="cuda"), np.zeros((0, 3)), np.zeros((0, 3)),
                                          (588.03, 587.07, 320.0, 240.0))
    assert img.shape == (0, 1, 128, 128) and trans.shape == (0, 3, 3)
    with pytest.raises(RuntimeError):
        ops.depth_crop_normalize(torch.zeros(1, 480, 640), np.zeros((1, 3)), np.zeros((1, 3)), (588.03, 587.07, 320.0, 240.0))


def test_eval_from_raw_frames_equals_eval_from_oracle_crops():
    """EvalStep.test_frames (device crop -> test_iter) against test_iter fed with the oracle's crops, transform and the
    reference's jointImgTo3D centre (render_loader.py:290-301)."""
    from oracle import data_ref
    from dsf_amd.eval_step import EvalStep
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.render_model.mano_layer import Render
    from dsf_amd.train_step import Config
    mgd, _, _, _, _ = _cases()
    depth, com, cube = mgd.frames(np.random.RandomState(3), 4)
    torch.manual_seed(0)
    net = MANO_OCR_stage("ResNet_stage_18", 21, True).cuda().eval()
    render = Render("synthetic", "nyu", mgd.PARAS, (640, 480)).cuda()
    ev = EvalStep(net, render, Config, dataset="nyu")
    xyz_gt = torch.randn(4, 21, 3, device="cuda") * 0.3
    e_dev = torch.stack(ev.test_frames(torch.tensor(depth).cuda(), com, cube, xyz_gt, mgd.PARAS))
    crops, Ms, centers = [], [], []
    fx, fy, fu, fv = mgd.PARAS
    for i in range(4):
        n, t, _ = data_ref.crop_and_normalize(depth[i], com[i], cube[i], (128, 128), mgd.PARAS)
        crops.append(n); Ms.append(t)
        u = com[i].astype(np.float32)
        centers.append([(u[0] - np.float32(fu)) * u[2] / np.float32(fx), (u[1] - np.float32(fv)) * u[2] / np.float32(fy), u[2]])
    T = lambda a, dt=torch.float32: torch.tensor(np.asarray(a), dtype=dt, device="cuda")
    e_ref = torch.stack(ev.test_iter(T(crops).unsqueeze(1), xyz_gt, T(centers), T(cube), T(Ms)))
    assert torch.isfinite(e_dev).all()
    assert torch.allclose(e_dev, e_ref, rtol=1e-4, atol=1e-3), (e_dev, e_ref)


def test_crop_normalize_adversarial_boxes():
    """Crop boxes entirely outside the frame (all background), larger than the frame (hand close to the camera), frames of
    another size, an all-zero frame: device == oracle bit for bit."""
    from oracle import data_ref
    from dsf_amd import ops
    rng = np.random.RandomState(99)
    H, W = 240, 320
    paras = (294.0, 293.5, 160.0, 120.0)
    depth = (rng.uniform(300, 1500, (6, H, W)) * (rng.uniform(size=(6, H, W)) > 0.1)).astype(np.float32)
    depth[5] = 0.0
    com = np.array([[-400.0, 100.0, 600.0],        # far left of the frame
                    [160.2, 119.7, 120.0],          # 12 cm from the camera: box larger than the frame
                    [319.0, 239.0, 900.0],          # bottom-right corner
                    [10.5, 200.25, 1400.0],
                    [160.0, 120.0, 700.0],
                    [100.0, 100.0, 650.0]])         # all-zero frame
    cube = np.array([[250.0] * 3, [250.0] * 3, [300.0] * 3, [180.0, 220.0, 200.0], [250.0, 250.0, 60.0], [250.0] * 3])
    img, trans, raw = ops.depth_crop_normalize(torch.tensor(depth).cuda(), com, cube, paras, 128, want_raw=True)
    for i in range(6):
        n, t, c = data_ref.crop_and_normalize(depth[i], com[i], cube[i], (128, 128), paras)
        assert np.array_equal(raw[i].cpu().numpy(), c), i
        assert np.array_equal(trans[i].cpu().numpy(), t), i
        assert np.array_equal(img[i, 0].cpu().numpy(), n), i
    assert float(img[0].min()) == 1.0 and float(img[5].max()) == 1.0           # nothing but background -> far plane


def test_training_augmentation_vs_oracle_and_reference_run():
    """SURVEY 8f row 1, training phase: ``loader.augmentCrop`` on the device (dsf_depth_augment_crop) for 12 frames x the 4
    modes ['rot', 'com', 'sc', 'none'], against the numpy oracle on the same inputs -- pixel selection, thresholds and the
    transform bit-exact for 'com' / 'sc' / 'none', 'rot' up to last-ulp differences of the device's cos / sin (a handful of
    pixels on thin structures) -- and against the vectors the REFERENCE's own augmentCrop produced
    (tests/golden/reference_aug.npz, tests/golden/make_golden_aug.py)."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden_data as mgd
    from oracle import data_ref
    from dsf_amd.data.render_loader import loader
    g = np.load(os.path.join(here, "golden", "reference_aug.npz"))
    depth, com, cube = mgd.frames(np.random.RandomState(11), 12)
    L = loader()
    dev = torch.device("cuda")
    _, trans, raw = L.crop_normalize(torch.tensor(depth, device=dev), com, cube, mgd.PARAS, want_raw=True)
    crops, Ms = raw.cpu().numpy(), trans.cpu().numpy()
    # every (frame, mode) pair as one batch of 48
    idx = np.repeat(np.arange(12), 4)
    mode = np.tile(np.arange(4), 12).astype(np.int32)
    img, j, cb, cm, M = L.augmentCrop(raw[idx], torch.tensor(g["joints_in"][idx], device=dev), com[idx], cube[idx], trans[idx], mode,
                                      g["off"][idx], g["rot"][idx], g["sc"][idx], mgd.PARAS)
    img, j, cb, cm, M = (t.cpu().numpy() for t in (img, j, cb, cm, M))
    same_M = n_com = 0
    for k in range(48):
        i, name = idx[k], data_ref.AUG_MODES[mode[k]]
        oi, oj, ocb, ocm, oM = data_ref.augment_crop(crops[i], g["joints_in"][i], com[i], cube[i], Ms[i], name, g["off"][i], float(g["rot"][i]),
                                                    float(g["sc"][i]), mgd.PARAS)
        # --- oracle on the same inputs ---
        assert np.array_equal(cb[k], ocb) and np.allclose(cm[k], np.asarray(ocm, dtype=np.float64), rtol=0, atol=0), (k, name)
        assert np.array_equal(M[k], oM), (k, name)
        bad = img[k, 0] != oi
        if name == "rot":
            assert bad.mean() < 1e-3, (k, bad.sum())
            assert np.abs(j[k] - oj).max() < 1e-3                          # mm
        else:
            assert not bad.any(), (k, name, bad.sum())
            assert np.array_equal(j[k], oj), (k, name)
        # --- the reference's own run (NumPy 2: 'com' bounds may sit one pixel off, see tests/test_oracle_data.py) ---
        if name == "com":
            n_com += 1
            if not np.allclose(M[k], g["M"][k], rtol=1e-9, atol=1e-9):
                continue
            same_M += 1
        assert (np.abs(img[k, 0] - g["img"][k]) > 1e-5).mean() < 2e-3, (k, name)
        assert np.abs(j[k] - g["joints"][k]).max() < 2e-3
    assert same_M >= n_com * 2 // 3
    # empty frame and empty batch
    z = torch.zeros(1, 128, 128, device=dev)
    out = L.augmentCrop(z, torch.zeros(1, 14, 3, device=dev), com[:1], cube[:1], trans[:1], [1], g["off"][:1], g["rot"][:1], g["sc"][:1], mgd.PARAS)
    assert torch.equal(out[0], torch.ones_like(out[0])) and torch.equal(out[1], torch.zeros(1, 14, 3, device=dev))
    e = L.augmentCrop(z[:0], torch.zeros(0, 14, 3, device=dev), com[:0], cube[:0], trans[:0], [], g["off"][:0], g["rot"][:0], g["sc"][:0], mgd.PARAS)
    assert e[0].shape == (0, 1, 128, 128)
    with pytest.raises(RuntimeError):
        L.augmentCrop(z.cpu(), torch.zeros(1, 14, 3), com[:1], cube[:1], trans[:1], [1], g["off"][:1], g["rot"][:1], g["sc"][:1], mgd.PARAS)
