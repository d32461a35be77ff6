"""Evaluation path: device-side xyz2error vs the reference's own outputs (golden) and the whole
test_iter vs a composition of oracle pieces; checkpoint key compatibility."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)


def test_xyz2error_matches_reference_golden():
    from dsf_amd.eval_step import xyz2error
    g = np.load(os.path.join(HERE, "golden", "reference_eval.npz"))
    T = lambda a: torch.tensor(a, device="cuda")
    for ds in ("nyu", "msra", "icvl"):
        a = (T(g[ds + "_pred"]), T(g[ds + "_gt"]), T(g[ds + "_center"]), T(g[ds + "_cube"]))
        assert abs(float(xyz2error(*a, dataset=ds)) - float(g[ds + "_err"])) < 1e-3
        assert np.allclose(xyz2error(*a, dataset=ds, keep_batch=True).cpu().numpy(), g[ds + "_err_batch"], atol=1e-3)
        assert np.allclose(xyz2error(*a, dataset=ds, keep_joint=True).cpu().numpy(), g[ds + "_err_joint"], atol=1e-3)


def test_test_iter_vs_oracle_composition_and_checkpoint_keys():
    from dsf_amd.eval_step import EvalStep
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.render_model.mano_layer import Render
    from dsf_amd.train_step import synthetic_batch, Config
    from dsf_amd import ops
    from oracle import eval_ref, image_ref
    torch.manual_seed(0)
    net = MANO_OCR_stage("ResNet_stage_18", 21, True).cuda()
    # reference checkpoints load: identical state-dict key list (tests/golden/reference_nets.npz, made from the reference)
    keys = list(np.load(os.path.join(HERE, "golden", "reference_nets.npz"))["r18s2_keys"])
    assert list(net.state_dict().keys()) == keys
    net.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
    render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
    B = 6
    p, c, cube = synthetic_batch(B, "cuda", seed=5)
    with torch.no_grad():
        img, juvd, jxyz, mesh = render.render(p, c, cube)
        _, M, _, _ = ops.crop_setup(c, cube, render.cam, 128)
    ev = EvalStep(net, render, Config)
    for _ in range(2):                                         # a few training-mode forwards to move the BN statistics
        net.train(); net(img, render, c, cube)
    net.eval()
    errs = [float(e) for e in ev.test_iter(img, jxyz[:, render.mano_layer.transfer], c, cube, M)]
    assert len(errs) == 4 and all(np.isfinite(errs))
    # recompute from the raw network outputs with the oracle's pieces
    with torch.no_grad():
        outs = net(img, render, c, cube)
    gt = jxyz[:, render.mano_layer.transfer].cpu().numpy()
    n = gt.shape[1]
    Minv = torch.inverse(M.cpu()).numpy()
    want = []
    for pix, mano in outs:
        uvd = image_ref.offset_maps_to_joints(pix.cpu().float().contiguous(), img.cpu(), 0.8).numpy()
        xyz = image_ref.uvd_to_xyz(uvd, c.cpu().numpy(), Minv, cube.cpu().numpy(), normalise=True)
        want.append(eval_ref.xyz_to_error(eval_ref.select_eval_joints(xyz, render.mano_layer.transfer), gt[:, :n - 1],
                                          c.cpu().numpy(), cube.cpu().numpy()))
        mj, _ = render.get_mesh_xyz(mano)
        want.append(eval_ref.xyz_to_error(eval_ref.select_eval_joints(mj.cpu().numpy(), render.mano_layer.transfer), gt[:, :n - 1],
                                          c.cpu().numpy(), cube.cpu().numpy()))
    for a, b in zip(errs, want):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (errs, want)
    mean, per = ev.test([(img, jxyz[:, render.mano_layer.transfer], juvd, c, M, cube)] * 2)
    assert abs(mean - sum(errs) / 4) < 1e-3 and net.training is False
