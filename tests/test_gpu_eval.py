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


def test_mask_img_matches_reference_golden_with_explicit_draws():
    """Render.mask_img (mano_layer.py:1326-1340): the reference's output for recorded random draws."""
    from dsf_amd.render_model.mano_layer import Render
    g = np.load(os.path.join(HERE, "golden", "reference_eval.npz"))
    render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
    T = lambda a: torch.tensor(a, device="cuda")
    out = render.mask_img(T(g["mask_img"]), T(g["mask_juvd"]), 0.15, 0.3,
                          draws=(list(g["mask_joint_id"]), T(g["mask_offset"]), T(g["mask_radius"])))
    assert np.array_equal(out.cpu().numpy(), g["mask_out"])


def test_native_networks_reproduce_reference_outputs():
    """Hourglass and CycleGAN generator on the HIP convolutions, weights taken from the torch.nn twin built with the
    seed the reference used for tests/golden/reference_nets.npz: outputs match the REFERENCE's recorded outputs."""
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.model.hourglass import PoseNet
    from dsf_amd.render_model.transfer import define_G
    g = np.load(os.path.join(HERE, "golden", "reference_nets.npz"))
    x = torch.tensor(g["x"]).cuda()

    from oracle import nets

    def twin(build):
        torch.manual_seed(7)
        cpu = nets.build(build)                                    # torch.nn twin initialised with the reference's seed
        gpu = build().cuda()
        gpu.load_state_dict(cpu.state_dict())
        return gpu.eval()

    net = twin(lambda: MANO_OCR_stage("ResNet_stage_18", 21, False))
    with torch.no_grad():
        (pix, par), = net(x)
    assert np.abs(pix.cpu().numpy()[:, :, ::8, ::8] - g["r18_pix_sub"]).max() < 2e-3 * max(1.0, np.abs(g["r18_pix_sub"]).max())
    assert np.abs(par.cpu().numpy() - g["r18_par"]).max() < 2e-3 * max(1.0, np.abs(g["r18_par"]).max())

    hg = twin(lambda: PoseNet(2, 21))
    flat = []

    def _flat(o):
        if isinstance(o, (list, tuple)):
            for q in o:
                _flat(q)
        else:
            flat.append(o)
    with torch.no_grad():
        _flat(hg(x))
    for i, o in enumerate(flat):
        want = g["hg_out%d_sub" % i]
        assert tuple(o.shape) == tuple(g["hg_out%d_shape" % i])
        assert np.abs(o.cpu().numpy()[:, ::8, ::4, ::4] - want).max() < 2e-3 * max(1.0, np.abs(want).max())

    gen = twin(lambda: define_G(1, 1, 64, "resnet_9blocks", "instance", False, "xavier"))
    with torch.no_grad():
        go = gen(x)
    assert np.abs(go.cpu().numpy()[:, :, ::4, ::4] - g["gen_out_sub"]).max() < 2e-3


def test_render_forward_tuple_is_self_consistent():
    """Render.forward (mano_layer.py:983-1039): 8-tuple; the image equals mesh2img of the returned mesh, the uvd
    outputs equal the transform of the xyz outputs, the rotation augmentation preserves distances to the centre."""
    from dsf_amd.render_model.mano_layer import Render
    from dsf_amd.data.render_loader import loader
    from dsf_amd.train_step import synthetic_batch
    render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
    p, c, cube = synthetic_batch(4, "cuda", seed=2)
    view = torch.tensor([[0.3, -0.2, 0.5]] * 4, device="cuda")
    out = render(p, c, cube, augmentView=view, mask=False)
    assert len(out) == 8
    img, j_uvd, v_uvd, j_xyz, v_xyz, center, cube_o, M = out
    assert img.shape == (4, 1, 128, 128) and j_uvd.shape == (4, 21, 3) and v_uvd.shape == (4, 779, 3) and M.shape == (4, 3, 3)
    world = v_xyz * cube_o.unsqueeze(1) / 2 + center.unsqueeze(1)
    again = render.mesh2img(world.contiguous(), center, cube_o)     # vertices round-tripped through the normalisation: a few
    assert ((again - img).abs() > 1e-4).float().mean() < 5e-3       # boundary pixels may flip
    L = loader()
    assert (L.xyz_nl2uvdnl_tensor(j_xyz, center, M, cube_o) - j_uvd).abs().max() < 1e-4
    plain = render(p, c, cube, mask=False)
    d0 = (plain[4] * cube.unsqueeze(1) / 2).norm(dim=-1)            # distances of the vertices to the crop centre
    d1 = (v_xyz * cube_o.unsqueeze(1) / 2).norm(dim=-1)
    assert (d0 - d1).abs().max() < 1e-2


def test_native_discriminator_matches_reference_output_and_trains():
    from dsf_amd.render_model import transfer as Tr
    g = np.load(os.path.join(HERE, "golden", "reference_eval.npz"))
    x = torch.tensor(g["D_x"]).cuda()
    torch.manual_seed(11)
    from oracle import nets
    cpu = nets.build(Tr.define_D, 1, 64, "basic", 3, "instance", "normal", 0.02)
    D = Tr.define_D(1, 64, "basic", 3, "instance", "normal", 0.02).cuda()
    D.load_state_dict(cpu.state_dict())
    with torch.no_grad():
        o = D(x)
    assert np.abs(o.cpu().numpy() - g["D_basic_out"]).max() < 2e-3
    # one lsgan discriminator update runs end to end on the HIP convolutions
    crit = Tr.GANLoss("lsgan").cuda()
    opt = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    l0 = None
    for _ in range(5):
        opt.zero_grad()
        loss = 0.5 * (crit(D(x), True) + crit(D(-x), False))
        loss.backward()
        opt.step()
        l0 = float(loss.detach()) if l0 is None else l0
    assert np.isfinite(float(loss.detach())) and float(loss.detach()) < l0
