"""Every trainer step of the BASELINE configs on the HIP path against its CPU oracle composition (oracle/step_ref.py) on
the same weights, inputs and random draws, at B = 2..4:

  config 3  MeshLossStep       hourglass-2-stack + MANO head, m2d + ICP + part ICP + collision
  config 4  PretrainStep(3)    ResNet-50 two-stage, every sample rendered from 3 augmentView rotations
  config 5  FinetuneStageStep  Trainer.FinetuneStage (train_render.py:622-823) incl. the frozen transfer generator
  +         PretrainStep(1)    Trainer.Pretrain (:415-488), FinetuneStep  Trainer.Finetune (:490-620)

Bars: loss within 2e-3 relative, cosine of the whole gradient vector > 0.9995 (BatchNorm at B = 2..4 amplifies fp32
summation-order noise through ~40..110 layers; the bar is on the gradient DIRECTION and on its relative L2 error).
Then one full-size property run per config (B = 64 per GPU: finite terms, gradients reach every trunk, loss falls).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CAM = (588.03, 587.07, 320.0, 240.0)


@pytest.fixture(scope="module")
def render():
    from dsf_amd.render_model.mano_layer import Render
    return Render("synthetic", "nyu", CAM, (640, 480)).cuda()


@pytest.fixture(scope="module")
def orender(mano_dict):
    from oracle import step_ref
    return step_ref.OracleRender(mano_dict)


def _twin_pair(builder, *args, seed=3, heads=True, fit=None):
    """(torch.nn twin on the CPU, product net on the GPU) with identical weights; the MANO heads are biased to
    non-degenerate hands (``fit``: a (62,) row the heads reproduce, so that the render matches data made from it)."""
    from oracle import nets
    torch.manual_seed(seed)
    net_cpu = nets.build(builder, *args)
    if heads:
        with torch.no_grad():
            for name in ("mano_regress", "mano_regress_s2"):
                head = getattr(net_cpu, name, None)
                if head is None:
                    continue
                if fit is not None:
                    head[2].bias.copy_(fit)
                else:
                    head[2].bias[58] = 1.0
                    head[2].bias[3:48] = 0.2 * torch.randn(45)
                    head[2].bias[:3] = torch.tensor([0.3, -0.2, 0.1])
    net_gpu = builder(*args).cuda()
    net_gpu.load_state_dict(net_cpu.state_dict())
    return net_cpu, net_gpu


def _compare(loss_c, loss_g, net_cpu, net_gpu, cos_min=0.9995, l2_max=2e-2, loss_rtol=2e-3):
    lc, lg = float(loss_c.detach()), float(loss_g.detach())
    assert abs(lg - lc) <= loss_rtol * abs(lc), (lg, lc)
    num = den = dot = ng = 0.0
    n_with = 0
    for (n, pc), (_, pg) in zip(net_cpu.named_parameters(), net_gpu.named_parameters()):
        assert (pc.grad is None) == (pg.grad is None), n
        if pc.grad is None:
            continue
        n_with += 1
        ref, got = pc.grad.double(), pg.grad.cpu().double()
        num += float(((got - ref) ** 2).sum()); den += float((ref ** 2).sum())
        dot += float((got * ref).sum()); ng += float((got ** 2).sum())
    assert n_with > 0 and den > 0
    cos, rel = dot / (den * ng) ** 0.5, (num / den) ** 0.5
    assert cos > cos_min, (cos, rel)
    assert rel < l2_max, (cos, rel)
    return cos, rel


def _terms_close(tc, tg, rtol=5e-3, atol=2e-5):
    for k in tc:
        a, b = float(tc[k]), float(tg[k])
        assert abs(a - b) <= rtol * abs(a) + atol, (k, a, b)


def _real_batch(orender, B, seed, same_pose=None):
    """'real' depth = the oracle's render of an independent parameter draw; same image for both sides."""
    from dsf_amd.train_step import synthetic_batch
    pr, cr, cube_r = synthetic_batch(B, "cpu", seed=seed)
    if same_pose is not None:
        pr = same_pose.view(1, 62).expand(B, 62).contiguous()
    with torch.no_grad():
        img_r = orender.render(pr, cr, cube_r)[0]
        M_r = orender._crop_geometry(cr, cube_r)[2]
    return pr, cr, cube_r, img_r, torch.from_numpy(M_r)


# ------------------------------------------------------------------------------------------------
# config 3
# ------------------------------------------------------------------------------------------------
def test_config3_mesh_loss_step_vs_oracle(render, orender):
    from oracle import step_ref
    from dsf_amd.model.hourglass import PoseNetMANO
    from dsf_amd.train_step import MeshLossStep, synthetic_batch, Config
    B = 3
    p, c, cube = synthetic_batch(B, "cpu", seed=9)
    # the head starts near the target pose so that ICP / part ICP / m2d work on overlapping geometry
    fit = p[0].clone()
    net_cpu, net_gpu = _twin_pair(PoseNetMANO, 2, 21, seed=4, heads=False)
    with torch.no_grad():
        net_cpu.mano_regress[2].bias.copy_(fit + 0.03 * torch.randn(62, generator=torch.Generator().manual_seed(1)))
    net_gpu.load_state_dict(net_cpu.state_dict())
    p = fit.view(1, 62).expand(B, 62).contiguous()
    g = torch.Generator().manual_seed(11)
    keys = [torch.randint(0, 2 ** 31 - 1, (B, 128 * 128), dtype=torch.int32, generator=g) for _ in range(2)]
    tgt_c = step_ref.mesh_targets(orender, p, c, cube, keys[0], keys[1])
    # GPU targets through the product's own make_targets on the same keys: the discrete pieces must agree
    step = MeshLossStep(net_gpu, render, Config)
    tgt_g = step.make_targets(p.cuda(), c.cuda(), cube.cuda(), keys=(keys[0].cuda(), keys[1].cuda()))
    assert (tgt_g["seg"].cpu() != tgt_c["seg"]).float().mean() < 2e-3          # integer labels (inputs differ by ulps)
    assert (tgt_g["pcl"].cpu() - tgt_c["pcl"]).abs().max() < 1e-4 or \
        ((tgt_g["pcl"].cpu() - tgt_c["pcl"]).abs().amax(-1) > 1e-4).float().mean() < 2e-3
    assert (tgt_g["crop"].cpu() != tgt_c["crop"]).float().mean() < 1e-4
    # the step itself on IDENTICAL targets (the oracle's), so that only the step is compared
    tg = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in tgt_c.items() if k != "Minv"}
    loss_c, terms_c = step_ref.mesh_step_loss(net_cpu, orender, tgt_c, Config)
    loss_c.backward()
    loss_g, terms_g = step.loss(tg)
    loss_g.backward()
    _terms_close(terms_c, terms_g)
    assert float(terms_c["d2m"]) > 0 and float(terms_c["pd2m"]) > 0 and float(terms_c["m2d"]) > 0
    _compare(loss_c, loss_g, net_cpu, net_gpu)


# ------------------------------------------------------------------------------------------------
# Trainer.Pretrain and config 4 (multi-view, ResNet-50)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("backbone,views,B", [("ResNet_stage_18", 1, 2), ("ResNet_stage_50", 3, 2)])
def test_pretrain_and_config4_multiview_step_vs_oracle(render, orender, backbone, views, B):
    from oracle import step_ref, nets
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.render_model.transfer import define_G
    from dsf_amd.train_step import PretrainStep, synthetic_batch, draws_to, Config
    net_cpu, net_gpu = _twin_pair(MANO_OCR_stage, backbone, 21, True, seed=5)
    torch.manual_seed(8)
    gen_cpu = nets.build(define_G, 1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier').eval()
    gen_gpu = define_G(1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier').cuda()
    gen_gpu.load_state_dict(gen_cpu.state_dict())
    step = PretrainStep(net_gpu, render, gen_gpu, Config, views=views)
    p, _, cube = synthetic_batch(B, "cpu", seed=31)
    d = step.draw(B, "cpu", torch.Generator().manual_seed(32), np.random.default_rng(33))
    assert d["aug_view"].shape == (B * views, 3) and (views == 1) == bool((d["aug_view"] == 0).all())
    loss_c = step_ref.pretrain_loss(net_cpu, orender, gen_cpu, p, cube, d, Config, views=views)
    loss_c.backward()
    loss_g, terms = step.loss(p.cuda(), cube.cuda(), draws_to(d, "cuda"))
    loss_g.backward()
    assert all(torch.isfinite(v) for v in terms.values())
    _compare(loss_c, loss_g, net_cpu, net_gpu)


# ------------------------------------------------------------------------------------------------
# Trainer.Finetune and config 5 (Trainer.FinetuneStage)
# ------------------------------------------------------------------------------------------------
def _selfsup_setup(orender, B, fitted):
    from oracle import nets
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.render_model.transfer import define_G
    from dsf_amd.train_step import synthetic_batch
    p, _, cube = synthetic_batch(B, "cpu", seed=21)
    pose = synthetic_batch(1, "cpu", seed=23)[0][0]
    pr, cr, cube_r, img_r, M_r = _real_batch(orender, B, 22, same_pose=pose if fitted else None)
    # fitted: both MANO heads reproduce the pose the real images were made from (+ a small error), so that the render
    # agrees with the data and the M2P selection (depth < 0.04, ICP < 1e-3, part ICP < 1e-3) is not empty
    fit = (pose + 0.01 * torch.randn(62, generator=torch.Generator().manual_seed(2))) if fitted else None
    net_cpu, net_gpu = _twin_pair(MANO_OCR_stage, "ResNet_stage_18", 21, True, seed=6, fit=fit)
    torch.manual_seed(8)
    gen_cpu = nets.build(define_G, 1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier').eval()
    gen_gpu = define_G(1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier').cuda()
    gen_gpu.load_state_dict(gen_cpu.state_dict())
    return p, cube, img_r, cr, cube_r, M_r, net_cpu, net_gpu, gen_cpu, gen_gpu


@pytest.mark.parametrize("fitted", [False, True])
def test_config5_finetune_stage_step_vs_oracle(render, orender, fitted):
    from oracle import step_ref
    from dsf_amd.train_step import FinetuneStageStep, draws_to, Config
    B = 2
    p, cube, img_r, cr, cube_r, M_r, net_cpu, net_gpu, gen_cpu, gen_gpu = _selfsup_setup(orender, B, fitted)
    step = FinetuneStageStep(net_gpu, render, gen_gpu, Config)
    d = step.draw(B, "cpu", torch.Generator().manual_seed(41), np.random.default_rng(42 + int(fitted)))
    assert 3 <= d["mask_joint_id"].numel() <= 9                       # the reference's occluder count range (:1328)
    loss_c, terms_c = step_ref.finetune_stage_loss(net_cpu, orender, gen_cpu, p, cube, img_r, cr, cube_r, d, Config)
    loss_c.backward()
    loss_g, terms_g = step.loss(p.cuda(), cube.cuda(), img_r.cuda(), cr.cuda(), cube_r.cuda(), M_r.cuda(),
                                draws=draws_to(d, "cuda"))
    loss_g.backward()
    if fitted:
        assert float(terms_c["M2P"]) > 0, "the fitted case must exercise the M2P selection"
    _terms_close(terms_c, terms_g)
    _compare(loss_c, loss_g, net_cpu, net_gpu)
    assert all(q.grad is None for q in gen_gpu.parameters())          # frozen generator


@pytest.mark.parametrize("fitted", [False, True])
def test_finetune_single_stage_step_vs_oracle(render, orender, fitted):
    from oracle import step_ref
    from dsf_amd.train_step import FinetuneStep, draws_to, Config
    B = 2
    p, cube, img_r, cr, cube_r, M_r, net_cpu, net_gpu, gen_cpu, gen_gpu = _selfsup_setup(orender, B, fitted)
    step = FinetuneStep(net_gpu, render, gen_gpu, Config)
    d = step.draw(B, "cpu", torch.Generator().manual_seed(51), np.random.default_rng(52))
    loss_c, terms_c = step_ref.finetune_loss(net_cpu, orender, gen_cpu, p, cube, img_r, cr, cube_r, d, Config)
    loss_c.backward()
    loss_g, terms_g = step.loss(p.cuda(), cube.cuda(), img_r.cuda(), cr.cuda(), cube_r.cuda(), M_r.cuda(), draws_to(d, "cuda"))
    loss_g.backward()
    if fitted:
        assert float(terms_c["M2P"]) > 0
    _terms_close(terms_c, terms_g)
    _compare(loss_c, loss_g, net_cpu, net_gpu)


# ------------------------------------------------------------------------------------------------
# full-size property runs (per-GPU share of each config)
# ------------------------------------------------------------------------------------------------
def test_config3_full_size_properties(render):
    from dsf_amd.model.hourglass import PoseNetMANO
    from dsf_amd.train_step import MeshLossStep, synthetic_batch, Config
    torch.manual_seed(0)
    net = PoseNetMANO(2, 21).cuda()
    step = MeshLossStep(net, render, Config)
    p, c, cube = synthetic_batch(64, "cuda", seed=9)
    tgt = step.make_targets(p, c, cube)
    assert tgt["joint_pcl"].shape == (64, 2048, 3) and 0 <= int(tgt["seg"].min()) and int(tgt["seg"].max()) <= 15
    l0, terms = step(tgt)
    hist = [float(step(tgt)[0]) for _ in range(30)]
    assert all(np.isfinite(hist)) and min(hist[-10:]) < float(l0)
    assert all(torch.isfinite(v) for v in terms.values())
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for n, q in net.named_parameters()
               if not n.startswith("body.outs_") and "merge" not in n)


def test_config4_full_size_properties(render):
    """Per-GPU share of config 4: 64 samples x 3 views = 192 meshes / images through ResNet-50 two-stage."""
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.train_step import PretrainStep, synthetic_batch, Config
    torch.manual_seed(0)
    net = MANO_OCR_stage("ResNet_stage_50", 21, True).cuda()
    step = PretrainStep(net, render, None, Config, views=3)
    p, _, cube = synthetic_batch(64, "cuda", seed=3)
    g = torch.Generator(device="cuda").manual_seed(4)
    d = step.draw(64, "cuda", g, np.random.default_rng(4))
    s = step.synth(p.repeat_interleave(3, 0), cube.repeat_interleave(3, 0), d)
    assert s["img"].shape == (192, 1, 128, 128) and float((s["img"] < 0.99).float().mean()) > 0.02
    # the three views of a sample are different images of one pose
    assert (s["img"].view(64, 3, -1)[:, 0] != s["img"].view(64, 3, -1)[:, 1]).any()
    l0, terms = step(p, cube, d)
    l1, _ = step(p, cube, d)
    l2, _ = step(p, cube, d)
    assert all(torch.isfinite(v) for v in terms.values()) and torch.isfinite(l2)
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in net.parameters())


def test_config5_full_size_properties(render):
    """Per-GPU share of config 5: B = 64 synthetic + 64 real images through the whole FinetuneStage step."""
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.render_model.transfer import define_G
    from dsf_amd.train_step import FinetuneStageStep, synthetic_batch, Config
    from dsf_amd import ops
    torch.manual_seed(1)
    net = MANO_OCR_stage("ResNet_stage_18", 21, True).cuda()
    with torch.no_grad():
        for head in (net.mano_regress[2], net.mano_regress_s2[2]):
            head.bias[58] = 1.0
    gen = define_G(1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier').cuda()
    step = FinetuneStageStep(net, render, gen, Config)
    B = 64
    p, c, cube = synthetic_batch(B, "cuda", seed=21)
    pr, cr, cube_r = synthetic_batch(B, "cuda", seed=22)
    with torch.no_grad():
        img_r = render.render(pr, cr, cube_r)[0]
        _, M_r, _, _ = ops.crop_setup(cr, cube_r, render.cam, 128)
    g = torch.Generator(device="cuda").manual_seed(5)
    loss, terms = step(p, cube, img_r, cr, cube_r, M_r, generator=g)
    assert torch.isfinite(loss) and all(torch.isfinite(v) for v in terms.values())
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in net.parameters())
    assert net.layer1[0].conv1.weight.grad.abs().sum() > 0 and net.layer4_s2[1].conv2.weight.grad.abs().sum() > 0
    assert all(q.grad is None for q in gen.parameters())
    loss2, _ = step(p, cube, img_r, cr, cube_r, M_r, generator=g)
    assert torch.isfinite(loss2)
