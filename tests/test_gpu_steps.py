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
                    head[2].weight.mul_(0.05)              # the head stays within ~0.01 of `fit` for any input
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


def _bridge_flips(net_cpu, net_gpu, orender, render, img, center, cube):
    """Crop pixels of the stage-2 bridge render (backbone.py:304) that differ between the two sides.  The crop rasteriser
    is bit-exact for identical vertices (tests/test_gpu_parity.py), but the fused MANO kernel and the oracle's torch
    formulation round differently (verts agree to ~1e-7), so a pixel centre within an ulp of a silhouette edge can fall
    on the other side.  Each such pixel switches 84 offset-map values of the fusion layer's input between 0 and O(1):
    a discrete input difference, not an arithmetic one -- the comparison bars below are strict when there is none."""
    with torch.no_grad():
        mano_c = net_cpu._run_trunk(net_cpu.pre(img), '')[3]
        mano_g = net_gpu._run_trunk(net_gpu.pre(img.cuda()), '')[3]
        ic = orender.render(mano_c, center, cube)[0]
        ig = render.render(mano_g, center.cuda(), cube.cuda())[0].cpu()
    return int(((ic - ig).abs() > 1e-4).sum())


class _Recording:
    """wraps a renderer (oracle side): records the image of every ``render()`` call"""

    def __init__(self, inner):
        self.__dict__["inner"], self.__dict__["images"] = inner, []

    def __getattr__(self, k):
        return getattr(self.inner, k)

    def render(self, *a, **k):
        out = self.inner.render(*a, **k)
        self.images.append(out[0].detach().clone())
        return out


class _PinnedBridge:
    """wraps the product renderer: ``render()`` returns everything from the HIP path except that the IMAGE of call i is
    replaced by ``images[i]`` (the oracle's).  Only used for the stage-2 bridge inside ``MANO_OCR_stage.forward``, where the
    image is a gradient-free input of ``joint2offset`` (its backward returns no image gradient): pinning it removes the
    silhouette-pixel flips described at ``_bridge_flips`` and nothing else."""

    def __init__(self, inner, images):
        self.__dict__.update(inner=inner, images=list(images), calls=0)

    def __getattr__(self, k):
        return getattr(self.inner, k)

    def __call__(self, *a, **k):
        return self.inner(*a, **k)

    def render(self, *a, **k):
        out = self.inner.render(*a, **k)
        img = self.images[self.calls].to(out[0].device)
        self.__dict__["calls"] += 1
        return (img,) + tuple(out[1:])


class _Net64:
    """The CPU twin with its convolutional trunk evaluated in FLOAT64 (inputs cast to double at the trunk's entry points,
    outputs back to float32 for the fp32 geometry oracle).  Two-stage ResNet-50 at B = 2..6 is ill-conditioned: torch's own
    fp32 and fp64 CPU gradients agree only to cosine ~0.997 (BatchNorm over <= 128 values per channel, ~110 layers), so an
    fp32-vs-fp32 comparison measures rounding luck.  The bar for such nets: the HIP path must be as close to the float64
    gradients as torch's fp32 CPU path is."""

    def __init__(self, net):
        import copy
        self.net = copy.deepcopy(net).double()
        self.refine = net.refine
        for q in self.net.parameters():
            q.grad = None

    def pre(self, x):
        return self.net.pre(x.double())

    def fusion(self, x):
        return self.net.fusion(x.double())

    def _run_trunk(self, x, suffix):
        c4, feat, pix, mano = self.net._run_trunk(x.double(), suffix)
        return c4, feat, pix.float(), mano.float()

    def named_parameters(self):
        return self.net.named_parameters()


def _grad_error(ref_net, other_net):
    """(cosine, relative L2 error) of other's gradient vector against ref's, in float64"""
    num = den = dot = ng = 0.0
    for (n, pr), (_, po) in zip(ref_net.named_parameters(), other_net.named_parameters()):
        if pr.grad is None:
            continue
        ref, got = pr.grad.double().cpu(), po.grad.double().cpu()
        num += float(((got - ref) ** 2).sum()); den += float((ref ** 2).sum())
        dot += float((got * ref).sum()); ng += float((got ** 2).sum())
    return dot / (den * ng) ** 0.5, (num / den) ** 0.5


def _terms_close(tc, tg, rtol=5e-3, atol=2e-5):
    for k in tc:
        a, b = (float(v.detach()) if torch.is_tensor(v) else float(v) for v in (tc[k], tg[k]))
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
    # the two renders start from MANO vertices that differ by ulps: same pixels kept, depths equal to 1e-4 (north star)
    assert ((tgt_g["crop"].cpu() - tgt_c["crop"]).abs() > 1e-4).float().mean() < 1e-3
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
def _freeze_statistics(net_cpu, net_gpu, orender, p, cube, d, views, gen=None):
    """Frozen-statistics BatchNorm for both twins: the running statistics become the batch statistics of ONE training-mode
    pass of the CPU twin over the step's own images (momentum 1), then both networks go to evaluation mode.  The network is
    then normalised as in training, but no BatchNorm couples the 6..24 samples of the batch any more -- the well-conditioned
    variant of a case (the ill-conditioning of two-stage ResNet-50 at tiny batches is the batch statistics' Jacobian)."""
    from oracle import step_ref
    import torch.nn as nn
    bns = [m for m in net_cpu.modules() if isinstance(m, nn.BatchNorm2d)]
    saved = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    with torch.no_grad():
        pp, cc = (p.repeat_interleave(views, 0), cube.repeat_interleave(views, 0)) if views > 1 else (p, cube)
        s_c = step_ref.synth_pass(orender, gen, pp, cc, d, True)
        net_cpu.train()
        step_ref.net_forward(net_cpu, s_c["img_t"], orender, s_c["center"], s_c["cube"])
    for m, mom in zip(bns, saved):
        m.momentum = mom
    net_gpu.load_state_dict(net_cpu.state_dict())
    net_cpu.eval(); net_gpu.eval()


@pytest.mark.parametrize("backbone,views,B,refine,frozen", [("ResNet_stage_18", 1, 2, True, False), ("ResNet_stage_50", 3, 2, False, False),
                                                            ("ResNet_stage_50", 3, 2, True, False), ("ResNet_stage_50", 3, 2, True, True)])
def test_pretrain_and_config4_multiview_step_vs_oracle(render, orender, backbone, views, B, refine, frozen):
    """Trainer.Pretrain (views 1) and BASELINE config 4 (ResNet-50, 3 views per sample).  Strict bars for the two-stage
    ResNet-18 and for the ONE-stage ResNet-50 (Bottleneck trunk + multi-view render + loss list).  The two-stage ResNet-50 is
    compared through float64: besides being ill-conditioned (_Net64), its stage-2 input is a DISCONTINUOUS function of the
    stage-1 estimate -- ``joint2offset`` masks the offset maps at ``heat >= 0`` (generateFeature.py:31-34, unit vectors jump
    0 <-> O(1) on the kernel-radius circle) and the bridge render switches silhouette pixels -- so two fp32 evaluations that
    differ by rounding feed stage 2 different inputs.  The bar there: loss within 2e-3 of the float64-trunk oracle, gradient
    error against it within a small multiple of the CPU-fp32 oracle's own error."""
    from oracle import step_ref, nets
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.render_model.transfer import define_G
    from dsf_amd.train_step import PretrainStep, synthetic_batch, draws_to, Config
    net_cpu, net_gpu = _twin_pair(MANO_OCR_stage, backbone, 21, refine, seed=5)
    torch.manual_seed(8)
    gen_cpu = nets.build(define_G, 1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier').eval()
    gen_gpu = define_G(1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier').cuda()
    gen_gpu.load_state_dict(gen_cpu.state_dict())
    step = PretrainStep(net_gpu, render, gen_gpu, Config, views=views)
    p, _, cube = synthetic_batch(B, "cpu", seed=31)
    d = step.draw(B, "cpu", torch.Generator().manual_seed(32), np.random.default_rng(33))
    if frozen:
        # the same case with FROZEN BatchNorm statistics.  Round 3 built it as the candidate "well-conditioned" two-stage ResNet-50
        # case; it is not one: torch's own fp32 and fp64 CPU gradients still differ by 8.5 % (train mode 6.9 %, B = 8 x 3 11.4 %;
        # DESIGN.md section 2) -- the batch statistics are not what makes this net's gradient sensitive, its ~110 ReLU / max-pool
        # layers' discrete switches are.  It is therefore held to the float64-relative bar like the training-mode case.
        _freeze_statistics(net_cpu, net_gpu, orender, p, cube, d, views, gen_cpu)
    assert d["aug_view"].shape == (B * views, 3) and (views == 1) == bool((d["aug_view"] == 0).all())
    rec = _Recording(orender)
    loss_c = step_ref.pretrain_loss(net_cpu, rec, gen_cpu, p, cube, d, Config, views=views)
    loss_c.backward()
    assert len(rec.images) == int(refine)                               # the stage-2 bridge is Pretrain's only render() call
    loss_g, terms = step.loss(p.cuda(), cube.cuda(), draws_to(d, "cuda"))
    assert all(torch.isfinite(v) for v in terms.values())
    with torch.no_grad():
        pp, cc = (p.repeat_interleave(views, 0), cube.repeat_interleave(views, 0)) if views > 1 else (p, cube)
        s_c = step_ref.synth_pass(orender, gen_cpu, pp, cc, d, True)
    flips = 0
    if refine:
        flips = _bridge_flips(net_cpu, net_gpu, orender, render, s_c["img_t"], s_c["center"], s_c["cube"])
        fg = int((rec.images[0] < 0.99).sum())
        assert flips <= max(2, fg // 100), (flips, fg)                  # a few silhouette pixels at most
    pinned = None
    if flips:
        # the un-pinned step agrees as far as those pixels allow; the gradient comparison runs with the bridge image -- a
        # gradient-free input -- pinned to the oracle's
        assert abs(float(loss_g) - float(loss_c)) <= 1e-3 * abs(float(loss_c))
        pinned = PretrainStep(net_gpu, _PinnedBridge(render, rec.images), gen_gpu, Config, views=views, optimizer=step.opt)
        loss_g, _ = pinned.loss(p.cuda(), cube.cuda(), draws_to(d, "cuda"))
    loss_g.backward()
    if backbone.endswith("18") or not refine:
        _compare(loss_c, loss_g, net_cpu, net_gpu)
    else:
        # both fp32 paths against the float64 trunk (see the docstring).  Round 4 closed the question of WHY the HIP path lands
        # further from float64 than torch-CPU fp32 here (0.16-0.41 against 0.07-0.11 in round 3) with three controls
        # (profiles/r04_r50_controls_*.txt, r04_r50_noise.txt): (a) the torch twin on the same GPU lands at
        # 0.087 (torch's native kernels) and 0.149 (MIOpen); (b) teacher-forced, block by block, the HIP path's backward is as
        # close to float64 as torch-CPU's (medians 3e-7, test_two_stage_resnet50_teacher_forced_blocks); (c) torch-CPU fp32 ITSELF,
        # with 1e-7 relative noise on its block outputs (half of its own rounding), lands at 0.07 in 6 of 15 draws, at 0.27 in 6
        # and at 0.15 in 3: the whole-step error is decided by ONE discrete switch that carries a quarter of the gradient norm,
        # and every fp32 evaluation is a draw from those levels.  The bar below therefore stays what a draw can give (5 x the
        # CPU draw = 0.35 > 0.27 + margin); what guards the arithmetic is the per-block test.
        from dsf_amd import _lib as L
        net64 = _Net64(net_cpu)
        loss_64 = step_ref.pretrain_loss(net64, _PinnedBridge(orender, rec.images), gen_cpu, p, cube, d, Config, views=views)
        loss_64.backward()
        assert abs(float(loss_g) - float(loss_64)) <= 2e-3 * abs(float(loss_64))
        cos_c, rel_c = _grad_error(net64, net_cpu)
        cos_g, rel_g = _grad_error(net64, net_gpu)
        assert rel_c > 1e-2, "expected an ill-conditioned case (else use the strict bars)"
        print("two-stage ResNet-50 %s: cpu32 vs f64 (cos %.4f rel %.4f)  hip vs f64 (cos %.4f rel %.4f)" % ("frozen" if frozen else "train", cos_c, rel_c, cos_g, rel_g))
        assert rel_g <= 5.0 * rel_c and cos_g > 0.9, ((cos_c, rel_c), (cos_g, rel_g))      # observed r3: 3.7x / 0.933 (train), 4.1x / 0.921 (frozen)
        # the same step with every forward-type convolution unsplit (deterministic mode): the closer of the two HIP evaluations
        old = L.set_deterministic(True)
        try:
            for q in net_gpu.parameters():
                q.grad = None
            runner = pinned if flips else step
            if flips:
                pinned.render.__dict__["calls"] = 0                     # replay the pinned bridge images from the first
            loss_d, _ = runner.loss(p.cuda(), cube.cuda(), draws_to(d, "cuda"))
            loss_d.backward()
        finally:
            L.set_deterministic(old)
        cos_d, rel_d = _grad_error(net64, net_gpu)
        print("   unsplit convolutions (deterministic mode): hip vs f64 (cos %.4f rel %.4f)" % (cos_d, rel_d))
        assert abs(float(loss_d) - float(loss_64)) <= 2e-3 * abs(float(loss_64))
        # (a second SAMPLE of the same sensitivity, not a better path: on the generator-free variant of this case the unsplit
        #  evaluation lands at 0.16 and the split-K one at 0.28; here it is the other way round, 0.41 vs 0.37)
        assert rel_d <= 5.0 * rel_c and cos_d > 0.9, ((cos_c, rel_c), (cos_d, rel_d))


# ------------------------------------------------------------------------------------------------
# teacher-forced blocks: per-block arithmetic of the two-stage ResNet-50 without the chaotic amplification
# ------------------------------------------------------------------------------------------------
def _block_names(net):
    """the trunk cut into blocks: stem, every residual block, the transposed-convolution stages, MANO heads, fusion layer"""
    names = ["pre"]
    for suffix in ("", "_s2"):
        for l in range(1, 5):
            layer = getattr(net, "layer%d%s" % (l, suffix), None)
            if layer is not None:
                names += ["layer%d%s.%d" % (l, suffix, i) for i in range(len(layer))]
        for n in ("deconv_layer4", "deconv_layer3", "deconv_layer2", "mano_regress"):
            if hasattr(net, n + suffix):
                names.append(n + suffix)
    if hasattr(net, "fusion"):
        names.append("fusion")
    return names


def _sub(net, name):
    m = net
    for part in name.split("."):
        m = m[int(part)] if part.isdigit() else getattr(m, part)
    return m


def _record_blocks(net):
    """forward / full-backward hooks on every block of ``net`` (and on the two pixel heads): -> {name: {x, y, gy, gx}}"""
    cap = {}

    def hook(name, mod):
        def fwd(m, inp, out):
            cap.setdefault(name, {})["x"] = inp[0].detach().clone()
            cap[name]["y"] = out.detach().clone()

        def bwd(m, gin, gout):
            cap[name]["gy"] = gout[0].detach().clone()
            cap[name]["gx"] = None if gin[0] is None else gin[0].detach().clone()
        mod.register_forward_hook(fwd)
        mod.register_full_backward_hook(bwd)
    for n in _block_names(net):
        hook(n, _sub(net, n))
    for suffix in ("", "_s2"):
        for i, h in enumerate(getattr(net, "finals" + suffix, [])):
            hook("finals%s.%d" % (suffix, i), h)
    return cap


def _teacher_forced_rows(net_cpu, net_gpu, net64, cap):
    """Every block run ALONE on the float64 run's own input and differentiated against the float64 run's own upstream
    gradient (both cast to fp32), on the torch-CPU fp32 twin and on the HIP modules: -> [(name, (err_y, err_gx, err_gW) of the
    CPU twin, the same of the HIP path)], relative L2 errors against float64 (nan where there is nothing to compare); a fourth
    entry err_gx_trimmed = the input-gradient error without its largest differences (0.5 %, at least 25 pixels' worth of channels)."""
    from dsf_amd import nn_conv

    def rel(a, ref):
        ref = ref.double().cpu()
        return float((a.double().cpu() - ref).norm() / (ref.norm() + 1e-300))

    def rel_trimmed(a, ref, frac=0.005):
        """relative L2 error with the `frac` largest differences left out: a ReLU / max-pool switch changes a gradient through
        ONE activation, i.e. in the few elements of its receptive field (plus a 1 / M share through the BatchNorm sums), while a
        wrong formula is wrong everywhere -- the trimmed error separates the two"""
        channels = ref.shape[1] if ref.dim() == 4 else 1
        ref = ref.double().cpu().flatten()
        d = (a.double().cpu().flatten() - ref).abs()
        k = max(1, int(frac * d.numel()), 25 * channels)          # at least a 5 x 5 pixel neighbourhood of every channel (the 8 x 8 maps)
        kept = d.topk(d.numel() - k, largest=False).values if d.numel() > k else d[:0]
        return float(kept.norm() / (ref.norm() + 1e-300))

    def pgrads(mod):
        g = [q.grad.double().cpu().flatten() for q in mod.parameters() if q.grad is not None]
        return torch.cat(g) if g else torch.zeros(0, dtype=torch.float64)

    def run(mod, x64, gy64, dev, call=None):
        for q in mod.parameters():
            q.grad = None
        x = x64.float().to(dev).requires_grad_(True)
        y = (call or mod)(x)
        y.backward(gy64.float().to(dev))
        if dev == "cuda":
            torch.cuda.synchronize()
        return y.detach(), x.grad, pgrads(mod)
    jobs = [(n, _sub(net_cpu, n), _sub(net_gpu, n), None, None) for n in _block_names(net_cpu)]
    for suffix in ("", "_s2"):
        if not hasattr(net_cpu, "finals" + suffix):
            continue
        hc, hg = getattr(net_cpu, "finals" + suffix), getattr(net_gpu, "finals" + suffix)
        a, b = cap["finals%s.0" % suffix], cap["finals%s.1" % suffix]
        cap["finals" + suffix] = {"x": a["x"], "y": torch.cat([a["y"], b["y"]], 1), "gy": torch.cat([a["gy"], b["gy"]], 1),
                                  "gx": a["gx"] + b["gx"]}
        jobs.append(("finals" + suffix, hc, hg, (lambda x, h=hc: torch.cat([m(x) for m in h], 1)),
                     (lambda x, h=hg: nn_conv.fused_heads(x, h))))
    rows = []
    for n, mc, mg, call_c, call_g in jobs:
        c = cap[n]
        p64 = pgrads(_sub(net64.net, n))
        nan = float("nan")

        def errs(y, gx, gw):
            return (rel(y, c["y"]), rel(gx, c["gx"]) if (gx is not None and c["gx"] is not None) else nan,
                    rel(gw, p64) if p64.numel() else nan,
                    rel_trimmed(gx, c["gx"]) if (gx is not None and c["gx"] is not None) else nan)
        rows.append((n, errs(*run(mc, c["x"], c["gy"], "cpu", call_c)), errs(*run(mg, c["x"], c["gy"], "cuda", call_g))))
    return rows


def test_two_stage_resnet50_teacher_forced_blocks(render, orender):
    """The per-block form of the config-4 comparison, which a discrete switch cannot dominate.  The whole-step gradient of
    the two-stage ResNet-50 is decided by a handful of ReLU / max-pool switches (tools/r50_noise.py: torch's own CPU fp32
    evaluation with 1e-7 relative noise on its block outputs lands at 0.07 OR at 0.27 from float64), so
    the whole-step bar is loose by nature.  Here every block (stem, 32 Bottlenecks, 6 transposed-convolution stages, fusion
    layer, MANO and pixel heads) is run alone on the float64 run's input and upstream gradient: errors cannot travel, and
    each block's forward output, input gradient and parameter gradients are held against float64 next to torch-CPU fp32's.
    Measured (round 4): both paths 2e-7 .. 1e-6 on 35-39 of the 44 blocks; a block that holds an activation within rounding
    of zero shows one switch, 1e-4 .. 5e-3, on either path (9 such blocks for torch-CPU, 8 for HIP, mostly different ones)."""
    import numpy as _np
    from oracle import step_ref
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.train_step import PretrainStep, synthetic_batch, Config
    views, B = 3, 2
    net_cpu, net_gpu = _twin_pair(MANO_OCR_stage, "ResNet_stage_50", 21, True, seed=5)
    step = PretrainStep(net_gpu, render, None, Config, views=views)
    p, _, cube = synthetic_batch(B, "cpu", seed=31)
    d = step.draw(B, "cpu", torch.Generator().manual_seed(32), np.random.default_rng(33))
    net64 = _Net64(net_cpu)
    cap = _record_blocks(net64.net)
    step_ref.pretrain_loss(net64, orender, None, p, cube, d, Config, views=views).backward()
    rows = _teacher_forced_rows(net_cpu, net_gpu, net64, cap)
    assert len(rows) == 44                  # stem, 2 x (16 Bottlenecks + 3 transposed-convolution stages + MANO head + pixel heads), fusion
    fwd = _np.array([r[2][0] for r in rows])
    gx = _np.array([r[2][1] for r in rows if r[2][1] == r[2][1]])
    gw = _np.array([r[2][2] for r in rows if r[2][2] == r[2][2]])
    gx_c = _np.array([r[1][1] for r in rows if r[1][1] == r[1][1]])
    gw_c = _np.array([r[1][2] for r in rows if r[1][2] == r[1][2]])
    gxt = _np.array([r[2][3] for r in rows if r[2][3] == r[2][3]])
    gxt_c = _np.array([r[1][3] for r in rows if r[1][3] == r[1][3]])
    print("teacher-forced blocks, input-gradient error WITHOUT its 0.5 %% largest differences: HIP median %.2e max %.2e; torch-CPU median %.2e max %.2e; "
          "blocks whose full error is above 1e-5: %s" % (_np.median(gxt), gxt.max(), _np.median(gxt_c), gxt_c.max(),
          [(r[0], "%.1e -> %.1e" % (r[2][1], r[2][3])) for r in rows if r[2][1] == r[2][1] and r[2][1] > 1e-5]))
    # the switch discriminator (round 5): with the few largest differences left out every block is back near rounding level
    # (observed: HIP median 2.8e-7, the ten flagged blocks 5.7e-4 -> 8.2e-6 ... 1.6e-3 -> 2.0e-5, worst 2.0e-3 -> 1.2e-4 on an
    # 8 x 8 map; torch-CPU median 2.9e-7, max 3.5e-5) -- a formula that is wrong by 0.3 % stays at 3e-3 whatever is trimmed
    assert _np.median(gxt) < 2e-6 and gxt.max() < 5e-4, (float(_np.median(gxt)), float(gxt.max()))
    worst = sorted(rows, key=lambda r: -max(v for v in r[2][1:3] if v == v))[:3]
    print("teacher-forced blocks (HIP): forward max %.2e; input-gradient median %.2e max %.2e; parameter-gradient median %.2e max %.2e; "
          "blocks above 1e-5: HIP %d, torch-CPU %d of %d; worst %s" % (fwd.max(), _np.median(gx), gx.max(), _np.median(gw), gw.max(),
          int((gx > 1e-5).sum()), int((gx_c > 1e-5).sum()), len(gx), [(r[0], "%.1e" % max(v for v in r[2][1:3] if v == v)) for r in worst]))
    # forward: no switch can show in a block's own output (observed <= 1.2e-6, torch-CPU <= 1.1e-6)
    assert fwd.max() < 5e-6, [(r[0], r[2][0]) for r in rows if r[2][0] >= 5e-6]
    # backward: the typical block is at rounding level (observed medians 3.1e-7 / 4.6e-7; torch-CPU 3.3e-7 / 5.5e-7) ...
    assert _np.median(gx) < 2e-6 and _np.median(gw) < 3e-6, (_np.median(gx), _np.median(gw))
    # (deterministic mode runs every reduction unsplit: one fp32 accumulation chain over all of K -- up to 4392 terms -- instead of
    #  several shorter ones, hence medians of ~1e-6 there against 3-5e-7: still accumulation order, see test_gpu_conv.py)
    from dsf_amd import _lib as _L
    slack = 4.0 if _L.deterministic() else 2.0
    assert _np.median(gx) <= slack * _np.median(gx_c) and _np.median(gw) <= slack * _np.median(gw_c)
    # ... a minority of blocks holds a switch (observed 8 of 44, torch-CPU 9), and a switch is small (observed <= 4.5e-3):
    # a wrong backward formula in any op would put its blocks at >= 1e-1
    assert int((gx > 1e-5).sum()) <= len(gx) // 3, int((gx > 1e-5).sum())
    assert gx.max() < 3e-2 and gw.max() < 3e-2, (gx.max(), gw.max())


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
    fit = (pose + 0.003 * torch.randn(62, generator=torch.Generator().manual_seed(2))) if fitted else None
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
def test_config2_full_size_properties(render):
    """The headline configuration at its full size (bench.py's default workload: B = 32, ResNet-18 two-stage + MANO + rasteriser,
    the whole optimizer step): every loss term finite, the loss falls within 60 steps, gradients reach every trunk (both
    stages, the decoder, the heads and the MANO heads), and the residual blocks run with their twin outputs."""
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config
    from dsf_amd import nn_norm
    torch.manual_seed(0)
    net = MANO_OCR_stage("ResNet_stage_18", 21, True).cuda()
    step = RenderSupervisedStep(net, render, Config)
    p, c, cube = synthetic_batch(32, "cuda", seed=0)
    tgt = step.make_targets(p, c, cube, seed=1)
    assert tgt["img"].shape == (32, 1, 128, 128) and float((tgt["img"] < 0.99).float().mean()) > 0.02
    l0, terms = step(tgt)
    assert len(terms) == 13 and all(torch.isfinite(v) for v in terms.values()) and torch.isfinite(l0)
    grads = {n: q.grad for n, q in net.named_parameters()}
    assert all(g is not None and torch.isfinite(g).all() for g in grads.values())
    for n in ("pre.0.weight", "layer1.0.conv1.weight", "layer4.1.conv2.weight", "deconv_layer2.0.weight", "finals.0.weight", "mano_regress.2.weight",
              "fusion.0.weight", "layer1_s2.0.conv1.weight", "layer2_s2.0.downsample.0.weight", "layer4_s2.1.bn2.weight", "deconv_layer4_s2.0.weight",
              "finals_s2.1.bias", "mano_regress_s2.2.bias"):
        assert float(grads[n].abs().sum()) > 0, n
    hist = [float(step(tgt)[0]) for _ in range(60)]          # (AdamW at lr 1e-3 from random init: the first steps overshoot, then it falls)
    assert all(np.isfinite(hist)) and min(hist[-20:]) < float(l0)
    if nn_norm.TWIN[0]:                                       # the block outputs are handed out twice (nn_norm.take_twin)
        x = torch.randn(2, 64, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        assert "_dsf_twin" in net.layer1[0](x).__dict__


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
    # the pixel heads of the last stack feed no loss term here: their gradients are None; everything upstream of the
    # MANO head must have one
    grads = {n: q.grad for n, q in net.named_parameters()}
    assert all(torch.isfinite(g).all() for g in grads.values() if g is not None)
    for n in ("mano_regress.2.weight", "body.pre.0.conv.weight", "body.hgs.1.low1.conv2.conv.weight", "body.hgs.0.up1.conv1.conv.weight"):
        assert grads[n] is not None and float(grads[n].abs().sum()) > 0, n


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


# ------------------------------------------------------------------------------------------------
# ResNet-50 / Bottleneck on the HIP path against the arrays recorded from the imported reference
# ------------------------------------------------------------------------------------------------
def test_resnet50_bottleneck_vs_reference_golden():
    """model/resnet.py:58-98 Bottleneck + MANO_OCR_stage('ResNet_stage_50') (model/backbone.py:188-343): eval outputs,
    training-mode (batch-statistics) outputs and gradients vs tests/golden/reference_r50.npz."""
    import os
    from oracle import nets
    from dsf_amd.model.backbone import MANO_OCR_stage
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_r50.npz"))
    rng = np.random.default_rng(15)
    x = torch.tensor(rng.uniform(-1, 1, (2, 1, 128, 128)).astype(np.float32))
    gw_pix = torch.tensor(rng.normal(size=(2, 84, 64, 64)).astype(np.float32)).cuda()
    gw_par = torch.tensor(rng.normal(size=(2, 62)).astype(np.float32)).cuda()
    assert np.array_equal(x.numpy(), g["x"])
    torch.manual_seed(7)
    cpu = nets.build(MANO_OCR_stage, "ResNet_stage_50", 21, False)           # the reference's seed -> the reference's weights
    net = MANO_OCR_stage("ResNet_stage_50", 21, False).cuda()
    net.load_state_dict(cpu.state_dict())
    assert list(net.state_dict().keys()) == list(g["keys"])
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
    net.eval()
    with torch.no_grad():
        (pix, par), = net(x.cuda())
    assert rel(pix.cpu().numpy()[:, :, ::8, ::8], g["eval_pix_sub"]) < 2e-3
    assert rel(par.cpu().numpy(), g["eval_par"]) < 2e-3
    net.train()
    xg = x.cuda().requires_grad_(True)
    (pix, par), = net(xg)
    ((pix * gw_pix).sum() + (par * gw_par).sum()).backward()
    # batch statistics over B = 2 at 8x8 maps: 128 values per channel; fp32 summation order moves them by ~1e-6 relative, and
    # 53 BatchNorm layers at B = 2 amplify that.  Deterministic mode (DSF_DETERMINISTIC=1 in the environment) sums every
    # convolution's K in one unsplit chain -- another, equally valid order -- and lands 1.1 % from the recorded CPU values
    from dsf_amd import _lib as L
    bar = 2e-2 if L.deterministic() else 5e-3
    assert rel(pix.detach().cpu().numpy()[:, :, ::8, ::8], g["train_pix_sub"]) < bar
    assert rel(par.detach().cpu().numpy(), g["train_par"]) < bar
    named = dict(net.named_parameters())
    head_l2 = []
    for i, n in enumerate(g["probe_names"]):
        got = named[str(n)].grad.detach().cpu().numpy()
        norm = float(np.sqrt((got.astype(np.float64) ** 2).sum()))
        want = float(g["probe%d_norm" % i][0])
        assert abs(norm - want) <= 2e-2 * want, (n, norm, want)
        ref = g["probe%d_head" % i]
        # 1e-1 in both modes: over 16 default-mode runs on one box the worst head lands at 0.021-0.026 thirteen times and at
        # 0.045-0.048 three times (layer4.0.downsample.1.bias: one more activation switch of the B = 2 batch falls the other way
        # with the order of the float atomics), and once above 0.05 in the runs of round 3; the norms (2e-2) and the forward
        # outputs (5e-3) above never moved past 0.002 / 1.1e-4
        assert np.abs(got.reshape(-1)[:64] - ref).max() <= 1e-1 * max(np.abs(ref).max(), 1e-12), n
        head_l2.append(float(np.linalg.norm(got.reshape(-1)[:64] - ref) / max(np.linalg.norm(ref), 1e-30)))
    print("R50 golden: per-probe relative L2 error of the 64-entry gradient heads: max %.4f median %.4f" % (max(head_l2), float(np.median(head_l2))))
    # a statistic one switch cannot flip: the MEDIAN over the 12 probed tensors (one activation switch of the B = 2 batch moves a
    # few entries of a few tensors: it decides the maximum above, which is why that bar is loose, but not the median)
    assert float(np.median(head_l2)) <= 3e-2, head_l2                 # observed 0.0153 (max 0.021)
    assert rel(xg.grad.cpu().numpy()[:, :, ::4, ::4], g["grad_x_sub"]) < 1e-1
    rm = net.layer4[2].bn3.running_mean.cpu().numpy()[:32]
    assert np.abs(rm - g["running_mean_layer4_2_bn3_head"]).max() < 1e-4 * max(1.0, np.abs(rm).max())
# Render.forward: every output asserted directly (not only through a loss)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("views", [1, 3])
def test_render_forward_all_eight_outputs_vs_oracle(render, orender, views):
    """``Render.forward`` (mano_layer.py:983-1039) with every augmentation and the occluder mask, output by output against
    ``OracleRender.forward`` on the same draws: perturbed centre / cube and the crop matrix ``M`` EXACT; joint / vertex uvd and
    xyz within 1e-4 (north star); the image bit-exact when the rasteriser is handed the oracle's vertices, and -- end to end,
    where the fused MANO kernel's vertices differ from torch's by ulps -- equal except for silhouette pixels, depths within
    1e-4; the mask (:1326-1340) bit-equal on equal inputs."""
    from oracle import step_ref, hand_ref as H
    from dsf_amd.train_step import PretrainStep, synthetic_batch, draws_to, Config
    B = 4
    step = PretrainStep(None, render, None, Config, views=views, optimizer=object())
    p, _, cube = synthetic_batch(B, "cpu", seed=61)
    d = step.draw(B, "cpu", torch.Generator().manual_seed(62), np.random.default_rng(63))
    if views > 1:
        p, cube = p.repeat_interleave(views, 0), cube.repeat_interleave(views, 0)
    dg = draws_to(d, "cuda")
    md = (d["mask_joint_id"], d["mask_offset"], d["mask_radius"])
    with torch.no_grad():
        ref = orender.forward(p, d["center0"], cube, d["aug_view"], d["aug_shape"], d["aug_center"], d["aug_size"], None)
        got = render(p.cuda(), dg["center0"], cube.cuda(), augmentView=dg["aug_view"], augmentShape=dg["aug_shape"],
                     augmentCenter=dg["aug_center"], augmentSize=dg["aug_size"], mask=False)
    img_c, juvd_c, vuvd_c, jxyz_c, vxyz_c, center_c, cube_c, M_c = ref
    img_g, juvd_g, vuvd_g, jxyz_g, vxyz_g, center_g, cube_g, M_g = [t.cpu() for t in got]
    assert torch.equal(center_g, center_c) and torch.equal(cube_g, cube_c)            # perturbed centre / cube: exact
    assert torch.equal(M_g.float(), M_c.float())                                       # crop matrix: exact (integer bounds)
    for name, a, b in (("joint_uvd", juvd_g, juvd_c), ("verts_uvd", vuvd_g, vuvd_c), ("joint_xyz", jxyz_g, jxyz_c),
                       ("verts_xyz", vxyz_g, vxyz_c)):
        assert a.shape == b.shape and (a - b).abs().max() < 1e-4, (name, float((a - b).abs().max()))
    assert img_g.shape == img_c.shape == (B * views, 1, 128, 128)
    diff = (img_g - img_c).abs()
    fg = int((img_c < 0.99).sum())
    assert fg > 200 * B * views
    assert int((diff > 1e-4).sum()) <= max(2, fg // 200), (int((diff > 1e-4).sum()), fg)   # silhouette pixels only
    # the SAME vertices through both rasterisers: bit-exact image (world vertices rebuilt exactly as :1001-1016 does)
    with torch.no_grad():
        beta = p[:, 48:58] + d["aug_shape"]
        v, j = H.mano_vertices(orender.hm, p[:, :3], p[:, 3:48], beta, p[:, 58:62])
        c = j.mean(dim=1, keepdim=True)
        v = v - c + d["center0"].unsqueeze(1)
        Rm = H.rodrigues(d["aug_view"]).unsqueeze(1)
        c3 = d["center0"].unsqueeze(1)
        v = torch.matmul(Rm, (v - c3).unsqueeze(-1)).squeeze(-1) + c3
        img_same, _, _, minv_g = render._depth_crop(v.cuda().contiguous(), center_c.cuda(), cube_c.cuda())
        # ... and the SAME M^-1 bits: the product inverts M on the device, the oracle with LAPACK on the host; the two differ
        # by ~1e-7 relative, which decides the exact-.5 nearest-neighbour ties of the warp -- M^-1 is an
        # explicit input of the crop kernel for that reason
        c2, cb, _, _ = orender._crop_geometry(center_c, cube_c)
        X, Y, Z = v.unbind(-1)
        pv = torch.stack([(-X * np.float32(CAM[0] / 320.0)) / Z, (-Y * np.float32(CAM[1] / 240.0)) / Z, Z], -1)
        fv = pv[:, orender.faces].reshape(-1, 3, 3)
        img_o = step_ref._RasterCrop.apply(fv, v.shape[0], orender.faces.shape[0], minv_g.cpu().numpy(), orender.rowmap, c2[:, 2], cb[:, 2])
    assert torch.equal(img_same.cpu(), img_o)
    # mask_img on equal inputs: bit-equal, and it does occlude something
    with torch.no_grad():
        m_c = step_ref.mask_image(img_c, juvd_c, *md)
        m_g = render.mask_img(img_c.cuda(), juvd_c.cuda(), 0.15, 0.3, draws=(dg["mask_joint_id"], dg["mask_offset"], dg["mask_radius"])).cpu()
    assert torch.equal(m_g, m_c) and (m_c != img_c).any()
    # forward(mask=True) draws its occluders itself (host numpy + torch, as the reference): it only ever sets pixels to 1.0
    np.random.seed(5); torch.manual_seed(6)
    with torch.no_grad():
        full = render(p.cuda(), dg["center0"], cube.cuda(), augmentView=dg["aug_view"], augmentShape=dg["aug_shape"],
                      augmentCenter=dg["aug_center"], augmentSize=dg["aug_size"], mask=True)[0].cpu()
    assert full.shape == img_g.shape and ((full == img_g) | (full == 1.0)).all() and (full != img_g).any()
