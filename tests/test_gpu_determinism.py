"""DSF_DETERMINISTIC / dsf_set_deterministic(1): two runs of a whole step from the same state are
BITWISE equal -- the raster / point-face / collision backward kernels (fixed-point accumulators), the convolutions (no
split-K, ordered backward-weights partials) and everything downstream -- and the deterministic results agree with the
default (float-atomic) ones to fp32 accuracy."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CAM = (588.03, 587.07, 320.0, 240.0)


@pytest.fixture(scope="module")
def render():
    from dsf_amd.render_model.mano_layer import Render
    return Render("synthetic", "nyu", CAM, (640, 480)).cuda()


@pytest.fixture()
def det_mode():
    from dsf_amd import _lib as L
    old = L.set_deterministic(True)
    yield
    L.set_deterministic(old)


def _grads(net):
    return [p.grad.detach().clone() for p in net.parameters() if p.grad is not None]


def _run_twice(make_loss, net):
    out = []
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        loss = make_loss()
        loss.backward()
        torch.cuda.synchronize()
        out.append((loss.detach().clone(), _grads(net)))
    return out


def test_geometry_backward_kernels_are_bit_reproducible(render, det_mode):
    """crop rasteriser, fused point-to-mesh distance (whole hand + 15 parts), sphere collision: gradients w.r.t. the MANO
    parameters, twice, B = 16."""
    from dsf_amd.metric.meshLoss import ICPLoss, JointICPLoss
    from dsf_amd.train_step import synthetic_batch
    p, c, cube = synthetic_batch(16, "cuda", seed=4)
    mano = render.mano_layer
    with torch.no_grad():
        _, _, jx, mesh = render.render(p, c, cube)
        pcl = (mesh[:, torch.randint(0, 779, (2048,), device="cuda")] + 0.01 * torch.randn(16, 2048, 3, device="cuda")).contiguous()
        seg = mano.seg_pcl(jx, jx, mesh, pcl)
    gw = torch.randn(16, 1, 128, 128, device="cuda")
    res = []
    for _ in range(3):
        q = (p + 0.01).clone().requires_grad_(True)
        mano.clear_cache()
        img, juvd, jxyz, m = render.render(q, c, cube)
        loss = (img * gw).sum() + ICPLoss(m, pcl, mano.faces).sum() * 100 + JointICPLoss(m, pcl, mano.joint_faces, seg).sum() * 100 \
            + mano.calculate_coll(jxyz, m) * 10
        g, = torch.autograd.grad(loss, q)
        res.append((loss.detach().clone(), g.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[1][1], res[2][1])
    assert float(res[0][1].abs().sum()) > 0


def test_whole_step_is_bit_reproducible_and_matches_the_default_mode(render):
    from dsf_amd import _lib as L
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config
    torch.manual_seed(0)
    net = MANO_OCR_stage("ResNet_stage_18", 21, True).cuda()
    with torch.no_grad():
        for head in (net.mano_regress[2], net.mano_regress_s2[2]):
            head.bias[58] = 1.0
    step = RenderSupervisedStep(net, render, Config)
    p, c, cube = synthetic_batch(8, "cuda", seed=2)
    tgt = step.make_targets(p, c, cube)

    def loss():
        render.mano_layer.clear_cache()
        return step.loss(tgt)[0]
    default = _run_twice(loss, net)
    old = L.set_deterministic(True)
    try:
        det = _run_twice(loss, net)
    finally:
        L.set_deterministic(old)
    assert torch.equal(det[0][0], det[1][0])
    assert len(det[0][1]) == len(det[1][1]) > 100
    for a, b in zip(det[0][1], det[1][1]):
        assert torch.equal(a, b)                                        # bitwise
    # same mathematics as the default mode
    assert abs(float(det[0][0]) - float(default[0][0])) <= 1e-5 * abs(float(default[0][0]))
    num = sum(float(((a - b).double() ** 2).sum()) for a, b in zip(det[0][1], default[0][1]))
    den = sum(float((b.double() ** 2).sum()) for b in default[0][1])
    assert (num / den) ** 0.5 < 2e-2                                   # (fp32 rounding through ~40 BatchNorm layers at B = 8: the bar of the step-vs-oracle tests)
    # (the default mode is allowed to differ between its two runs: float atomics)


def test_flag_round_trip_and_wrw_scratch_contract():
    import ctypes
    from dsf_amd import _lib as L
    old = L.set_deterministic(True)
    try:
        assert L.deterministic()
        assert int(L.lib().dsf_conv_x6_wrw_workspace_bytes(*(ctypes.c_int(v) for v in (4, 16, 16, 64, 64, 3, 3)))) > 0
    finally:
        L.set_deterministic(old)
    assert L.deterministic() == old
    if not old:
        assert int(L.lib().dsf_conv_x6_wrw_workspace_bytes(*(ctypes.c_int(v) for v in (4, 16, 16, 64, 64, 3, 3)))) == 0


@pytest.mark.parametrize("kind", ["config2", "config3"])
def test_graphed_step_equals_the_eager_step_bitwise(render, det_mode, kind):
    """train_step.GraphedStep (forward + backward captured once in a HIP graph, replayed per batch; optimizer eager): in
    deterministic mode the trajectory -- losses, parameters, BatchNorm buffers incl. num_batches_tracked -- is BITWISE the
    eager one over several steps and a change of batch, and pure replays reproduce themselves (a captured hipMemsetAsync
    did not on ROCm 7.2, tools/graph_memset.py: every zero fill in csrc/ is a kernel)."""
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.model.hourglass import PoseNetMANO
    from dsf_amd.train_step import RenderSupervisedStep, MeshLossStep, GraphedStep, synthetic_batch, Config

    def make():
        torch.manual_seed(0)
        if kind == "config2":
            net = MANO_OCR_stage("ResNet_stage_18", 21, True).cuda()
            with torch.no_grad():
                for head in (net.mano_regress[2], net.mano_regress_s2[2]):
                    head.bias[58] = 1.0
            return RenderSupervisedStep(net, render, Config)
        net = PoseNetMANO(1, 21).cuda()
        return MeshLossStep(net, render, Config, n_points=512)
    eager, other = make(), make()
    tgts = []
    for seed in (2, 5):
        p, c, cube = synthetic_batch(4, "cuda", seed=seed)
        tgts.append(eager.make_targets(p, c, cube))
    graphed = GraphedStep(other, tgts[0], warmup=2)                       # (building the wrapper does not train: ADVICE r2)
    assert graphed.node_types.get(2, 0) == 0 and graphed.node_types.get(0, 0) > 300          # kernels only (+ a few copies)
    for (n, a), (_, b) in zip(eager.net.state_dict().items(), other.net.state_dict().items()):
        assert torch.equal(a, b), n                                      # warm-up and validation left parameters and statistics alone
    seq = [tgts[0], tgts[0], tgts[1], tgts[0]]
    for t in seq:
        le, _ = eager(t)
        lg, _ = graphed(t)
        assert torch.equal(le, lg)
    for (n, a), b in zip(eager.net.named_parameters(), other.net.parameters()):
        assert torch.equal(a, b), n
    for (n, a), (_, b) in zip(eager.net.state_dict().items(), other.net.state_dict().items()):
        assert torch.equal(a, b), n                                      # running statistics and the batch counters
    # pure replays from one state reproduce themselves
    graphed.graph.replay()
    torch.cuda.synchronize()
    first = [p.grad.clone() for p in other.net.parameters() if p.grad is not None]
    for _ in range(3):
        graphed.graph.replay()
    torch.cuda.synchronize()
    for a, p in zip(first, [p for p in other.net.parameters() if p.grad is not None]):
        assert torch.equal(a, p.grad)


@pytest.mark.parametrize("kind", ["config3", "config2", "config2_r50"])
def test_forked_streams_are_invisible(render, det_mode, kind):
    """Config 3 issues the two arms of every hourglass level and the four loss chains behind the MANO layer on forked streams
    (dsf_amd/streams.py); the two-stage ResNet step its downsample arms, the stage-2 bridge (MANO head -> MANO layer -> rasteriser
    -> offset map) beside the decoder, and the model branch of the loss beside the pixel branch.  In deterministic mode the
    trajectory of the eager step -- losses, every gradient, parameters, BatchNorm buffers -- is BITWISE that of the single-stream
    run, over several steps and a change of batch, and repeated passes from one state reproduce themselves (a missing stream
    dependency shows up as a difference here)."""
    from dsf_amd import streams
    from dsf_amd.model.hourglass import PoseNetMANO
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.train_step import MeshLossStep, RenderSupervisedStep, synthetic_batch, Config

    def make():
        torch.manual_seed(0)
        if kind == "config3":
            return MeshLossStep(PoseNetMANO(2, 21).cuda(), render, Config, n_points=512)
        net = MANO_OCR_stage("ResNet_stage_18" if kind == "config2" else "ResNet_stage_50", 21, True).cuda()
        with torch.no_grad():
            for head in (net.mano_regress[2], net.mano_regress_s2[2]):
                head.bias[58] = 1.0
        return RenderSupervisedStep(net, render, Config)
    one, forked = make(), make()
    tgts = []
    for seed in (3, 8):
        p, c, cube = synthetic_batch(4, "cuda", seed=seed)
        tgts.append(one.make_targets(p, c, cube))
    assert streams.ENABLED[0]
    for t in (tgts[0], tgts[1], tgts[0]):
        streams.ENABLED[0] = False
        try:
            l1, terms1 = one(t)
        finally:
            streams.ENABLED[0] = True
        l2, terms2 = forked(t)
        assert len(streams._STREAMS) >= 1                                   # the forks really happened
        assert torch.equal(l1, l2) and all(torch.equal(terms1[k], terms2[k]) for k in terms1)
        for (n, a), b in zip(one.net.named_parameters(), forked.net.parameters()):
            assert (a.grad is None) == (b.grad is None) and (a.grad is None or torch.equal(a.grad, b.grad)), n
    for (n, a), (_, b) in zip(one.net.state_dict().items(), forked.net.state_dict().items()):
        assert torch.equal(a, b), n
    # repeated forward + backward passes from one state: the same gradients every time
    first = None
    for _ in range(4):
        forked.forward_backward(tgts[1])
        torch.cuda.synchronize()
        g = [q.grad.clone() for q in forked.net.parameters() if q.grad is not None]
        if first is None:
            first = g
        assert all(torch.equal(a, b) for a, b in zip(first, g))


def test_graphed_step_refuses_a_graph_with_memset_nodes(render, monkeypatch):
    """A step holding a torch multi-block reduction (here: a one-shot global mean over a channels-last map) captures a
    hipMemsetAsync node; GraphedStep must refuse it rather than replay wrong numbers at some later step."""
    from dsf_amd.model.hourglass import PoseNetMANO
    from dsf_amd.train_step import MeshLossStep, GraphedStep, synthetic_batch, Config
    torch.manual_seed(0)
    net = PoseNetMANO(1, 21).cuda()
    net.mano_regress[0] = torch.nn.AdaptiveAvgPool2d(1)
    from dsf_amd import ops
    monkeypatch.setattr(ops, "pool_linear", lambda *a, **k: None)       # (round 6 fuses pooling + Linear: here torch's own pooling must run)
    step = MeshLossStep(net, render, Config, n_points=512)
    p, c, cube = synthetic_batch(4, "cuda", seed=2)
    tgt = step.make_targets(p, c, cube)
    # a data-parallel caller that catches the refusal and steps eagerly must find its reducer as it left it (ADVICE r3:
    # the hooks used to stay switched off, so the fallback trained without any all-reduce)
    import types
    step.grad_sync = types.SimpleNamespace(enabled=True, finish=lambda: None)
    with pytest.raises(RuntimeError, match="memset node"):
        GraphedStep(step, tgt, warmup=1)
    assert step.grad_sync.enabled is True
    step(tgt)                                                        # the step itself is untouched and still runs eagerly
    with pytest.raises(TypeError, match="forward_backward"):
        GraphedStep(types.SimpleNamespace(grad_sync=step.grad_sync, net=net), tgt)
    assert step.grad_sync.enabled is True


def test_backward_weights_side_stream_is_invisible(det_mode, monkeypatch):
    """nn_conv runs dW of a convolution on a second stream when nothing can read that gradient before the backward pass ends
    (leaf weight in kernel layout, no gradient yet, no foreign hooks).  In deterministic mode every variant below must
    give BITWISE the gradients of the single-stream run: one use (side stream taken), a network applied to two batches (the
    engine adds the two contributions on the main stream: the second joins the streams first), accumulation over two
    backward calls, a tensor hook on the weight, a non-leaf (merged) weight, a create_graph pass followed by a second
    backward, a standard-layout leaf weight (AccumulateGrad would clone its dW on the main stream) and a
    retained graph differentiated together with a new one (a forward-time use count under-counts there)."""
    from dsf_amd import nn_conv, nn_norm, _lib as L
    if nn_conv.MATH != "x6":
        pytest.skip("DSF_CONV_MATH=f32: the add-into-dW launches of the side stream are the split kernels'")
    monkeypatch.setattr(nn_conv, "WRW_MIN_WORK", [0.0])          # (these layers are below the default size threshold of the side stream)
    torch.manual_seed(2)
    net = torch.nn.Sequential(nn_conv.Conv2d(64, 128, 3, 1, 1, bias=False), nn_norm.FusedBatchNorm2d(128, fuse_relu=True),
                              nn_conv.Conv2d(128, 128, 3, 2, 1, bias=True), nn_norm.FusedBatchNorm2d(128, fuse_relu=True),
                              nn_conv.Conv2d(128, 64, 1, 1, 0, bias=False),
                              nn_conv.ConvTranspose2d(64, 32, 4, stride=2, padding=1, bias=False)).cuda()
    nn_conv.weights_changed()
    xa = torch.randn(8, 64, 32, 32, device="cuda")
    xb = torch.randn(8, 64, 32, 32, device="cuda")
    convs = [m for m in net if isinstance(m, nn_conv.Conv2d)]

    def grads():
        torch.cuda.synchronize()
        return [p.grad.clone() for p in net.parameters()]

    def one_use():
        net.zero_grad(set_to_none=True)
        net(xa).square().mean().backward()
        return grads()

    def two_uses():
        net.zero_grad(set_to_none=True)
        (net(xa).square().mean() + net(xb).abs().mean()).backward()
        return grads()

    def two_backwards():
        net.zero_grad(set_to_none=True)
        net(xa).square().mean().backward()
        net(xb).abs().mean().backward()
        return grads()

    def hooked():
        net.zero_grad(set_to_none=True)
        seen = []
        h = convs[1].weight.register_hook(lambda g: seen.append(float(g.abs().sum())) or g)
        try:
            net(xa).square().mean().backward()
        finally:
            h.remove()
        assert seen and seen[0] > 0
        return grads() + [torch.tensor(seen[0])]

    def merged_weight():
        net.zero_grad(set_to_none=True)
        w = torch.cat([convs[2].weight, convs[2].weight * 0.5], 0)                # non-leaf: autograd splits its gradient at once
        y = net[:4](xa)
        nn_conv.Conv2dFunction.apply(y, w, None, 1, (0, 0)).square().mean().backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in net[:5].parameters()]

    def double_backward():                                                       # (convolutions only: the fused BatchNorm is once-differentiable)
        net.zero_grad(set_to_none=True)
        xi = xa.clone().requires_grad_(True)
        out = convs[2](torch.nn.functional.leaky_relu(convs[0](xi), 0.2))
        g, = torch.autograd.grad(out.sum(), xi, create_graph=True)
        (g.square().mean() + out.square().mean()).backward()
        torch.cuda.synchronize()
        return [convs[0].weight.grad.clone(), convs[2].weight.grad.clone()]

    def standard_layout_weight():
        # a leaf weight in torch's standard memory order handed straight to the Function: dW (kernel order) would not obey
        # AccumulateGrad's layout contract, which then CLONES it on the main stream -- such a weight must stay on one stream
        net.zero_grad(set_to_none=True)
        w = convs[2].weight.detach().contiguous().clone().requires_grad_(True)
        assert not w.permute(2, 3, 1, 0).is_contiguous()
        y = net[:4](xa)
        nn_conv.Conv2dFunction.apply(y, w, None, 1, (0, 0)).square().mean().backward()
        torch.cuda.synchronize()
        return [w.grad.clone()] + [p.grad.clone() for p in net[:4].parameters()]

    def retained_plus_new_graph():
        # one backward pass over a RETAINED graph and a new one: every weight receives two contributions in that pass although
        # each forward saw it once -- decided per backward pass (nn_conv._wrw_dispatch), not from a forward-time count
        net.zero_grad(set_to_none=True)
        l1 = net(xa).square().mean()
        l1.backward(retain_graph=True)
        net.zero_grad(set_to_none=True)
        l2 = net(xb).abs().mean()
        (l1 + l2).backward()
        return grads()

    for case in (one_use, two_uses, two_backwards, hooked, merged_weight, double_backward, standard_layout_weight,
                 retained_plus_new_graph):
        nn_conv.WRW_STREAM[0] = False
        try:
            ref = case()
        finally:
            nn_conv.WRW_STREAM[0] = True
        for rep in range(3):
            got = case()
            for i, (a, b) in enumerate(zip(ref, got)):
                assert torch.equal(a, b), (case.__name__, rep, i)
    # a network applied to two batches in the default (float-atomic) mode: the second node of each weight ADDS into the first
    # node's dW on the side stream and hands autograd nothing -- same sums as the single-stream run up to atomic order, also
    # when the retained graph is differentiated a second time after zero_grad
    old = L.set_deterministic(False)
    try:
        nn_conv.WRW_STREAM[0] = False
        ref = two_uses()
        again = two_uses()                                                     # the float-atomic mode's own run-to-run noise
        dev = lambda u, v: max(float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)) for a, b in zip(u, v))
        bar = 5 * dev(ref, again) + 1e-5
        nn_conv.WRW_STREAM[0] = True
        for rep in range(3):
            assert dev(ref, two_uses()) <= bar
        net.zero_grad(set_to_none=True)
        loss = net(xa).square().mean() + net(xb).abs().mean()
        loss.backward(retain_graph=True)
        net.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize()
        assert dev(ref, [p_.grad for p_ in net.parameters()]) <= bar
        shared = []
        orig_side = nn_conv._on_side_stream
        nn_conv._on_side_stream = lambda fn, inputs: shared.append(1) or orig_side(fn, inputs)
        try:
            two_uses()
        finally:
            nn_conv._on_side_stream = orig_side
        assert len(shared) == 8                                                # 4 weights x 2 uses, all on the side stream
    finally:
        L.set_deterministic(old)
        nn_conv.WRW_STREAM[0] = True
    # the side stream is really taken in the plain case
    taken = []
    orig = nn_conv._on_side_stream
    nn_conv._on_side_stream = lambda fn, inputs: taken.append(1) or orig(fn, inputs)
    try:
        one_use()
        n_one = len(taken)
        two_uses()
        n_two = len(taken) - n_one
        standard_layout_weight()
        n_std = len(taken) - n_one - n_two
    finally:
        nn_conv._on_side_stream = orig
    # 3 convolutions + 1 transposed; a net used twice (deterministic mode): each weight's first contribution only, the second
    # joins the streams and runs on the main one; the standard-layout weight never (the two convolutions below it do)
    assert n_one == 4 and n_two == 4 and n_std == 2
    # the size threshold (DSF_WRW_MIN_GFLOP): a layer below it keeps its dW on the chain's stream -- same bits, nothing forked
    monkeypatch.setattr(nn_conv, "WRW_MIN_WORK", [1e30])
    ref = one_use()
    taken = []
    nn_conv._on_side_stream = lambda fn, inputs: taken.append(1) or orig(fn, inputs)
    try:
        got = one_use() + two_uses()
    finally:
        nn_conv._on_side_stream = orig
    assert not taken and all(torch.equal(a, b) for a, b in zip(ref, got))
    monkeypatch.setattr(nn_conv, "WRW_MIN_WORK", [2.0 * 8 * 128 * 32 * 32 * 64 * 9])       # the first layer's own work: it and nothing smaller forks
    taken = []
    nn_conv._on_side_stream = lambda fn, inputs: taken.append(1) or orig(fn, inputs)
    try:
        got = one_use()
    finally:
        nn_conv._on_side_stream = orig
    assert len(taken) == 1 and all(torch.equal(a, b) for a, b in zip(ref, got))


def test_backward_on_a_worker_thread_matches_the_main_thread(det_mode, monkeypatch):
    """The weight-gradient stream's bookkeeping (module-level lists, the per-pass record on the weight, the engine callback) is
    driven from whichever thread runs the backward pass: a backward pass started on a worker thread -- autograd then runs the
    device's nodes on its own engine thread -- gives bitwise the gradients of the main-thread run, repeatedly."""
    import threading
    from dsf_amd import nn_conv, nn_norm
    monkeypatch.setattr(nn_conv, "WRW_MIN_WORK", [0.0])
    torch.manual_seed(4)
    net = torch.nn.Sequential(nn_conv.Conv2d(32, 64, 3, 1, 1, bias=False), nn_norm.FusedBatchNorm2d(64, fuse_relu=True),
                              nn_conv.Conv2d(64, 64, 3, 2, 1, bias=True), nn_conv.ConvTranspose2d(64, 32, 4, stride=2, padding=1, bias=False)).cuda()
    nn_conv.weights_changed()
    x = torch.randn(6, 32, 24, 24, device="cuda")

    def run(out):
        net.zero_grad(set_to_none=True)
        net(x).square().mean().backward()
        torch.cuda.synchronize()
        out.append([p.grad.clone() for p in net.parameters()])
    ref = []
    run(ref)
    for _ in range(3):
        got = []
        th = threading.Thread(target=run, args=(got,))
        th.start(); th.join(timeout=120)
        assert got, "the worker thread did not finish"
        for a, b in zip(ref[0], got[0]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("kind,B", [("config2", 8), ("config2", 32), ("config3", 8), ("config5", 8)])
def test_every_kernel_of_a_step_is_stable_beside_convolution_workgroups(render, det_mode, kind, B):
    """Found in round 4, root-caused in round 5: a kernel that is correct alone can return different bits while conv_x6
    workgroups share its CUs.  The trigger is a platform erratum (tools/platform/pk_opsel_beside_mfma_lds.hip,
    profiles/r05_pk_opsel_erratum.txt): a packed-FP32 instruction whose low result selects (source 0 low, source 1 HIGH) --
    `op_sel:[0,1]` -- reads source 1 as 0.0 in lanes 48-63 while a wave issuing bf16 MFMAs runs on the same SIMD.  The compiler's
    vectorisers emit such instructions (the SLP-vectorised MANO backward lost one term of d/d(v_posed).x in 100 of 100 launches
    beside backward-weights launches), so the library is built with -fno-slp-vectorize -fno-vectorize and tests/test_isa_lint.py
    fails on any packed-FP32 instruction in the shipped code objects.  This test is the dynamic side of the same guarantee: every
    kernel of the step (one stream, deterministic mode) while an unrelated stream keeps conv_x6 workgroups on every CU must
    return the bits of the unloaded run.  Config 2 also at the benchmark's batch size (B = 32: torch's own elementwise / reduce
    kernels then launch hundreds of workgroups and really share SIMDs with the MFMA waves; profiles/r06_foreign_isa_scan.txt is the
    static side for those foreign kernels)."""
    from dsf_amd import nn_conv
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.model.hourglass import PoseNetMANO
    from dsf_amd.train_step import RenderSupervisedStep, MeshLossStep, FinetuneStageStep, synthetic_batch, draws_to, Config
    torch.manual_seed(0)
    p, c, cube = synthetic_batch(B, "cuda", seed=2)
    if kind in ("config2", "config5"):
        net = MANO_OCR_stage("ResNet_stage_18", 21, True).cuda()
        with torch.no_grad():
            for head in (net.mano_regress[2], net.mano_regress_s2[2]):
                head.bias[58] = 1.0
    if kind == "config2":
        step = RenderSupervisedStep(net, render, Config)
        tgt = step.make_targets(p, c, cube)
        loss_fn = lambda: step.loss(tgt)[0]
    elif kind == "config3":
        net = PoseNetMANO(1, 21).cuda()
        step = MeshLossStep(net, render, Config, n_points=512)
        tgt = step.make_targets(p, c, cube)
        loss_fn = lambda: step.loss(tgt)[0]
    else:                                                                  # the self-boosting step incl. the frozen generator
        from dsf_amd import ops
        from dsf_amd.render_model.transfer import define_G
        gen = define_G(1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier').cuda()
        step = FinetuneStageStep(net, render, gen, Config)
        pr, cr, cube_r = synthetic_batch(B, "cuda", seed=3)
        with torch.no_grad():
            img_r = render.render(pr, cr, cube_r)[0]
            _, M_r, _, _ = ops.crop_setup(cr, cube_r, render.cam, 128)
        d = draws_to(step.draw(B, "cpu", torch.Generator().manual_seed(7), np.random.default_rng(8)), "cuda")
        loss_fn = lambda: step.loss(p, cube, img_r, cr, cube_r, M_r, draws=d)[0]
    x = torch.randn(32, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    gy = torch.randn(32, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    conv = nn_conv.Conv2d(256, 256, 3, 1, 1, bias=False).cuda()
    side = torch.cuda.Stream()

    def run(load):
        if load:
            with torch.cuda.stream(side), torch.no_grad():                 # ~8 ms of conv_x6 workgroups on every CU (B = 32: ~35 ms)
                for _ in range(3 if B <= 8 else 14):
                    nn_conv._wrw(x, gy, 3, 3, 1, (1, 1))
                    conv(x)
        net.zero_grad(set_to_none=True)
        render.mano_layer.clear_cache()
        loss = loss_fn()
        loss.backward()
        torch.cuda.synchronize()
        return [loss.detach().clone()] + _grads(net)
    old = nn_conv.WRW_STREAM[0]
    nn_conv.WRW_STREAM[0] = False                                          # the step's own kernels all on one stream
    try:
        ref = run(False)
        again = run(False)
        assert all(torch.equal(a, b) for a, b in zip(ref, again))
        for r in range(6):
            got = run(True)
            bad = [i for i, (a, b) in enumerate(zip(ref, got)) if not torch.equal(a, b)]
            assert not bad, "run %d beside conv_x6 workgroups: %d of %d tensors differ from the unloaded run" % (r, len(bad), len(ref))
    finally:
        nn_conv.WRW_STREAM[0] = old


def test_mano_backward_is_stable_beside_convolution_workgroups(render):
    """the reproducer of the finding above, on the shipped library: 100 MANO backward calls while backward-weights launches of
    conv_x6 run on a second stream must all return the bits of a call that ran alone"""
    from dsf_amd import nn_conv, ops
    from dsf_amd.train_step import synthetic_batch
    p, _, _ = synthetic_batch(32, "cuda", seed=4)
    mano = render.mano_layer
    gV = torch.randn(32, 779, 3, device="cuda")
    gJ = torch.randn(32, 21, 3, device="cuda")
    x = torch.randn(32, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    gy = torch.randn(32, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    side = torch.cuda.Stream()

    def grad():
        q = p.clone().requires_grad_(True)
        v, j = ops.ManoPackedFunction.apply(mano._native(), q, 1000.0, 1.0)
        g, = torch.autograd.grad([v, j], q, [gV, gJ])
        torch.cuda.synchronize()
        return g
    ref = grad()
    bad = 0
    for _ in range(100):
        with torch.cuda.stream(side):
            for _ in range(2):
                nn_conv._wrw(x, gy, 3, 3, 1, (1, 1))
        bad += int(not torch.equal(grad(), ref))
    assert bad == 0, "%d of 100 MANO backward calls beside conv_x6 workgroups differ from the call that ran alone" % bad


def test_geometry_kernels_are_stable_beside_convolution_workgroups(render):
    """Round 5: the four kernels in which round 4's build still had packed-FP32 instructions (the loop vectoriser's, which
    -fno-slp-vectorize does not stop) -- the crop rasteriser forward (the kernel whose face / pixel indices must be bit-exact),
    its backward, Img2pcl and the Huber backward -- 200 times each while conv_x6 backward-weights launches (bf16 MFMAs, the
    aggressor of tools/platform/pk_opsel_beside_mfma_lds.hip) run on a second stream of the same process, exactly as the
    weight-gradient stream does in every step; bitwise against the call that ran alone.  The static side of the same
    guarantee is tests/test_isa_lint.py (no packed-FP32 instruction in any shipped code object)."""
    from dsf_amd import nn_conv, ops
    from dsf_amd.data.render_loader import loader
    from dsf_amd.metric.losses import SmoothL1Loss
    from dsf_amd.train_step import synthetic_batch
    B = 32
    p, c, cube = synthetic_batch(B, "cuda", seed=6)
    _, M, _, _ = ops.crop_setup(c, cube, render.cam, 128)
    L1, ld = SmoothL1Loss(), loader()
    gimg = torch.randn(B, 1, 128, 128, device="cuda")
    keys = torch.randint(0, 2 ** 31 - 1, (B, 128 * 128), device="cuda", dtype=torch.int32)
    tgt = torch.randn(B, 84, 64, 64, device="cuda") * 0.02
    pred0 = torch.randn(B, 84, 64, 64, device="cuda") * 0.02
    x = torch.randn(32, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    gy = torch.randn(32, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    side = torch.cuda.Stream()

    def crop():                                             # render_crop_fwd_kernel + render_crop_bwd_kernel (+ the MANO pair)
        render.mano_layer.clear_cache()
        q = p.clone().requires_grad_(True)
        verts, _ = render.mano_layer.get_mano_vertices_packed(q[:, :62], global_scale=1 / 125)
        verts = verts * cube.unsqueeze(1) / 2 + c.unsqueeze(1)
        center2d, M_auto, _, minv_c = ops.crop_setup(c, cube, render.cam, 128)
        minv = render._inverse(M_auto, minv_c)
        img, p2f = ops.RenderCropFunction.apply(verts, render.mano_layer.faces_i32, minv, render.resize_rowmap, center2d[:, 2].contiguous(),
                                                cube[:, 2].contiguous(), render.cam, 640, 128)
        g, = torch.autograd.grad(img, q, gimg)
        assert (p2f >= 0).float().mean() > 0.02
        return [img.detach(), p2f, g]

    def pcl():                                              # img2pcl_kernel, the random branch with explicit keys
        with torch.no_grad():
            img = render.render(p, c, cube)[0]
            return [ld.Img2pcl(img, 128, c, M, cube, 1024, rand_keys=keys)]

    def huber():                                            # huber_partial / final / bwd kernels
        pr = pred0.clone().requires_grad_(True)
        loss = L1(pr, tgt, weight=3.0)
        g, = torch.autograd.grad(loss, pr)
        return [loss.detach(), g]

    from dsf_amd import _lib as L
    old = L.set_deterministic(True)                         # the raster backward's float atomics would differ between runs by themselves
    try:
        for name, fn in (("crop rasteriser forward + backward", crop), ("Img2pcl", pcl), ("Huber forward + backward", huber)):
            ref = fn()
            torch.cuda.synchronize()
            again = fn()
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(ref, again)), name + ": not reproducible even alone"
            bad = 0
            for _ in range(200):
                with torch.cuda.stream(side):
                    for _ in range(2):
                        nn_conv._wrw(x, gy, 3, 3, 1, (1, 1))
                got = fn()
                torch.cuda.synchronize()
                bad += int(not all(torch.equal(a, b) for a, b in zip(ref, got)))
            assert bad == 0, "%s: %d of 200 calls beside conv_x6 backward-weights differ from the call that ran alone" % (name, bad)
    finally:
        L.set_deterministic(old)
