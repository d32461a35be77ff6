"""Round 6: the loss-side glue of the trainer steps as fused launches (csrc/step_ops.hip) against the reference's own torch expressions
(train_render.py:463-464, 728-732, 786-789; mano_layer.py:773-805, 874-884, 1078-1092; meshLoss.py:389-394), forward and backward;
the per-batch memo of the crop geometry; a BatchNorm layer applied twice in one backward pass."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def _crops(B, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    mk = lambda: torch.where(torch.rand(B, 1, 128, 128, device="cuda", generator=g) < 0.3,
                             torch.rand(B, 1, 128, 128, device="cuda", generator=g) * 1.6 - 0.8, torch.ones(B, 1, 128, 128, device="cuda"))
    return mk(), mk()


@pytest.mark.parametrize("B", [1, 7, 64])
def test_m2d_term_and_gate_sums(B):
    from dsf_amd import ops
    from dsf_amd.render_model.render_loss import m2d_loss
    real, synth = _crops(B, 3 + B)
    synth.requires_grad_(True)
    union = (real.lt(0.99) | synth.lt(0.99)).float()
    both = (real.lt(0.99) & synth.lt(0.99)).float()
    d = (real - synth).abs()
    ref = ((d * union).sum(-1).sum(-1) / (union.sum(-1).sum(-1) + 1e-8)).mean() * 0.1
    gref, = torch.autograd.grad(ref, synth)
    loss, sums, per = ops.m2d(real, synth)
    assert _rel(loss.double(), ref.detach().double()) < 2e-6
    assert torch.equal(sums[:, 1], union.sum((1, 2, 3))) and torch.equal(sums[:, 3], both.sum((1, 2, 3)))          # counts: exact
    assert _rel(sums[:, 0].double(), (d * union).double().sum((1, 2, 3))) < 2e-6 and _rel(sums[:, 2].double(), (d * both).double().sum((1, 2, 3))) < 2e-6
    assert not sums.requires_grad and not per.requires_grad
    g, = torch.autograd.grad(loss * 3.0, synth)
    assert _rel(g, gref * 3.0) < 2e-6
    assert _rel(m2d_loss(real, synth).detach(), ref.detach()) < 2e-6                 # the public function takes the fused path
    # an all-background pair: 0 / (0 + 1e-8) = 0, gradient 0
    ones = torch.ones(B, 1, 128, 128, device="cuda")
    l0, _, _ = ops.m2d(ones, ones.clone().requires_grad_(True))
    assert float(l0) == 0.0


def test_cube_points_are_the_reference_expressions_bit_for_bit():
    from dsf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    B = 9
    v = torch.randn(B, 779, 3, device="cuda", generator=g).requires_grad_(True)
    j = torch.randn(B, 21, 3, device="cuda", generator=g).requires_grad_(True)
    center = torch.randn(B, 3, device="cuda", generator=g) * 100 + 700
    cube = torch.rand(B, 3, device="cuda", generator=g) * 100 + 200
    ref = lambda p: (p * cube.unsqueeze(1) / 2 + center.unsqueeze(1))
    nrm = lambda w: (w - center.unsqueeze(1)) / cube.unsqueeze(1) * 2
    vw, jw, vn, jn = ops.CubePoints.apply(v, j, center, cube)
    assert torch.equal(vw, ref(v)) and torch.equal(jw, ref(j)) and torch.equal(vn, nrm(ref(v))) and torch.equal(jn, nrm(ref(j)))
    w = [torch.randn_like(t) for t in (vw, jw, vn, jn)]
    for use in ([0, 1, 2, 3], [2], [0, 3], [1]):                                       # any subset of the outputs may reach the loss
        loss = sum((o * w[i]).sum() for i, o in enumerate((vw, jw, vn, jn)) if i in use)
        gv, gj = torch.autograd.grad(loss, [v, j], retain_graph=True, allow_unused=True)
        rv, rj = torch.autograd.grad(sum((o * w[i]).sum() for i, o in enumerate((ref(v), ref(j), nrm(ref(v)), nrm(ref(j)))) if i in use), [v, j], allow_unused=True)
        for a, b_ in ((gv, rv), (gj, rj)):
            if b_ is None:
                assert a is None or float(a.abs().max()) == 0.0
            else:
                assert _rel(a, b_) < 1e-6


@pytest.mark.parametrize("rot_dim", [3, 4])
def test_view_rotation_in_one_launch(rot_dim):
    from dsf_amd.render_model import mano_layer as ml
    g = torch.Generator(device="cuda").manual_seed(rot_dim)
    B = 12
    v = torch.randn(B, 779, 3, device="cuda", generator=g) * 50 + 600
    j = torch.randn(B, 21, 3, device="cuda", generator=g) * 50 + 600
    c = torch.randn(B, 3, device="cuda", generator=g) * 10 + 600
    rot = torch.rand(B, rot_dim, device="cuda", generator=g) * 6.28
    Rt = ml._rotmat(rot.double()).transpose(1, 2)
    rv = torch.bmm(v.double() - c.double().unsqueeze(1), Rt) + c.double().unsqueeze(1)
    rj = torch.bmm(j.double() - c.double().unsqueeze(1), Rt) + c.double().unsqueeze(1)
    with torch.no_grad():
        ov, oj = ml.RotationPoints(v, j, c, rot)
    assert _rel(ov.double(), rv) < 2e-6 and _rel(oj.double(), rj) < 2e-6
    # with gradients wanted the differentiable torch formulation is taken
    v2 = v.clone().requires_grad_(True)
    assert ml.RotationPoints(v2, j, c, rot)[0].requires_grad


@pytest.mark.parametrize("n_parts", [15, 5])
def test_masked_part_means(n_parts):
    from dsf_amd.metric import meshLoss
    from dsf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(n_parts)
    B, P = 6, 2048
    dis = torch.rand(B, P, device="cuda", generator=g) * 1e-3
    dis[dis < 2e-4] = 0.0                                                           # exact zeros: selected but not counted
    seg = torch.randint(0, n_parts + 1, (B, P), device="cuda", generator=g)
    seg[0][seg[0] == 3] = 0                                                         # an empty part: 0, gradient 0
    dis.requires_grad_(True)
    labels = torch.arange(1, n_parts + 1, device="cuda").view(1, n_parts, 1)
    sel = seg.unsqueeze(1).eq(labels)
    per = torch.where(sel, dis.unsqueeze(1), torch.zeros_like(dis).unsqueeze(1))
    valid = per.gt(0).sum(-1)
    ref = per.sum(-1) / (valid + 1e-8)
    ref = torch.where(valid.eq(0), torch.zeros_like(ref), ref)
    w = torch.randn(B, n_parts, device="cuda", generator=g)
    gref, = torch.autograd.grad((ref * w).sum(), dis)
    out = meshLoss._masked_part_mean(dis, seg, n_parts)
    assert out.shape == (B, n_parts) and _rel(out.detach().double(), ref.detach().double()) < 2e-6 and float(out[0, 2]) == 0.0
    gg, = torch.autograd.grad((out * w).sum(), dis)
    assert _rel(gg, gref) < 2e-6


def test_mano_regularisers():
    from dsf_amd.train_step import _mano_regularisers
    g = torch.Generator(device="cuda").manual_seed(9)
    p = torch.randn(32, 62, device="cuda", generator=g).requires_grad_(True)
    ref_b = torch.mean(torch.pow(p[:, 48:58], 2)) * 1000.0
    ref_s = torch.mean(torch.abs(torch.clamp(p[:, 58], max=0.0))) * 0.1
    gref, = torch.autograd.grad(ref_b * 0.7 + ref_s * 1.3, p)
    b, s = _mano_regularisers(p, 1000.0, 0.1)
    assert _rel(b.detach(), ref_b.detach()) < 2e-6 and _rel(s.detach(), ref_s.detach()) < 2e-6
    gg, = torch.autograd.grad(b * 0.7 + s * 1.3, p)
    assert gg.shape == p.shape and _rel(gg, gref) < 2e-6 and float(gg[:, :48].abs().max()) == 0.0


def test_crop_geometry_is_computed_once_per_input_tensor(render):
    from dsf_amd import ops
    from dsf_amd.train_step import synthetic_batch
    ops.memo_clear()
    _, c, cube = synthetic_batch(5, "cuda", seed=4)
    a = ops.crop_setup(c, cube, render.cam, 128)
    b = ops.crop_setup(c, cube, render.cam, 128)
    assert all(x is y for x, y in zip(a, b))                                        # served from the memo: no launch
    assert ops.crop_setup(c, cube, render.cam, 64)[1] is not a[1]                    # other arguments: another entry
    mi = ops.inverse3x3(a[1])
    assert ops.inverse3x3(a[1]) is mi and torch.equal(mi, torch.linalg.inv_ex(a[1])[0])
    ref_c2, ref_M = a[0].clone(), a[1].clone()
    c.add_(1.0)                                                                     # an in-place write of the input: recomputed
    a2 = ops.crop_setup(c, cube, render.cam, 128)
    assert a2[1] is not a[1] and not torch.equal(a2[0], ref_c2)
    c.sub_(1.0)
    a3 = ops.crop_setup(c, cube, render.cam, 128)
    assert torch.equal(a3[1], ref_M)
    # never inside a stream capture (the kernels must be part of the graph)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=s):
            inside = ops.crop_setup(c, cube, render.cam, 128)
    torch.cuda.current_stream().wait_stream(s)
    assert inside[1] is not a3[1]


@pytest.fixture(scope="module")
def render():
    from dsf_amd.render_model.mano_layer import Render
    return Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()


@pytest.mark.parametrize("acc", [False, True])
def test_a_layer_applied_twice_adds_its_affine_gradients_in_the_kernel(acc, monkeypatch):
    """train_render.py:628-703 runs the network on two batches before ONE backward(): the second contribution to dgamma / dbeta is
    added by the BatchNorm backward itself (accumulate_affine), autograd gets None -- same values as autograd's own sum."""
    import contextlib
    from dsf_amd import nn_norm, _lib as L
    if acc and L.deterministic():
        pytest.skip("deterministic mode keeps the ordered-partials path")
    torch.manual_seed(3)
    bn = nn_norm.FusedBatchNorm2d(64).cuda().train()
    with torch.no_grad():
        bn.weight.normal_(); bn.bias.normal_()
    xs = [torch.randn(4, 64, 24, 24, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True) for _ in range(3)]
    out = {}
    for on in (True, False):
        monkeypatch.setattr(nn_norm, "AFFINE_ACCUMULATE", [on])
        bn.zero_grad(set_to_none=True)
        with (nn_norm.stat_pool(8 * nn_norm.acc_rows() * 2 * 64, "cuda") if acc else contextlib.nullcontext()):
            for x in xs:
                x.grad = None
            loss = sum((bn(x, None, True) * (i + 1.0)).square().mean() for i, x in enumerate(xs))
            loss.backward()
        out[on] = [bn.weight.grad.clone(), bn.bias.grad.clone()] + [x.grad.clone() for x in xs]
    for a, b_ in zip(out[True], out[False]):
        assert _rel(a, b_) < 1e-6


@pytest.mark.parametrize("case", ["mixed", "none", "only_row0"])
def test_m2p_term_in_one_launch(case):
    """selection + masked Huber + the reference's "sum of the selected indices == 0" rule (train_render.py:590-603 / 787-801)"""
    from dsf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(len(case))
    B = 16
    pix = (torch.randn(B, 21, 3, device="cuda", generator=g) * 0.02).requires_grad_(True)
    mano = torch.randn(B, 21, 3, device="cuda", generator=g) * 0.02
    pd = torch.rand(B, 15, device="cuda", generator=g) * 2e-3
    pd[3, 4] = float("nan")                                                         # lt() is False for NaN
    ok = torch.rand(B, device="cuda", generator=g) < 0.6
    if case == "none":
        ok[:] = False
    if case == "only_row0":
        ok[:] = False; ok[0] = True; pd[0] = 1.0                                    # only (sample 0, wrist) = flat row index 0 is selected
    jm = pd.lt(1e-3)
    jm = torch.cat((torch.ones(B, 1, device="cuda", dtype=torch.bool), jm, jm[:, [2, 5, 8, 11, 14]]), dim=-1)
    rows = (ok.unsqueeze(-1) & jm).reshape(-1)
    z = (pix.reshape(-1, 3) - mano.reshape(-1, 3)).float()
    az = z.abs()
    per_row = torch.where(az < 0.01, 0.5 * z * z, 0.01 * (az - 0.005)).mean(-1)
    m = rows.to(per_row.dtype)
    idx_sum = (torch.arange(m.numel(), device="cuda", dtype=per_row.dtype) * m).sum()
    val = (per_row * m).sum() / torch.clamp(m.sum(), min=1.0)
    ref = torch.where(idx_sum == 0, torch.zeros_like(val), val) * 100.0
    gref, = torch.autograd.grad(ref, pix)
    out = ops.M2P.apply(pix, mano, ok, pd, 100.0)
    assert _rel(out.detach().double(), ref.detach().double()) < 2e-6 or (float(ref) == 0.0 and float(out) == 0.0)
    gg, = torch.autograd.grad(out, pix)
    assert (float(gref.abs().max()) == 0.0 and float(gg.abs().max()) == 0.0) or _rel(gg, gref) < 2e-6
    if case != "mixed":
        assert float(out) == 0.0


def test_synthetic_render_with_the_fused_point_kernels_equals_the_elementwise_chain(render):
    """Render.forward (mano_layer.py:983-1039): the no-gradient path (placement + view rotation + cube normalisation as one launch
    each) against the differentiable path (the reference's chain of elementwise operators)."""
    from dsf_amd.train_step import synthetic_batch
    g = torch.Generator(device="cuda").manual_seed(11)
    B = 6
    p, _, cube = synthetic_batch(B, "cuda", seed=6)
    c0 = torch.cat((torch.zeros(B, 2, device="cuda"), torch.rand(B, 1, device="cuda", generator=g) * 700 + 500), -1)
    kw = dict(augmentView=torch.rand(B, 3, device="cuda", generator=g) * 6.28, augmentShape=torch.randn(B, 10, device="cuda", generator=g),
              augmentCenter=(torch.rand(B, 3, device="cuda", generator=g) - 0.5) * 40, augmentSize=1 + (torch.rand(B, 1, device="cuda", generator=g) - 0.5) * 0.4,
              mask=False)
    with torch.no_grad():
        a = render(p, c0, cube, **kw)
    b = render(p.clone().requires_grad_(True), c0, cube, **kw)                        # gradients wanted: the elementwise chain
    assert b[3].requires_grad and not a[3].requires_grad
    names = ("img", "joint_uvd", "verts_uvd", "joint_xyz", "verts_xyz", "center3d", "cube", "M")
    for n, x, y in zip(names, a, b):
        if n == "img":
            assert float(((x - y.detach()).abs() > 1e-4).float().mean()) < 2e-3    # (depths agree to rounding; a vertex moved by an ulp can flip a boundary pixel)
        else:
            assert _rel(x.double(), y.detach().double()) < 5e-6, n
    # without a view rotation as well
    kw["augmentView"] = None
    with torch.no_grad():
        a = render(p, c0, cube, **kw)
    b = render(p.clone().requires_grad_(True), c0, cube, **kw)
    assert _rel(a[4].double(), b[4].detach().double()) < 5e-6 and _rel(a[1].double(), b[1].detach().double()) < 5e-6


@pytest.mark.parametrize("B,J,S", [(5, 21, 64), (3, 21, 36), (2, 14, 8), (32, 21, 64)])
def test_soft_argmax_decode_reads_channels_last_maps_in_place(B, J, S, monkeypatch):
    """GFM.offset2joint_softmax (util/generateFeature.py:39-59) on a channels-last map (what the network produces), forward and backward,
    with the gradient coming back channels-last (no layout copy either way).  The pixel-chunk kernels (dsf_offset2joint_*_cl, round 6:
    contiguous reads of a pixel's record) evaluate the same expressions per element as the (sample, joint) kernels and fold their sums
    in another order: equal to the NCHW decode to fp32 summation distance, run-to-run bitwise; through the (sample, joint) kernels
    (DSF_DECODE_CL=0) the two layouts agree to the bit."""
    from dsf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(21 + S)
    H = 2 * S
    maps = torch.randn(B, 4 * J, S, S, device="cuda", generator=g) * 0.3
    depth = torch.where(torch.rand(B, 1, H, H, device="cuda", generator=g) < 0.4, torch.rand(B, 1, H, H, device="cuda", generator=g) * 1.6 - 0.8,
                        torch.ones(B, 1, H, H, device="cuda"))
    gj = torch.randn(B, J, 3, device="cuda", generator=g)
    a = maps.clone().requires_grad_(True)                                            # NCHW
    b = maps.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ja = ops.Offset2Joint.apply(a, depth, 0.8, 30.0)
    ga, = torch.autograd.grad((ja * gj).sum(), a)
    for cl in (True, False):
        monkeypatch.setattr(ops, "DECODE_CL", [cl])
        jb = ops.Offset2Joint.apply(b, depth, 0.8, 30.0)
        gb, = torch.autograd.grad((jb * gj).sum(), b)
        assert gb.is_contiguous(memory_format=torch.channels_last) and ga.is_contiguous()
        if cl:
            assert _rel(jb.detach().double(), ja.detach().double()) < 2e-6 and _rel(gb.double(), ga.double()) < 5e-6
            jb2 = ops.Offset2Joint.apply(b, depth, 0.8, 30.0)
            gb2, = torch.autograd.grad((jb2 * gj).sum(), b)
            assert torch.equal(jb, jb2) and torch.equal(gb, gb2)                     # ordered folds: deterministic
        else:
            assert torch.equal(ja, jb) and torch.equal(ga, gb)
    # a channel slice (not a dense layout) still works through a copy
    wide = torch.randn(B, 4 * J + 8, S, S, device="cuda", generator=g)
    assert torch.equal(ops.Offset2Joint.apply(wide[:, :4 * J], depth, 0.8, 30.0), ops.Offset2Joint.apply(wide[:, :4 * J].contiguous(), depth, 0.8, 30.0))


@pytest.mark.parametrize("B,C,H", [(32, 512, 8), (6, 2048, 4), (3, 256, 32)])
def test_pooled_linear_head_in_one_launch(B, C, H):
    """AdaptiveAvgPool2d(1) -> Flatten -> Linear(C, 62) (model/backbone.py:225-226) as ops.PoolLinear against the torch modules"""
    from dsf_amd import ops
    torch.manual_seed(C)
    seq = torch.nn.Sequential(torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(C, 62)).cuda()
    x = torch.randn(B, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    g = torch.randn(B, 62, device="cuda")
    ref = seq(xa)
    rg = torch.autograd.grad((ref * g).sum(), [xa, seq[2].weight, seq[2].bias])
    out = ops.pool_linear(xb, seq[2])
    assert out is not None and _rel(out.detach().double(), ref.detach().double()) < 2e-6
    og = torch.autograd.grad((out * g).sum(), [xb, seq[2].weight, seq[2].bias])
    assert og[0].is_contiguous(memory_format=torch.channels_last)
    for a, b_ in zip(og, rg):
        assert _rel(a.double(), b_.double()) < 5e-6


@pytest.mark.parametrize("B,H,W,cs", [(4, 16, 16, (64, 256, 84, 84)), (3, 7, 5, (8, 4)), (2, 9, 11, (4, 12, 8)), (1, 1, 1, (4, 4, 4, 4)), (5, 6, 6, (128, 128))])
def test_channel_concatenation_in_one_launch_equals_torch_cat(B, H, W, cs):
    """ops.cat_channels (dsf_cat_channels_nhwc): the output of torch.cat(dim=1) bit for bit, channels-last; the gradients are the slices"""
    from dsf_amd import ops
    g = torch.Generator().manual_seed(B + H + sum(cs))
    maps = [torch.randn(B, c, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(i != 1) for i, c in enumerate(cs)]
    ref = [m.detach().clone().requires_grad_(m.requires_grad) for m in maps]
    out = ops.cat_channels(maps)
    want = torch.cat(ref, dim=1)
    assert out.is_contiguous(memory_format=torch.channels_last) and torch.equal(out, want)
    gy = torch.randn(out.shape, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    out.backward(gy); want.backward(gy)
    for a, b in zip(maps, ref):
        assert (a.grad is None) == (b.grad is None)
        if a.grad is not None:
            assert torch.equal(a.grad, b.grad)
    # what the fused launch does not cover goes to torch.cat
    odd = [torch.randn(2, 3, 4, 4, device="cuda"), torch.randn(2, 5, 4, 4, device="cuda")]
    assert torch.equal(ops.cat_channels(odd), torch.cat(odd, dim=1))
