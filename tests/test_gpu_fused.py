"""Fused / layout-aware variants must give the results of the plain formulations they replace."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _composite_huber(x, y, size_average=True):
    z = (x - y).float()
    a = z.abs()
    per = torch.where(a < 0.01, 0.5 * z * z, 0.01 * (a - 0.005)).mean(dim=-1)
    return per.mean() if size_average else per.sum()


@pytest.mark.parametrize("shape,cl", [((32, 84, 16, 16), True), ((32, 84, 16, 16), False), ((7, 21, 3), False),
                                      ((5, 779, 3), False), ((3, 1, 1, 5), True), ((2, 5000, 3), False)])
@pytest.mark.parametrize("size_average", [True, False])
def test_fused_huber_matches_composite_and_oracle(shape, cl, size_average):
    from dsf_amd.metric.losses import SmoothL1Loss
    from oracle import image_ref
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.randn(shape, device="cuda", generator=g) * 0.02)
    y = (torch.randn(shape, device="cuda", generator=g) * 0.02)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    loss = SmoothL1Loss(size_average)(x, y)
    (loss * 3.0).backward()
    xr = x.detach().clone().requires_grad_(True)
    ref = _composite_huber(xr, y, size_average)
    (ref * 3.0).backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-6 * abs(float(ref.detach())) + 1e-12
    assert torch.allclose(x.grad, xr.grad, rtol=1e-5, atol=1e-12)
    assert x.grad.stride() == x.stride()
    if size_average:
        want = float(image_ref.huber(x.detach().cpu(), y.cpu()))
        assert abs(float(loss.detach()) - want) <= 2e-6 * abs(want) + 1e-12


def test_fused_huber_is_deterministic_and_falls_back_when_target_needs_grad():
    from dsf_amd.metric.losses import SmoothL1Loss
    x = torch.randn(32, 84, 64, 64, device="cuda") * 0.02
    y = torch.randn(32, 84, 64, 64, device="cuda") * 0.02
    a, b = SmoothL1Loss()(x, y), SmoothL1Loss()(x, y)
    assert torch.equal(a, b)
    yg = y.clone().requires_grad_(True)
    l = SmoothL1Loss()(x, yg)
    l.backward()
    assert yg.grad is not None and torch.isfinite(yg.grad).all()


def test_joint2offset_channels_last_output_and_strided_gradient():
    from dsf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(9)
    B, J, S = 6, 21, 64
    joints = (torch.rand(B, J, 3, device="cuda", generator=g) * 1.6 - 0.8).requires_grad_(True)
    img = torch.rand(B, 1, 128, 128, device="cuda", generator=g) * 2 - 1
    img = torch.where(img > 0.3, torch.ones_like(img), img)
    a = ops.Joint2Offset.apply(joints, img, 0.8, S, False)
    b = ops.Joint2Offset.apply(joints, img, 0.8, S, True)
    assert a.is_contiguous() and b.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(a, b)
    # gradient arriving as a channel slice of a wider channels-last tensor (the stage-2 fusion input)
    wide = torch.randn(B, 40 + 4 * J, S, S, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    gslice = wide[:, 40:]
    ga, = torch.autograd.grad(a, joints, gslice.contiguous(), retain_graph=True)
    gb, = torch.autograd.grad(b, joints, gslice)
    assert torch.equal(ga, gb)


def test_fused_heads_equal_separate_heads():
    from dsf_amd import nn_conv
    torch.manual_seed(3)
    heads = torch.nn.ModuleList([nn_conv.Conv2d(256, 63, 1), nn_conv.Conv2d(256, 21, 1)]).cuda()
    x = torch.randn(4, 256, 32, 32, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = nn_conv.fused_heads(x, heads)
    y2 = torch.cat([h(x) for h in heads], dim=1)
    assert y.shape == (4, 84, 32, 32)
    assert torch.allclose(y, y2, rtol=1e-5, atol=1e-5)
    gy = torch.randn_like(y)
    params = [heads[0].weight, heads[0].bias, heads[1].weight, heads[1].bias]
    g1 = torch.autograd.grad(y, [x] + params, gy)
    g2 = torch.autograd.grad(y2, [x] + params, gy)
    for a, b in zip(g1, g2):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))


def test_fused_adamw_matches_torch_adamw():
    from dsf_amd.optim import FusedAdamW
    from dsf_amd.nn_conv import kernel_layout_
    torch.manual_seed(11)
    shapes = [(64, 32, 3, 3), (257,), (5000, 3), (1,), (84, 256, 1, 1)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    kernel_layout_(pa[0], (2, 3, 1, 0))                                 # a parameter with the conv kernels' memory order
    pb = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in pa]
    oa = FusedAdamW(pa, lr=1e-3, weight_decay=0.01)
    ob = torch.optim.AdamW(pb, lr=1e-3, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.StepLR(oa, step_size=2, gamma=0.5)
    schedb = torch.optim.lr_scheduler.StepLR(ob, step_size=2, gamma=0.5)
    for it in range(5):
        for a, b in zip(pa, pb):
            g = torch.randn(a.shape, device="cuda") * (10.0 ** (it - 2))
            a.grad = g.clone() if it % 2 else g.clone().contiguous()
            b.grad = g.clone()
        oa.step(); ob.step(); sched.step(); schedb.step()
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (it, a.shape, float((a - b).abs().max()))
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"][0]["lr"] == sb["param_groups"][0]["lr"]
    for k in sb["state"]:
        ea, eb = sa["state"][k]["exp_avg"], sb["state"][k]["exp_avg"]            # torch's lerp may contract to an fma
        assert torch.allclose(ea, eb, rtol=1e-5, atol=1e-6 * float(eb.abs().max()))
        va, vb = sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"]
        assert torch.allclose(va, vb, rtol=1e-5, atol=1e-6 * float(vb.abs().max()))
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"]) == 5.0


def test_fused_adamw_keeps_a_step_count_per_parameter():
    """A parameter that sits out a step (grad None under zero_grad(set_to_none=True)) keeps its own step count and bias
    corrections, as torch.optim.AdamW does: A and B at step 1, only A at step 2, both at step 3."""
    from dsf_amd.optim import FusedAdamW
    torch.manual_seed(5)
    pa = [torch.nn.Parameter(torch.randn(300, device="cuda")), torch.nn.Parameter(torch.randn(7, 9, device="cuda"))]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa, ob = FusedAdamW(pa, lr=1e-2, weight_decay=0.01), torch.optim.AdamW(pb, lr=1e-2, weight_decay=0.01)
    for it, active in enumerate([(0, 1), (0,), (0, 1), (1,), (0, 1)]):
        for i in range(2):
            g = torch.randn(pa[i].shape, device="cuda") if i in active else None
            pa[i].grad = None if g is None else g.clone()
            pb[i].grad = None if g is None else g.clone()
        oa.step(); ob.step()
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (it, float((a - b).abs().max()))
    sa, sb = oa.state_dict(), ob.state_dict()
    assert [float(sa["state"][k]["step"]) for k in (0, 1)] == [float(sb["state"][k]["step"]) for k in (0, 1)] == [4.0, 4.0]


def test_mano_memo_sees_raw_pointer_writes():
    """The MANO result memo is keyed on the parameter rows' version AND the global write epoch: a leaf row block fitted
    directly with FusedAdamW (whose kernel writes behind torch's version counter) must not be served the previous step's
    vertices."""
    from dsf_amd.render_model.mano_layer import Render
    from dsf_amd.optim import FusedAdamW
    from dsf_amd.train_step import synthetic_batch
    render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
    p, _, _ = synthetic_batch(3, "cuda", seed=2)
    rows = torch.nn.Parameter(p.clone())
    opt = FusedAdamW([rows], lr=0.05, weight_decay=0.0)
    j0, v0 = render.get_mesh_xyz(rows)
    (v0.square().sum() + j0.square().sum()).backward()
    ver = rows._version
    opt.step()
    assert rows._version == ver                              # torch did not see the write
    j1, v1 = render.get_mesh_xyz(rows)
    fresh = render.mano_layer.get_mano_vertices(rows[:, :3], rows[:, 3:48], rows[:, 48:58], rows[:, 58:62], 1 / 125)[0]
    assert torch.equal(v1, fresh) and not torch.equal(v1, v0)
    (v1.square().sum()).backward()                           # and its graph is alive


def test_mano_packed_rows_equal_sliced_call():
    from dsf_amd.render_model.mano_layer import Render
    from dsf_amd.train_step import synthetic_batch
    render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
    mano = render.mano_layer
    p, _, _ = synthetic_batch(7, "cuda", seed=21)
    pa = p.clone().requires_grad_(True)
    pb = p.clone().requires_grad_(True)
    va, ja = mano.get_mano_vertices_packed(pa, 1 / 125)
    vb, jb = mano.get_mano_vertices(pb[:, :3], pb[:, 3:48], pb[:, 48:58], pb[:, 58:62], 1 / 125)
    assert torch.equal(va, vb) and torch.equal(ja, jb)
    gv, gj = torch.randn_like(va), torch.randn_like(ja)
    ((va * gv).sum() + (ja * gj).sum()).backward()
    ((vb * gv).sum() + (jb * gj).sum()).backward()
    assert torch.equal(pa.grad, pb.grad)
    # a (B, 64) row block (extra trailing columns are ignored by slicing to 62 first)
    wide = torch.cat([p, torch.zeros(7, 2, device="cuda")], 1)
    vc, _ = mano.get_mano_vertices_packed(wide[:, :62].contiguous(), 1 / 125)
    assert torch.equal(vc, va.detach())


def test_hip_maxpool_matches_torch():
    """dsf_maxpool_forward / _backward (NHWC, 1-byte argmax, gather backward) against torch.nn.functional.max_pool2d on the
    CPU: values bit-equal, gradients bit-equal (ties: first maximum in scan order), both pooling geometries of the nets."""
    from dsf_amd.nn_pool import MaxPool2d
    g = torch.Generator().manual_seed(3)
    for (B, C, H, W, k, s, p) in [(3, 64, 32, 32, 3, 2, 1), (2, 8, 17, 23, 3, 2, 1), (2, 128, 16, 16, 2, 2, 0), (1, 4, 5, 7, 3, 1, 1), (0, 8, 8, 8, 2, 2, 0)]:
        x = torch.randn(B, C, H, W, generator=g)
        x = (x * 4).round() / 4                                                  # many exact ties inside windows
        xc = x.clone().requires_grad_(True)
        yc = F.max_pool2d(xc, k, s, p)
        gy = torch.randn(yc.shape, generator=g)
        yc.backward(gy)
        xg = x.cuda().requires_grad_(True)
        yg = MaxPool2d(k, s, p)(xg)
        assert yg.shape == yc.shape and (B == 0 or yg.is_contiguous(memory_format=torch.channels_last))
        yg.backward(gy.cuda())
        assert torch.equal(yg.cpu(), yc) and torch.equal(xg.grad.cpu(), xc.grad), (B, C, H, W, k, s, p)
    x = torch.randn(1, 4, 6, 6)
    x[0, 1, 2, 2] = float("nan")
    assert torch.equal(torch.isnan(MaxPool2d(3, 2, 1)(x.cuda()).cpu()), torch.isnan(F.max_pool2d(x, 3, 2, 1)))
    with pytest.raises(RuntimeError):
        MaxPool2d(2, 2)(torch.randn(1, 4, 4, 4))
