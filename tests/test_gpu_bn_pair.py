"""Round 6: the fan-in sum of a block output's two gradients inside the fused BatchNorm backward (csrc/norm.hip:
dsf_bn_backward_pair / dsf_bn_backward_acc_pair; nn_norm twin outputs; reference model/resnet.py:39-55, 78-98).
Against torch CPU (fp32 tolerance), against the one-addend entry point on the pre-summed gradient (bitwise on the ordered path),
and through chains of residual blocks with the twin outputs on and off (bitwise in deterministic mode)."""
import contextlib
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


CL = torch.channels_last


@pytest.mark.parametrize("acc", [False, True])
@pytest.mark.parametrize("use", ["both", "twin_only", "first_only"])
@pytest.mark.parametrize("shape,res,relu", [((6, 64, 16, 16), True, True), ((6, 256, 8, 8), False, True), ((6, 512, 4, 4), True, False),
                                            ((3, 2048, 4, 4), True, True), ((6, 8, 5, 5), False, False), ((2, 64, 153, 153), True, True),
                                            # small maps (one launch per pass): M <= 256, <= 1024, just above
                                            ((1, 64, 1, 255), True, True), ((1, 128, 1, 1024), False, True), ((1, 8, 1, 1025), True, True)])
def test_twin_output_backward_matches_torch_cpu(shape, res, relu, use, acc):
    """y = bn(x) (+ r) (relu) handed out twice; loss = <y, ga> + <twin, gb>: every gradient against torch CPU with gy = ga + gb,
    also when only one of the two handles reaches the loss."""
    from dsf_amd import nn_norm, _lib as L
    from dsf_amd.nn_norm import FusedBatchNorm2d, take_twin
    if acc and L.deterministic():
        pytest.skip("deterministic mode keeps the ordered-partials path")
    C = shape[1]
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) * 2 + 0.5).requires_grad_(True)
    r = torch.randn(shape, generator=g).requires_grad_(True) if res else None
    ref = torch.nn.BatchNorm2d(C, momentum=0.1)
    with torch.no_grad():
        ref.weight.copy_(torch.randn(C, generator=g)); ref.bias.copy_(torch.randn(C, generator=g))
    fused = FusedBatchNorm2d(C, momentum=0.1).cuda()
    fused.load_state_dict(ref.state_dict())
    y = ref(x)
    if res:
        y = y + r
    if relu:
        y = F.relu(y)
    ga, gb = torch.randn(y.shape, generator=g), torch.randn(y.shape, generator=g)
    gy = {"both": ga + gb, "twin_only": gb, "first_only": ga}[use]
    inputs = [x, ref.weight, ref.bias] + ([r] if res else [])
    grads = torch.autograd.grad((y * gy).sum(), inputs)
    xg = x.detach().cuda().requires_grad_(True)
    rg = r.detach().cuda().requires_grad_(True) if res else None
    with (nn_norm.stat_pool(2 * nn_norm.acc_rows() * 2 * C, "cuda") if acc else contextlib.nullcontext()):
        yg = fused(xg, rg, relu, twin=True)
        assert "_dsf_twin" in yg.__dict__
        ya, yb = take_twin(yg)
        assert "_dsf_twin" not in yg.__dict__ and yb is not ya and yb.data_ptr() == ya.data_ptr() and yb.stride() == ya.stride()
        assert _rel(ya.cpu(), y.detach()) < 1e-5
        loss = 0
        if use in ("both", "first_only"):
            loss = loss + (ya * ga.cuda()).sum()
        if use in ("both", "twin_only"):
            loss = loss + (yb * gb.cuda().contiguous(memory_format=CL)).sum()
        gin = [xg, fused.weight, fused.bias] + ([rg] if res else [])
        gg = torch.autograd.grad(loss, gin)
    for a, b_ in zip(gg, grads):
        assert _rel(a.cpu(), b_) < 1e-4


@pytest.mark.parametrize("M,C,relu,res", [(4096, 64, 1, True), (6000, 256, 2, False), (2048 + 17, 128, 0, True), (3000, 2048, 1, True),
                                          (700, 64, 1, True), (5000, 32, 0, False)])
def test_pair_entry_point_equals_the_presummed_gradient_bitwise(M, C, relu, res, monkeypatch):
    """Ordered-partials path (fixed summation order): dsf_bn_backward_pair(gy, gy2) == dsf_bn_backward(gy + gy2) to the bit, with the
    sums pass writing the masked gradient (default) and without (DSF_BN_WRITE_G=0)."""
    from dsf_amd import _lib as L
    from dsf_amd.nn_norm import _workspace
    I, I64 = ctypes.c_int, ctypes.c_int64
    g = torch.Generator(device="cuda").manual_seed(M + C)
    dev = "cuda"
    x = torch.randn(M, C, device=dev, generator=g) * 1.5 + 0.3
    ga, gb = torch.randn(M, C, device=dev, generator=g), torch.randn(M, C, device=dev, generator=g)
    gamma, beta = torch.randn(C, device=dev, generator=g), torch.randn(C, device=dev, generator=g)
    mean = x.double().mean(0).float()
    invstd = (1.0 / torch.sqrt(x.double().var(0, unbiased=False) + 1e-5)).float()
    rsd = torch.randn(M, C, device=dev, generator=g)
    y = (x - mean) * (invstd * gamma) + beta + (rsd if res else 0)
    y = torch.relu(y) if relu else y
    ws = _workspace(torch.device("cuda", torch.cuda.current_device()), C)
    st = L.stream_ptr

    def run(fn_pair, write_g):
        monkeypatch.setenv("DSF_BN_WRITE_G", "1" if write_g else "0")
        gx = torch.empty_like(x); gr = torch.empty_like(x) if res else None
        gg, gbt = torch.empty(C, device=dev), torch.empty(C, device=dev)
        if fn_pair:
            L.check(L.lib().dsf_bn_backward_pair(_p(x), _p(ga), _p(gb), _p(y), _p(gamma), _p(beta), _p(mean), _p(invstd), I64(M), I(C), I(relu),
                                                 _p(gx), _p(gr), _p(gg), _p(gbt), I(0), _p(ws), st()), "dsf_bn_backward_pair")
        else:
            gs = ga + gb
            L.check(L.lib().dsf_bn_backward(_p(x), _p(gs), _p(y), _p(gamma), _p(beta), _p(mean), _p(invstd), I64(M), I(C), I(relu),
                                            _p(gx), _p(gr), _p(gg), _p(gbt), _p(ws), st()), "dsf_bn_backward")
        torch.cuda.synchronize()
        return [t for t in (gx, gr, gg, gbt) if t is not None]

    ref = run(False, False)
    for pair, wg in ((True, True), (True, False), (False, True)):
        for a, b_ in zip(run(pair, wg), ref):
            assert torch.equal(a, b_), (pair, wg)
    # and the values themselves against float64
    s = (ga + gb).double()
    if relu:
        s = s * (y > 0)
    xh = (x.double() - mean.double()) * invstd.double()
    dx = (gamma.double() * invstd.double()) * (s - s.mean(0) - xh * (s * xh).mean(0))
    assert _rel(ref[0].double(), dx) < 1e-5


def _blocks(kind):
    from dsf_amd.model import backbone, resnet
    from dsf_amd.nn_norm import ConvBN
    L_ = backbone._Layers()
    with L_:
        if kind == "basic":
            down = ConvBN(L_.Conv2d(32, 64, kernel_size=1, stride=2, bias=False), L_.bn(64))
            net = torch.nn.Sequential(resnet.BasicBlock(32, 32), resnet.BasicBlock(32, 64, 2, down), resnet.BasicBlock(64, 64))
            cin = 32
        else:
            d1 = ConvBN(L_.Conv2d(64, 128, kernel_size=1, stride=1, bias=False), L_.bn(128))
            d2 = ConvBN(L_.Conv2d(128, 256, kernel_size=1, stride=2, bias=False), L_.bn(256))
            net = torch.nn.Sequential(resnet.Bottleneck(64, 32, 1, d1), resnet.Bottleneck(128, 32), resnet.Bottleneck(128, 64, 2, d2))
            cin = 64
    return net.cuda().train(), cin


@pytest.mark.parametrize("kind", ["basic", "bottleneck"])
def test_residual_chain_with_twin_outputs_equals_autograds_own_sum(kind, monkeypatch):
    """Three residual blocks (one with a strided downsample arm on the forked stream): outputs, input gradient and every
    parameter gradient with the twin outputs on == off, bitwise in deterministic mode (the in-kernel sum is autograd's fp32 add);
    the intermediate twins are consumed, the chain's own output still carries one."""
    from dsf_amd import nn_norm, _lib as L
    was = L.set_deterministic(True)
    try:
        torch.manual_seed(5)
        net, cin = _blocks(kind)
        x0 = torch.randn(4, cin, 32, 32, device="cuda").contiguous(memory_format=CL)
        gy = torch.randn(4, net[-1].bn2.num_features if kind == "basic" else net[-1].bn3.num_features, 16, 16, device="cuda").contiguous(memory_format=CL)
        state = {k: v.clone() for k, v in net.state_dict().items()}
        out = {}
        for on in (True, False):
            net.load_state_dict(state)
            monkeypatch.setattr(nn_norm, "TWIN", [on])
            x = x0.clone().requires_grad_(True)
            # the chain's input read twice as well (as a block output would be): a plain tensor has no twin, autograd adds
            y = net(x)
            assert ("_dsf_twin" in y.__dict__) == on
            ya, yb = nn_norm.take_twin(y)
            loss = (ya * gy).sum() + (yb * yb).sum() * 0.25
            grads = torch.autograd.grad(loss, [x] + list(net.parameters()))
            torch.cuda.synchronize()
            out[on] = [y.detach().clone()] + [g_.clone() for g_ in grads]
        for a, b_ in zip(out[True], out[False]):
            assert torch.equal(a, b_)
    finally:
        L.set_deterministic(was)


def test_twin_outputs_in_the_float_atomic_mode_agree_with_autograds_sum(monkeypatch):
    """default (non-deterministic) mode with an open statistics pool: same chain, twin on vs off, to accumulation-order noise"""
    from dsf_amd import nn_norm, _lib as L
    if L.deterministic():
        pytest.skip("default mode only")
    torch.manual_seed(6)
    net, cin = _blocks("basic")
    x0 = torch.randn(8, cin, 64, 64, device="cuda").contiguous(memory_format=CL)
    gy = torch.randn(8, 64, 32, 32, device="cuda").contiguous(memory_format=CL)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    out = {}
    for on in (True, False):
        net.load_state_dict(state)
        monkeypatch.setattr(nn_norm, "TWIN", [on])
        x = x0.clone().requires_grad_(True)
        with nn_norm.stat_pool(nn_norm.stat_floats(net), "cuda"):
            y = net(x)
            grads = torch.autograd.grad((y * gy).sum(), [x] + list(net.parameters()))
        torch.cuda.synchronize()
        out[on] = [y.detach().clone()] + [g_.clone() for g_ in grads]
    for a, b_ in zip(out[True], out[False]):
        assert _rel(a, b_) < 2e-5
