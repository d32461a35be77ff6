"""Round 6: the backbone stem's BatchNorm -> ReLU -> MaxPool2d (reference model/backbone.py:200-204) as one training-mode layer
(csrc/norm.hip: dsf_bn_relu_pool_forward / _backward; nn_norm.FusedBatchNorm2d.forward_pooled; model/backbone.py _stem).
Against the separate layers' kernels (outputs, argmax routing and running statistics bit for bit; gradients up to the order of the
double-precision atomic sums), against torch on the CPU (fp32 tolerance), twice in one backward pass, with the statistics pool
running out between the passes, and through the stem helper in training and evaluation mode."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

CL = torch.channels_last

SHAPES = [((4, 64, 32, 32), 3, 2, 1),       # the stem's geometry
          ((20, 8, 7, 9), 3, 2, 1),         # odd map: windows cut by both borders
          ((18, 16, 10, 6), 2, 2, 0),       # the hourglass pooling's geometry
          ((2, 64, 33, 17), 3, 2, 0),       # rows / columns that no window contains
          ((42, 256, 5, 5), 2, 1, 1),       # overlapping 2 x 2 windows, stride 1
          ((30, 1024, 6, 6), 3, 2, 1),      # C / 4 = 256: one channel quad per thread
          ((65, 2048, 4, 4), 3, 3, 1),      # two column blocks
          ((1, 4, 3, 3), 3, 2, 1),          # one window per sample, one channel quad
          ((2, 8, 7, 9), 3, 2, 1)]          # (<= 1024 rows: the separate BatchNorm takes its one-launch kernel, whose sums differ in order)


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def _layer(C, seed):
    from dsf_amd.nn_norm import FusedBatchNorm2d
    g = torch.Generator().manual_seed(seed)
    bn = FusedBatchNorm2d(C, momentum=0.1, fuse_relu=True)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C, generator=g)); bn.bias.copy_(torch.randn(C, generator=g) * 0.5)
        bn.running_mean.copy_(torch.randn(C, generator=g)); bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    return bn.cuda().train()


def _pool_floats(C, passes=2):
    from dsf_amd import nn_norm
    return passes * nn_norm.acc_rows() * 2 * C


@pytest.mark.parametrize("shape,k,s,p", SHAPES)
def test_pooled_layer_equals_the_separate_layers(shape, k, s, p):
    from dsf_amd import nn_norm, nn_pool, _lib as L
    if L.deterministic():
        pytest.skip("deterministic mode keeps the separate layers")
    C = shape[1]
    g = torch.Generator().manual_seed(sum(shape) + k)
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3).cuda().contiguous(memory_format=CL)
    a, b = _layer(C, 7), _layer(C, 7)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    with nn_norm.stat_pool(_pool_floats(C), "cuda"):
        ya = a.forward_pooled(xa, k, s, p)
        assert ya is not None
        gy = torch.randn(ya.shape, generator=g).cuda().contiguous(memory_format=CL)
        ya.backward(gy)
    with nn_norm.stat_pool(_pool_floats(C), "cuda"):
        yb = nn_pool.MaxPool2d(k, s, p)(b(xb))
        yb.backward(gy)
    assert ya.shape == yb.shape and ya.is_contiguous(memory_format=CL)
    if shape[0] * shape[2] * shape[3] > 1024:                # both paths: the same reduction launch, the same apply arithmetic
        assert torch.equal(ya, yb)
        assert torch.equal(a.running_mean, b.running_mean) and torch.equal(a.running_var, b.running_var)
    else:
        assert _rel(ya, yb) <= 2e-6 and _rel(a.running_mean, b.running_mean) <= 2e-6 and _rel(a.running_var, b.running_var) <= 2e-6
    assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == 1
    # the gathered gradient is the pooling backward's bit for bit; the two channel sums meet through double atomics in both paths
    assert _rel(xa.grad, xb.grad) <= 2e-6
    assert _rel(a.weight.grad, b.weight.grad) <= 2e-6 and _rel(a.bias.grad, b.bias.grad) <= 2e-6
    # rows / columns outside every window, and elements the ReLU cut, get exactly the mean-subtraction terms: same zero pattern of g
    if shape[0] * shape[2] * shape[3] > 1024:
        assert torch.equal(xa.grad == 0, xb.grad == 0)


@pytest.mark.parametrize("shape,k,s,p", SHAPES[:5])
def test_pooled_layer_against_torch_cpu(shape, k, s, p):
    from dsf_amd import nn_norm, _lib as L
    if L.deterministic():
        pytest.skip("deterministic mode keeps the separate layers")
    C = shape[1]
    g = torch.Generator().manual_seed(11 * sum(shape))
    x = (torch.randn(shape, generator=g) * 2 - 0.2).requires_grad_(True)
    fused = _layer(C, 3)
    ref = nn.BatchNorm2d(C, momentum=0.1)
    ref.load_state_dict(fused.state_dict())
    y = nn.functional.max_pool2d(torch.relu(ref(x)), k, s, p)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xg = x.detach().cuda().requires_grad_(True)
    with nn_norm.stat_pool(_pool_floats(C), "cuda"):
        yg = fused.forward_pooled(xg, k, s, p)
        yg.backward(gy.cuda())
    assert (yg.cpu() - y).abs().max().item() <= 2e-5 * max(1.0, y.abs().max().item())
    assert _rel(xg.grad.cpu(), x.grad) <= 2e-4
    assert _rel(fused.weight.grad.cpu(), ref.weight.grad) <= 1e-4 and _rel(fused.bias.grad.cpu(), ref.bias.grad) <= 1e-4
    assert _rel(fused.running_mean.cpu(), ref.running_mean) <= 1e-6 and _rel(fused.running_var.cpu(), ref.running_var) <= 1e-5


def test_pooled_layer_applied_twice_in_one_backward_pass():
    """the layer on two batches before one backward() (the synthetic and the real batch of a step, train_render.py:628-703): the second
    contribution to dgamma / dbeta is added inside the kernel"""
    from dsf_amd import nn_norm, nn_pool, _lib as L
    if L.deterministic():
        pytest.skip("deterministic mode keeps the separate layers")
    C, shape = 64, (3, 64, 24, 24)
    g = torch.Generator().manual_seed(5)
    x1, x2 = [torch.randn(shape, generator=g).cuda().contiguous(memory_format=CL) for _ in range(2)]
    a, b = _layer(C, 9), _layer(C, 9)
    outs = []
    for bn, pooled in ((a, True), (b, False)):
        u, v = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
        with nn_norm.stat_pool(_pool_floats(C, 4), "cuda"):
            f = (lambda t: bn.forward_pooled(t, 3, 2, 1)) if pooled else (lambda t: nn_pool.MaxPool2d(3, 2, 1)(bn(t)))
            y1, y2 = f(u), f(v)
            ((y1 * y1).sum() + (y2 * 0.5).sum()).backward()
        outs.append((y1, y2, u.grad, v.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()))
    for fa, fb in zip(*outs):
        assert _rel(fa, fb) <= 2e-6


def test_statistics_pool_running_out_between_the_passes():
    """forward on the pooled path, no accumulation block left for the backward: the pooling's own backward + the ordered BatchNorm backward"""
    from dsf_amd import nn_norm, nn_pool, _lib as L
    if L.deterministic():
        pytest.skip("deterministic mode keeps the separate layers")
    C, shape = 32, (6, 32, 20, 12)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(shape, generator=g).cuda().contiguous(memory_format=CL)
    a, b = _layer(C, 1), _layer(C, 1)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    with nn_norm.stat_pool(_pool_floats(C, 1), "cuda"):      # ONE block
        ya = a.forward_pooled(xa, 3, 2, 1)
        assert ya is not None
        gy = torch.randn(ya.shape, generator=g).cuda()
        ya.backward(gy)
    yb = nn_pool.MaxPool2d(3, 2, 1)(b(xb))
    yb.backward(gy)
    assert torch.equal(ya, yb)
    assert _rel(xa.grad, xb.grad) <= 2e-6 and _rel(a.weight.grad, b.weight.grad) <= 2e-6 and _rel(a.bias.grad, b.bias.grad) <= 2e-6


def test_pooled_path_declines_what_it_does_not_cover():
    from dsf_amd import nn_norm
    bn = _layer(16, 0)
    x = torch.randn(2, 16, 8, 8, device="cuda").contiguous(memory_format=CL)
    assert bn.forward_pooled(x, 3, 2, 1) is None                       # no open statistics pool
    with nn_norm.stat_pool(_pool_floats(16), "cuda"):
        assert bn.forward_pooled(x, 5, 2, 2) is None                   # window sizes other than 2, 3
        assert bn.forward_pooled(x, 3, 1, 1) is None                   # more than 2 x 2 windows per pixel
        assert bn.forward_pooled(x, 2, 2, 2) is None                   # padding beyond half the window
        bn.eval()
        assert bn.forward_pooled(x, 3, 2, 1) is None                   # frozen statistics
        bn.train()
        sync = nn_norm.FusedSyncBatchNorm2d(16, fuse_relu=True).cuda().train() if hasattr(nn_norm, "FusedSyncBatchNorm2d") else None
        if sync is not None:
            assert sync.forward_pooled(x, 3, 2, 1) is None             # cross-replica statistics: the exchange sits between the passes
        assert bn.forward_pooled(x, 3, 2, 1) is not None


@pytest.mark.parametrize("training", [True, False])
def test_stem_helper_equals_the_module_sequence(training):
    from dsf_amd import nn_norm, nn_conv, _lib as L
    from dsf_amd.model import backbone
    if L.deterministic():
        pytest.skip("deterministic mode keeps the separate layers")
    Ls = backbone._Layers()
    torch.manual_seed(4)
    pre = nn.Sequential(Ls.Conv2d(1, 64, kernel_size=5, stride=1, padding=2, bias=False), *Ls.bn_relu(64, momentum=0.1),
                        Ls.MaxPool2d(kernel_size=3, stride=2, padding=1)).cuda().train(training)
    ref = copy.deepcopy(pre)
    start = copy.deepcopy(pre[1].state_dict())
    img = torch.randn(4, 1, 32, 32, device="cuda")
    floats = nn_norm.stat_floats(pre)
    with nn_norm.stat_pool(floats, "cuda"):
        ya = backbone._stem(pre, img)
        ga = torch.autograd.grad((ya * ya).sum(), [q for q in pre.parameters()])
    saved = nn_norm.POOL_FUSED[0]
    nn_norm.POOL_FUSED[0] = False
    try:
        with nn_norm.stat_pool(floats, "cuda"):
            yb = backbone._stem(ref, img)
            gb = torch.autograd.grad((yb * yb).sum(), [q for q in ref.parameters()])
    finally:
        nn_norm.POOL_FUSED[0] = saved
    assert torch.equal(ya, yb)
    for u, v in zip(ga, gb):
        assert _rel(u, v) <= 5e-6
    assert torch.equal(pre[1].running_mean, ref[1].running_mean) and torch.equal(pre[1].running_var, ref[1].running_var)
    # ... and the plain module sequence (the BatchNorm's own statistics pass instead of the convolution's epilogue sums)
    seq = copy.deepcopy(ref)
    seq[1].load_state_dict(start)
    with nn_norm.stat_pool(floats, "cuda"):
        yc = seq(img)
        gc = torch.autograd.grad((yc * yc).sum(), [q for q in seq.parameters()])
    assert _rel(ya, yc) <= 1e-5
    for u, v in zip(ga, gc):
        assert _rel(u, v) <= 2e-4
    assert _rel(pre[1].running_mean, seq[1].running_mean) <= 1e-5 and _rel(pre[1].running_var, seq[1].running_var) <= 1e-5


@pytest.mark.parametrize("B,H,W,Co,K,stride,pad", [(4, 32, 32, 64, 5, 1, 2), (3, 37, 29, 64, 7, 2, 3), (2, 16, 24, 32, 5, 2, 2), (70, 64, 64, 64, 5, 1, 2)])
def test_stem_convolution_epilogue_statistics(B, H, W, Co, K, stride, pad):
    """dsf_conv_c1_forward_bn_acc: the output of dsf_conv_c1_forward bit for bit, and the per-channel sum / sum of squares of that
    output in the accumulation rows (double sums of float partials: 1e-6 of the float64 sums)"""
    import ctypes
    from dsf_amd import _lib as L
    if L.deterministic():
        pytest.skip("deterministic mode keeps the BatchNorm's own ordered statistics pass")
    g = torch.Generator().manual_seed(B + H + K)
    x = torch.randn(B, H, W, 1, generator=g).cuda()
    w = torch.randn(K, K, 1, Co, generator=g).cuda()
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    y0, y1 = torch.empty(B, Ho, Wo, Co, device="cuda"), torch.empty(B, Ho, Wo, Co, device="cuda")
    rows = int(L.lib().dsf_bn_acc_rows())
    acc = torch.zeros(rows, 2, Co, device="cuda", dtype=torch.float64)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    I = ctypes.c_int
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.lib().dsf_conv_c1_forward(P(x), P(w), ctypes.c_void_p(0), P(y0), I(B), I(H), I(W), I(Ho), I(Wo), I(Co), I(K), I(stride), I(pad), st) == 0
    assert L.lib().dsf_conv_c1_forward_bn_acc(P(x), P(w), ctypes.c_void_p(0), P(y1), I(B), I(H), I(W), I(Ho), I(Wo), I(Co), I(K), I(stride), I(pad), P(acc), I(rows), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    yd = y0.double().reshape(-1, Co)
    s = acc.sum(0)
    assert _rel(s[0], yd.sum(0)) <= 1e-6 * max(1.0, (yd.abs().sum(0).max() / yd.sum(0).abs().max()).item())
    assert _rel(s[1], (yd * yd).sum(0)) <= 1e-6


@pytest.mark.parametrize("B,H,W,Co,K,stride,pad,pool,relu", [(4, 32, 32, 64, 5, 1, 2, (3, 2, 1), 1), (3, 37, 29, 64, 7, 2, 3, (2, 2, 0), 1),
                                                             (5, 20, 24, 32, 5, 1, 2, None, 1), (5, 40, 44, 64, 7, 2, 3, None, 0),
                                                             (2, 33, 47, 64, 5, 1, 2, (3, 2, 1), 1), (9, 16, 16, 16, 5, 1, 2, (2, 2, 0), 1)])
def test_apply_and_weight_gradient_in_one_launch_bit_for_bit(B, H, W, Co, K, stride, pad, pool, relu):
    """dsf_conv_c1_wrw_bn on the accumulation rows of a sums pass == the BatchNorm backward's apply pass (dsf_bn_relu_pool_backward /
    dsf_bn_backward_acc) followed by dsf_conv_c1_wrw on the gradient it wrote: dW, dgamma, dbeta bit for bit"""
    import ctypes
    from dsf_amd import _lib as L
    if L.deterministic():
        pytest.skip("accumulation rows are not used in deterministic mode")
    lib = L.lib()
    g = torch.Generator().manual_seed(B * H + K)
    dev = "cuda"
    x = torch.randn(B, H, W, 1, generator=g).to(dev)
    w = torch.randn(K, K, 1, Co, generator=g).to(dev)
    gamma, beta = (torch.rand(Co, generator=g) + 0.5).to(dev), (torch.randn(Co, generator=g) * 0.3).to(dev)
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    P = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
    I, F, I64 = ctypes.c_int, ctypes.c_float, ctypes.c_int64
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = int(lib.dsf_bn_acc_rows())
    zeros = lambda: torch.zeros(rows * 2 * Co, device=dev, dtype=torch.float64)
    y = torch.empty(B, Ho, Wo, Co, device=dev)
    acc_f = zeros()
    bias = (torch.randn(Co, generator=g) * 0.2).to(dev) if K == 7 else None
    assert lib.dsf_conv_c1_forward_bn_acc(P(x), P(w), P(bias), P(y), I(B), I(H), I(W), I(Ho), I(Wo), I(Co), I(K), I(stride), I(pad), P(acc_f), I(rows), st) == 0
    mean, invstd = torch.empty(Co, device=dev), torch.empty(Co, device=dev)
    if pool:
        k, s, p = pool
        Po, Qo = (Ho + 2 * p - k) // s + 1, (Wo + 2 * p - k) // s + 1
        out, arg = torch.empty(B, Po, Qo, Co, device=dev), torch.empty(B, Po, Qo, Co, device=dev, dtype=torch.uint8)
        assert lib.dsf_bn_relu_pool_forward(P(y), P(gamma), P(beta), I(B), I(Ho), I(Wo), I(Co), I(k), I(s), I(p), F(1e-5), F(0.1), P(None), P(None),
                                            P(out), P(arg), P(mean), P(invstd), P(acc_f), I(1), st) == 0
    else:
        k = s = p = 0
        Po, Qo, arg = Ho, Wo, None
        out = torch.empty_like(y)
        assert lib.dsf_bn_forward_acc(P(y), P(None), P(gamma), P(beta), I64(B * Ho * Wo), I(Co), F(1e-5), F(0.1), I(relu), P(None), P(None), P(out),
                                      P(mean), P(invstd), P(acc_f), I(1), st) == 0
    gy = torch.randn(B, Po, Qo, Co, generator=g).to(dev)
    # two launches: apply pass writes dx, the stem's dW kernel reads it
    acc1, dx = zeros(), torch.empty_like(y)
    gg1, gb1 = torch.empty(Co, device=dev), torch.empty(Co, device=dev)
    if pool:
        assert lib.dsf_bn_relu_pool_backward(P(y), P(gy), P(arg), P(gamma), P(beta), P(mean), P(invstd), I(B), I(Ho), I(Wo), I(Co), I(k), I(s), I(p),
                                             P(dx), P(gg1), P(gb1), I(0), P(acc1), st) == 0
    else:
        assert lib.dsf_bn_backward_acc(P(y), P(gy), P(None), P(gamma), P(beta), P(mean), P(invstd), I64(B * Ho * Wo), I(Co), I(2 if relu else 0),
                                       P(dx), P(None), P(gg1), P(gb1), P(acc1), st) == 0
    ws = torch.empty(lib.dsf_conv_c1_workspace_bytes(I(K), I(K)) // 4, device=dev)
    dw1 = torch.empty(K * K, Co, device=dev)
    assert lib.dsf_conv_c1_wrw(P(x), P(dx), P(dw1), P(ws), I(B), I(H), I(W), I(Ho), I(Wo), I(Co), I(K), I(stride), I(pad), I(0), st) == 0
    # one launch on the same rows; and the sums-only pass leaves the same kind of rows
    dw2, gg2, gb2 = torch.empty(K * K + 1, Co, device=dev), torch.empty(Co, device=dev), torch.empty(Co, device=dev)
    ws2 = torch.empty_like(ws)
    assert lib.dsf_conv_c1_wrw_bn(P(x), P(y), P(gy), P(arg), P(gamma), P(beta), P(mean), P(invstd), P(acc1), I(rows), I(relu), I(k), I(s), I(p), P(dw2),
                                  P(gg2), P(gb2), I(0), P(ws2), I(B), I(H), I(W), I(Ho), I(Wo), I(Co), I(K), I(stride), I(pad), st) == 0
    acc3 = zeros()
    if pool:
        assert lib.dsf_bn_relu_pool_backward(P(y), P(gy), P(arg), P(gamma), P(beta), P(mean), P(invstd), I(B), I(Ho), I(Wo), I(Co), I(k), I(s), I(p),
                                             P(None), P(None), P(None), I(0), P(acc3), st) == 0
    else:
        assert lib.dsf_bn_backward_acc(P(y), P(gy), P(None), P(gamma), P(beta), P(mean), P(invstd), I64(B * Ho * Wo), I(Co), I(2 if relu else 0),
                                       P(None), P(None), P(None), P(None), P(acc3), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(dw1, dw2[:K * K]) and torch.equal(gg1, gg2) and torch.equal(gb1, gb2)
    # row K*K: the convolution's bias gradient = the per-channel sum of dx (zero in exact arithmetic behind a BatchNorm: rounding noise)
    dxd = dx.double().reshape(-1, Co)
    assert (dw2[K * K].double() - dxd.sum(0)).abs().max().item() <= 1e-6 * dxd.abs().sum(0).max().item()
    assert _rel(acc3.view(rows, -1).sum(0), acc1.view(rows, -1).sum(0)) <= 1e-12
    # against float64 on the CPU: dW of conv(x, w) -> BatchNorm(train) -> (ReLU) -> (MaxPool2d)
    xc = x.cpu().double().permute(0, 3, 1, 2)
    wc = w.cpu().double().permute(3, 2, 0, 1).clone().requires_grad_(True)
    gc, bc = gamma.cpu().double().requires_grad_(True), beta.cpu().double().requires_grad_(True)
    t = torch.nn.functional.batch_norm(torch.nn.functional.conv2d(xc, wc, bias.cpu().double() if bias is not None else None, stride, pad),
                                       None, None, gc, bc, True, 0.1, 1e-5)
    if relu:
        t = torch.relu(t)
    if pool:
        t = torch.nn.functional.max_pool2d(t, k, s, p)
    t.backward(gy.cpu().double().permute(0, 3, 1, 2))
    assert _rel(dw2[:K * K].cpu().double(), wc.grad.permute(2, 3, 1, 0).reshape(K * K, Co)) <= 2e-4
    assert _rel(gg2.cpu().double(), gc.grad) <= 1e-4 and _rel(gb2.cpu().double(), bc.grad) <= 1e-4


@pytest.mark.parametrize("pool", [True, False])
def test_stem_as_one_autograd_node_equals_two(pool):
    """conv_bn_act on a 1-channel convolution: nn_norm._StemFunction (DSF_C1_BN) against the convolution and the BatchNorm as separate
    nodes -- outputs and running statistics bit for bit, gradients to the order of the double atomics; applied twice in one pass"""
    from dsf_amd import nn_norm, nn_conv, _lib as L
    from dsf_amd.model import backbone
    if L.deterministic():
        pytest.skip("deterministic mode keeps the separate layers")
    Ls = backbone._Layers()
    torch.manual_seed(8)
    pre = nn.Sequential(Ls.Conv2d(1, 64, kernel_size=5, stride=1, padding=2, bias=False), *Ls.bn_relu(64, momentum=0.1),
                        Ls.MaxPool2d(kernel_size=3, stride=2, padding=1)).cuda().train()
    ref = copy.deepcopy(pre)
    imgs = [torch.randn(4, 1, 32, 32, device="cuda"), torch.randn(4, 1, 32, 32, device="cuda")]
    floats = 2 * nn_norm.stat_floats(pre)
    run = (lambda m, t: backbone._stem(m, t)) if pool else (lambda m, t: nn_norm.conv_bn_act(m[0], m[1], t))
    res = []
    for m, on in ((pre, True), (ref, False)):
        saved = nn_norm.C1_BN[0]
        nn_norm.C1_BN[0] = on
        nn_conv.RECORD = []
        try:
            with nn_norm.stat_pool(floats, "cuda"):
                ya, yb = run(m, imgs[0]), run(m, imgs[1])
                grads = torch.autograd.grad((ya * ya).sum() + yb.sum(), [q for q in m.parameters()])
            kinds = [r[0] for r in nn_conv.RECORD]
        finally:
            nn_norm.C1_BN[0] = saved
            nn_conv.RECORD = None
        assert ("c1_fwd_bn" in kinds) == on and (("c1_wrw_bn1" if pool else "c1_wrw_bn0") in kinds) == on and ("c1_wrw" in kinds) == (not on)
        res.append((ya, yb, grads, m[1].running_mean.clone(), m[1].running_var.clone()))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    for u, v in zip(a[2], b[2]):
        assert u.shape == v.shape and _rel(u, v) <= 5e-6
