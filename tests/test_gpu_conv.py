"""K11 fp32 MFMA implicit-GEMM convolution vs torch's CPU convolution (fp32), forward and all
three gradients.  Tolerance: 1e-4 of the largest reference magnitude (accumulation-order noise only;
the f32 MFMA is an exact fmaf chain)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # Ci, Co, K, stride, pad, H, bias
    (64, 64, 3, 1, 1, 16, False),
    (1, 64, 5, 1, 2, 32, False),          # stem (scalar gather path)
    (64, 128, 3, 2, 1, 16, False),
    (64, 128, 1, 2, 0, 16, False),        # downsample 1x1 stride 2
    (488, 256, 3, 1, 1, 8, True),         # stage-2 fusion conv (Ci % 32 != 0)
    (256, 63, 1, 1, 0, 8, True),          # offset head (Co % 4 != 0)
    (256, 21, 1, 1, 0, 8, True),
    (1, 64, 7, 2, 3, 32, True),           # hourglass stem
    (64, 1, 7, 1, 0, 22, True),           # generator output conv (Co = 1)
    (128, 256, 3, 1, 1, 9, False),        # odd spatial size -> M tail
    (1, 36, 5, 2, 2, 29, True),           # 1-channel direct kernels (conv_c1.hip): partial lanes, ragged row segments
    (1, 64, 7, 1, 3, 21, False),
    (1, 8, 5, 1, 0, 13, True),
]


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.mark.parametrize("Ci,Co,K,s,p,H,bias", CASES)
def test_conv2d_matches_torch_cpu(Ci, Co, K, s, p, H, bias):
    from dsf_amd.nn_conv import Conv2dFunction
    g = torch.Generator().manual_seed(Ci * 131 + Co * 7 + K)
    B = 3
    x = torch.randn(B, Ci, H, H, generator=g, requires_grad=True)
    w = (torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5).requires_grad_(True)
    b = torch.randn(Co, generator=g).requires_grad_(True) if bias else None
    y = F.conv2d(x, w, b, stride=s, padding=p)
    gy = torch.randn(y.shape, generator=g)
    grads = torch.autograd.grad((y * gy).sum(), [x, w] + ([b] if bias else []))
    xg = x.detach().cuda().requires_grad_(True)
    wg = w.detach().cuda().requires_grad_(True)
    bg = b.detach().cuda().requires_grad_(True) if bias else None
    yg = Conv2dFunction.apply(xg, wg, bg, s, (p, p))
    assert yg.shape == y.shape
    assert _rel(yg.cpu(), y.detach()) < 1e-4
    gg = torch.autograd.grad((yg * gy.cuda()).sum(), [xg, wg] + ([bg] if bias else []))
    for a, r in zip(gg, grads):
        assert _rel(a.cpu(), r) < 1e-4


@pytest.mark.parametrize("Ci,Co,K,s,p,op,H", [(512, 256, 4, 2, 1, 0, 4), (256, 128, 3, 2, 1, 1, 8), (64, 32, 4, 2, 1, 0, 7)])
def test_conv_transpose2d_matches_torch_cpu(Ci, Co, K, s, p, op, H):
    from dsf_amd.nn_conv import ConvTranspose2dFunction
    g = torch.Generator().manual_seed(Ci + Co + K)
    B = 2
    x = torch.randn(B, Ci, H, H, generator=g, requires_grad=True)
    w = (torch.randn(Ci, Co, K, K, generator=g) / (Ci * K * K) ** 0.5).requires_grad_(True)
    b = torch.randn(Co, generator=g).requires_grad_(True)
    y = F.conv_transpose2d(x, w, b, stride=s, padding=p, output_padding=op)
    gy = torch.randn(y.shape, generator=g)
    grads = torch.autograd.grad((y * gy).sum(), [x, w, b])
    xg, wg, bg = (t.detach().cuda().requires_grad_(True) for t in (x, w, b))
    yg = ConvTranspose2dFunction.apply(xg, wg, bg, s, (p, p), (op, op))
    assert yg.shape == y.shape
    assert _rel(yg.cpu(), y.detach()) < 1e-4
    for a, r in zip(torch.autograd.grad((yg * gy.cuda()).sum(), [xg, wg, bg]), grads):
        assert _rel(a.cpu(), r) < 1e-4


@pytest.mark.parametrize("C,H,res,relu", [(64, 16, True, True), (256, 8, False, True), (512, 4, True, False), (8, 5, False, False),
                                          (1024, 3, True, True), (2048, 4, True, True), (2048, 3, False, False)])
def test_fused_batchnorm_matches_torch_cpu(C, H, res, relu):
    """bn(x) (+ residual) (relu) in training mode vs torch CPU: output, running stats, all gradients."""
    from dsf_amd.nn_norm import FusedBatchNorm2d
    g = torch.Generator().manual_seed(C + H)
    B = 6
    x = (torch.randn(B, C, H, H, generator=g) * 2 + 0.5).requires_grad_(True)
    r = torch.randn(B, C, H, H, generator=g).requires_grad_(True) if res else None
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
    gy = torch.randn(y.shape, generator=g)
    inputs = [x, ref.weight, ref.bias] + ([r] if res else [])
    grads = torch.autograd.grad((y * gy).sum(), inputs)
    xg = x.detach().cuda().requires_grad_(True)
    rg = r.detach().cuda().requires_grad_(True) if res else None
    yg = fused(xg, rg, relu)
    assert _rel(yg.cpu(), y.detach()) < 1e-5
    gin = [xg, fused.weight, fused.bias] + ([rg] if res else [])
    gg = torch.autograd.grad((yg * gy.cuda()).sum(), gin)
    for a, b_ in zip(gg, grads):
        assert _rel(a.cpu(), b_) < 1e-4
    assert _rel(fused.running_mean.cpu(), ref.running_mean) < 1e-5
    assert _rel(fused.running_var.cpu(), ref.running_var) < 1e-5
    assert int(fused.state_dict()["num_batches_tracked"]) == 1            # deferred add, folded in when the state is read
    # eval mode: frozen statistics
    ref.eval(); fused.eval()
    with torch.no_grad():
        ye = ref(x)
        assert _rel(fused(xg.detach()).cpu(), ye) < 1e-5


# ---- conv_x6: fp32 products on the bf16 matrix cores by exact three-way operand splitting --------------------------------
X6_CASES = [
    # Ci, Co, K, stride, pad, H, B
    (488, 256, 3, 1, 1, 16, 4),           # stage-2 fusion conv: Ci % 16 != 0 (zero-padded last chunk), two n tiles
    (64, 64, 3, 1, 1, 64, 2),
    (256, 84, 1, 1, 0, 32, 2),            # merged heads: ragged n tile
    (20, 36, 3, 2, 1, 17, 3),             # ragged everything, stride 2
    (512, 512, 3, 1, 1, 8, 32),           # small map: split-K with float atomics
    (256, 256, 4, 2, 1, 32, 2),           # ConvTranspose2d backward-data geometry
]


@pytest.mark.parametrize("Ci,Co,K,s,p,H,B", X6_CASES)
def test_x6_matches_float64_as_closely_as_the_fp32_mfma(Ci, Co, K, s, p, H, B, monkeypatch):
    """Both kernels against a float64 convolution: the split path must be in the same error class as the fp32 MFMA
    (its products are exact to 2^-26; what is left is fp32 accumulation order), far inside the 1e-4 parity bar."""
    from dsf_amd import nn_conv
    g = torch.Generator().manual_seed(Ci + 3 * Co + K)
    x = torch.randn(B, Ci, H, H, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5).cuda().requires_grad_(True)
    gy = None
    out = {}
    for math in ("x6", "f32"):
        monkeypatch.setattr(nn_conv, "MATH", math)
        nn_conv.RECORD = []
        y = nn_conv.Conv2dFunction.apply(x, w, None, s, (p, p))
        gy = torch.randn(y.shape, generator=g).cuda() if gy is None else gy
        gx, gw = torch.autograd.grad((y * gy).sum(), [x, w])
        kinds = {r[0] for r in nn_conv.RECORD}
        nn_conv.RECORD = None
        assert ("x6" in kinds) == (math == "x6")
        out[math] = (y.detach().double().cpu(), gx.double().cpu(), gw.double().cpu())
    xd = x.detach().double().cpu().requires_grad_(True)
    wd = w.detach().double().cpu().requires_grad_(True)
    yd = F.conv2d(xd, wd, None, stride=s, padding=p)
    gxd, gwd = torch.autograd.grad((yd * gy.double().cpu()).sum(), [xd, wd])
    for i, ref in enumerate((yd.detach(), gxd, gwd)):
        e6, e32 = _rel(out["x6"][i], ref), _rel(out["f32"][i], ref)
        assert e6 < 2e-6, (i, e6)
        assert e6 < 3 * e32 + 1e-7, (i, e6, e32)


def test_x6_weight_images_follow_the_weights(monkeypatch):
    """The split image of a weight is kept from one use to the next only for MANAGED parameters (FusedAdamW's, EvalStep's):
    in-place torch updates (version counter) and FusedAdamW's raw-pointer updates (nn_conv.weights_changed) both invalidate
    it, an unchanged managed weight is not split again.  An unmanaged parameter is re-split at every use, so a write
    through ``.data`` -- invisible to torch's version counter -- can never be served from a stale image."""
    from dsf_amd import nn_conv
    from dsf_amd.optim import FusedAdamW
    monkeypatch.setattr(nn_conv, "MATH", "x6")            # (the suite also runs under DSF_CONV_MATH=f32)
    torch.manual_seed(0)
    conv = nn_conv.Conv2d(32, 48, 3, padding=1, bias=False).cuda()
    x = torch.randn(2, 32, 12, 12, device="cuda")
    ref = lambda: F.conv2d(x.double(), conv.weight.detach().double(), padding=1)
    # unmanaged: forward, write through .data, forward must change
    y0 = conv(x)
    v0 = conv.weight._version
    conv.weight.data.mul_(2.0)
    assert conv.weight._version == v0                       # torch did not see the write ...
    y1 = conv(x)
    assert _rel(y1.double(), 2.0 * y0.double()) < 1e-6 and _rel(y1.double(), ref()) < 2e-6      # ... the layer did
    # init_weights-style re-initialisation after a forward
    with torch.no_grad():
        conv.weight.normal_(0, 0.1)
    assert _rel(conv(x).double(), ref()) < 2e-6
    # managed: cached until announced
    opt = FusedAdamW(conv.parameters(), lr=0.1)
    assert conv.weight.__dict__["_dsf_managed"]
    y0 = conv(x)
    img = conv.weight.__dict__["_dsf_x6"][0]
    conv(x)
    assert conv.weight.__dict__["_dsf_x6"][0] is img and _rel(y0.double(), ref()) < 2e-6     # same cache entry: no re-split
    with torch.no_grad():
        conv.weight.mul_(-2.0)
    assert _rel(conv(x).double(), ref()) < 2e-6
    conv(x).square().mean().backward()
    opt.step()
    assert _rel(conv(x).double(), ref()) < 2e-6
    conv.weight.data.mul_(0.5)
    nn_conv.weights_changed()                               # the documented way to announce a .data write on managed weights
    assert _rel(conv(x).double(), ref()) < 2e-6


def test_x6_random_geometries_against_float64():
    """Seeded sweep over ragged geometries (non-square maps, odd sizes, every channel-count class the split kernels accept,
    kernel 1..5, stride 1 / 2, transposed convolutions): forward and all gradients against float64."""
    from dsf_amd import nn_conv
    rng = np.random.RandomState(7)
    saved, nn_conv.RECORD = nn_conv.RECORD, []
    try:
        for case in range(36):
            Ci = int(rng.choice([16, 20, 36, 64, 100, 132]))
            Co = int(rng.choice([1, 4, 36, 64, 72, 130]))
            K = int(rng.choice([1, 2, 3, 4, 5]))
            s = int(rng.choice([1, 2]))
            p = int(rng.randint(0, K))
            H, W, B = int(rng.randint(K + 1, 20)), int(rng.randint(K + 1, 20)), int(rng.choice([1, 3]))
            transposed = case % 3 == 2 and Co % 4 == 0
            g = torch.Generator().manual_seed(case)
            x = torch.randn(B, Ci, H, W, generator=g).cuda().requires_grad_(True)
            if transposed:
                op = int(rng.randint(0, s))
                w = (torch.randn(Ci, Co, K, K, generator=g) / (Ci * K * K) ** 0.5).cuda().requires_grad_(True)
                if (H - 1) * s - 2 * p + K + op < 1 or (W - 1) * s - 2 * p + K + op < 1:
                    continue
                y = nn_conv.ConvTranspose2dFunction.apply(x, w, None, s, (p, p), (op, op))
                ref_fn = lambda xd, wd: F.conv_transpose2d(xd, wd, None, stride=s, padding=p, output_padding=op)
            else:
                w = (torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5).cuda().requires_grad_(True)
                y = nn_conv.Conv2dFunction.apply(x, w, None, s, (p, p))
                ref_fn = lambda xd, wd: F.conv2d(xd, wd, None, stride=s, padding=p)
            gy = torch.randn(y.shape, generator=g).cuda()
            gx, gw = torch.autograd.grad((y * gy).sum(), [x, w])
            xd = x.detach().double().cpu().requires_grad_(True)
            wd = w.detach().double().cpu().requires_grad_(True)
            yd = ref_fn(xd, wd)
            assert yd.shape == y.shape, (case, yd.shape, y.shape)
            gxd, gwd = torch.autograd.grad((yd * gy.double().cpu()).sum(), [xd, wd])
            for name, a, r in (("y", y, yd), ("gx", gx, gxd), ("gw", gw, gwd)):
                assert _rel(a.detach().double().cpu(), r.detach()) < 3e-6, (case, name, Ci, Co, K, s, p, H, W, B, transposed)
        kinds = [r[0] for r in nn_conv.RECORD]
        assert kinds.count("x6") >= 30, kinds          # the sweep really exercised the split kernels
    finally:
        nn_conv.RECORD = saved
