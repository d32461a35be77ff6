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
