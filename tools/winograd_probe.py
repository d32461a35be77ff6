"""Winograd F(2x2, 3x3) for the 3x3 stride-1 layers: probe, bound, decision (round-3 verdict item 6; the reference runs
cudnn.benchmark = True, train_render.py:87, which may pick Winograd for such layers).

For two layers of the config-2 net (64 -> 64 at 64 x 64 and 128 -> 128 at 32 x 32, B = 32) this measures
  (1) the direct split-operand kernel (what ships),
  (2) the best case of a NON-FUSED Winograd built from the same kernels: the 16 element-wise GEMMs of the transformed domain,
      timed as ONE launch of the shipped 1 x 1 kernel over the same M x K x N volume (16 x B x H/2 x W/2 rows; a real
      implementation needs 16 weight matrices, i.e. can only be slower), plus the two transform passes priced at the rate a
      streaming copy of the same tensors reaches on this GPU (input transform: read X, write 4 X; output transform: read 4 Y,
      write Y -- 1.25 x the bytes of a copy of the 4 x tensor each),
  (3) the arithmetic error of F(2x2, 3x3) evaluated in fp32 (transforms and element-wise products in fp32, as a kernel
      would) against float64, next to the direct convolution's.
A FUSED form (transforms in the loader and the epilogue of one kernel) needs the 16 transformed-domain accumulators of a
tile at once: 4 x the accumulator registers of the direct kernel per output pixel (64 -> 256 per thread for the 128 x 128
tile), i.e. a 32 x 64 tile per workgroup, whose operand fragments are re-read from LDS 4 x as often per MFMA -- on kernels
whose loaders already take 5 of the ~6 issue slots an MFMA hides.  Not built; (2) is its upper bound
on what the arithmetic saving can buy once the transforms have to go through memory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsf_amd import nn_conv

BT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
AT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])


def winograd(x, w, dtype):
    """F(2x2, 3x3), padding 1, every step in `dtype`"""
    x, w = x.to(dtype), w.to(dtype)
    bt, g, at = BT.to(x), G.to(x), AT.to(x)
    B, C, H, W = x.shape
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                              # (B, C, H/2, W/2, 4, 4)
    V = bt @ d @ bt.T
    U = g @ w @ g.T                                                     # (Co, Ci, 4, 4)
    M = torch.einsum("oiuv,bityuv->botyuv", U, V)
    Y = at @ M @ at.T                                                   # (B, Co, H/2, W/2, 2, 2)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], H, W)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


B = 32
print("%-22s %10s %10s %12s %12s %12s | %s" % ("layer", "direct us", "16 GEMM us", "transforms", "non-fused", "vs direct", "max err / max|ref|: direct fp32, Winograd fp32"))
for C, H in ((64, 64), (128, 32)):
    torch.manual_seed(0)
    conv3 = nn_conv.Conv2d(C, C, 3, 1, 1, bias=False).cuda()
    conv1 = nn_conv.Conv2d(C, C, 1, 1, 0, bias=False).cuda()
    nn_conv.manage_weights(list(conv3.parameters()) + list(conv1.parameters()))
    x = torch.randn(B, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    xt = torch.randn(B, C, 2 * H, 2 * H, device="cuda").contiguous(memory_format=torch.channels_last)     # 16 x (H/2 x W/2) rows per sample
    with torch.no_grad():
        t_direct = timed(lambda: conv3(x))
        t_gemm = timed(lambda: conv1(xt))
        t_copy = timed(lambda: xt.clone())                              # reads + writes the 4 x tensor once each
        t_tr = 2 * (1.25 / 2.0) * t_copy                                # input + output transform at that streaming rate
        # arithmetic error on a smaller batch (the einsum of the transformed domain is done by torch)
        xs = x[:4].float()
        w = conv3.weight.detach().float()
        ref = torch.nn.functional.conv2d(xs.double(), w.double(), padding=1)
        e_direct = float((conv3(xs.contiguous(memory_format=torch.channels_last)).double() - ref).abs().max() / ref.abs().max())
        e_wino = float((winograd(xs, w, torch.float32).double() - ref).abs().max() / ref.abs().max())
        assert float((winograd(xs, w, torch.float64) - ref).abs().max() / ref.abs().max()) < 1e-12
    total = t_gemm + t_tr
    print("%3d -> %3d at %2d x %2d   %10.1f %10.1f %12.1f %12.1f %11.2fx | %.2e  %.2e (%.1f x)" % (C, C, H, H, t_direct, t_gemm, t_tr, total, total / t_direct, e_direct, e_wino, e_wino / e_direct))
print("decision rule (verdict): adopt only if >= 15 % faster on these layers at <= 3 x the direct kernel's float64 error")
