"""Backward-weights A/B, interleaved in one process: the LDS-staged kernel (igemm_wrw_x6_kernel) against the dY-image path
(x6_split_dy_kernel + igemm_wrw_x6b_kernel) on the weight-gradient shapes of the B=32 ResNet-18 two-stage step.
Prints per shape: us staged, us direct (image pass included), TFLOP/s of both, max relative difference of the two results."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsf_amd import nn_conv
CL = torch.channels_last
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [  # Hi, Ci, Ho, Co, K, stride, pad, launches per step
    (64, 488, 64, 256, 3, 1, 1, 1), (64, 256, 32, 256, 4, 2, 1, 2), (32, 128, 32, 128, 3, 1, 1, 6), (16, 256, 16, 256, 3, 1, 1, 6),
    (8, 512, 8, 512, 3, 1, 1, 6), (32, 256, 16, 256, 4, 2, 1, 2), (64, 256, 64, 84, 1, 1, 0, 2), (16, 256, 8, 512, 4, 2, 1, 2),
    (64, 64, 32, 128, 3, 2, 1, 2), (16, 256, 8, 512, 3, 2, 1, 2), (32, 128, 16, 256, 3, 2, 1, 2), (64, 64, 64, 64, 3, 1, 1, 7),
    (64, 256, 64, 64, 3, 1, 1, 1), (64, 64, 64, 256, 3, 1, 1, 1), (64, 64, 64, 256, 1, 1, 0, 1)]
rounds, iters = 5, 5
tot = {False: 0.0, True: 0.0}
for Hi, Ci, Ho, Co, K, s, pad, n in SHAPES:
    x = torch.randn(B, Ci, Hi, Hi, device="cuda").contiguous(memory_format=CL)
    gy = torch.randn(B, Co, Ho, Ho, device="cuda").contiguous(memory_format=CL)
    res, t = {}, {False: [], True: []}
    for r in range(rounds):
        for direct in (False, True):
            nn_conv.WRW_DIRECT[0] = direct
            res[direct] = nn_conv._wrw(x, gy, K, K, s, (pad, pad))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                nn_conv._wrw(x, gy, K, K, s, (pad, pad))
            e1.record()
            torch.cuda.synchronize()
            t[direct].append(e0.elapsed_time(e1) * 1e3 / iters)
    fl = 2.0 * B * Ho * Ho * Co * Ci * K * K
    a, b = statistics.median(t[False]), statistics.median(t[True])
    tot[False] += a * n; tot[True] += b * n
    rel = float((res[True] - res[False]).abs().max() / res[False].abs().max())
    print("in%dx%dx%d out%dx%dx%d k%d s%d  n=%d  staged %7.1f us %6.1f TF | direct %7.1f us %6.1f TF | ratio %.3f  maxrel %.1e"
          % (Hi, Hi, Ci, Ho, Ho, Co, K, s, n, a, fl / a / 1e6, b, fl / b / 1e6, b / a, rel))
print("per step: staged %.0f us, direct %.0f us" % (tot[False], tot[True]))
