"""The stem's BatchNorm + ReLU + MaxPool2d(3, 2, 1) on its own (B x 64 x 128 x 128): the pooled layer (dsf_bn_relu_pool_*) against the
separate layers, forward and backward timed with events (median of 20), per launch under rocprofv3 when run beneath it.
    python tools/bn_pool_probe.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsf_amd import nn_norm, nn_pool
from dsf_amd.nn_norm import FusedBatchNorm2d

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
C, H = 64, 128
x = torch.randn(B, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
bn = FusedBatchNorm2d(C, fuse_relu=True).cuda().train()
mp = nn_pool.MaxPool2d(3, 2, 1)
gy = torch.randn(B, C, H // 2, H // 2, device="cuda").contiguous(memory_format=torch.channels_last)
floats = 2 * nn_norm.acc_rows() * 2 * C


def timed(fwd):
    tf, tb = [], []
    for it in range(25):
        x.grad = None; bn.weight.grad = None; bn.bias.grad = None
        with nn_norm.stat_pool(floats, "cuda"):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(); y = fwd(x); e[1].record(); y.backward(gy); e[2].record()
        torch.cuda.synchronize()
        if it >= 5:
            tf.append(e[0].elapsed_time(e[1]) * 1e3); tb.append(e[1].elapsed_time(e[2]) * 1e3)
    tf.sort(); tb.sort()
    return tf[len(tf) // 2], tb[len(tb) // 2]


for name, f in (("separate layers", lambda t: mp(bn(t))), ("pooled layer", lambda t: bn.forward_pooled(t, 3, 2, 1)), ("separate layers", lambda t: mp(bn(t))),
                ("pooled layer", lambda t: bn.forward_pooled(t, 3, 2, 1))):
    a, b = timed(f)
    print("%-16s B=%d: forward %.1f us, backward %.1f us" % (name, B, a, b), flush=True)
