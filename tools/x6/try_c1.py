"""1-channel direct kernels (conv_c1.hip) vs the generic implicit-GEMM path: timing of the ResNet stem shapes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsf_amd import nn_conv
CL = torch.channels_last

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (B, H, Co, K, s, p) in ((32, 128, 64, 5, 1, 2), (64, 128, 64, 7, 2, 3)):
    x = torch.randn(B, 1, H, H, device="cuda").contiguous(memory_format=CL)
    wk = torch.randn(K, K, 1, Co, device="cuda")
    Ho = (H + 2 * p - K) // s + 1
    gy = torch.randn(B, Co, Ho, Ho, device="cuda").contiguous(memory_format=CL)
    print(f"B{B} {H}x{H}x1->{Co} k{K} s{s}: fwd c1 {timeit(lambda: nn_conv._fwd_c1(x, wk, None, (Ho, Ho), Co, K, s, p)):6.1f} us"
          f" | generic {timeit(lambda: nn_conv._fwd(x, wk, None, (Ho, Ho), Co, K, K, s, 1, (p, p))):6.1f} us"
          f" || wrw c1 {timeit(lambda: nn_conv._wrw_c1(x, gy, K, s, p)):6.1f} us | generic {timeit(lambda: nn_conv._wrw(x, gy, K, K, s, (p, p))):6.1f} us")
