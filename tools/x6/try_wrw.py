"""Experiment: bf16x6 backward-weights (csrc/conv_x6.hip) against the fp32-MFMA kernel and a float64 reference."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsf_amd import nn_conv, _lib as L
from dsf_amd._lib import I, ptr, check, stream_ptr
CL = torch.channels_last


def run(B, H, Ci, Co, k, stride, pad, check_ref=True):
    torch.manual_seed(0)
    x = torch.randn(B, Ci, H, H, device="cuda").contiguous(memory_format=CL)
    Ho = (H + 2 * pad - k) // stride + 1
    gy = torch.randn(B, Co, Ho, Ho, device="cuda").contiguous(memory_format=CL)
    dw = torch.empty(k, k, Ci, Co, device="cuda")
    def f6():
        check(L.lib().dsf_conv_x6_wrw(nn_conv.ptr_nhwc(x), nn_conv.ptr_nhwc(gy), ptr(dw), I(B), I(H), I(H), I(Ci), I(Ho), I(Ho), I(Co),
                                      I(k), I(k), I(stride), I(pad), I(pad), I(0), stream_ptr()), "x6_wrw")
    saved, nn_conv.MATH = nn_conv.MATH, "f32"
    f32 = lambda: nn_conv._wrw(x, gy, k, k, stride, (pad, pad))
    f6(); d32 = f32(); torch.cuda.synchronize()
    msg = ""
    if check_ref:
        w = torch.zeros(Co, Ci, k, k, dtype=torch.float64, device="cuda", requires_grad=True)
        y = torch.nn.functional.conv2d(x.double(), w, None, stride=stride, padding=pad)
        ref, = torch.autograd.grad((y * gy.double()).sum(), [w])
        ref = ref.permute(2, 3, 1, 0)
        sc = ref.abs().mean()
        msg = f"max err / mean|dw|: x6 {((dw.double()-ref).abs().max()/sc).item():.2e}  fp32-mfma {((d32.double()-ref).abs().max()/sc).item():.2e}"
    def timeit(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t6, t32 = timeit(f6), timeit(f32)
    nn_conv.MATH = saved
    fl = 2.0 * B * Ho * Ho * Co * k * k * Ci
    print(f"wrw B{B} {H}x{H}x{Ci}->{Co} k{k} s{stride}: x6 {t6:7.1f} us {fl/t6/1e6:6.1f} TF | fp32 {t32:7.1f} us {fl/t32/1e6:6.1f} TF | {msg}", flush=True)


if __name__ == "__main__":
    run(3, 17, 20, 36, 3, 2, 1)
    run(2, 9, 12, 132, 5, 1, 2)
    run(32, 64, 488, 256, 3, 1, 1, check_ref=False)
    run(32, 64, 256, 256, 4, 2, 1, check_ref=False)
    run(32, 64, 64, 64, 3, 1, 1)
    run(32, 32, 128, 128, 3, 1, 1)
    run(32, 16, 256, 256, 3, 1, 1)
    run(32, 8, 512, 512, 3, 1, 1)
    run(32, 64, 256, 84, 1, 1, 0)
    run(32, 64, 64, 128, 3, 2, 1)
