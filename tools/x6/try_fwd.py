"""Experiment: bf16x6 split convolution (csrc/conv_x6.hip) against the fp32-MFMA kernel and a float64 reference."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsf_amd import nn_conv, _lib as L
x6 = L.lib()
I = ctypes.c_int
P = lambda t: ctypes.c_void_p(t.data_ptr())
CL = torch.channels_last


def run(B, H, Ci, Co, k, stride, pad, splits=1, check=True):
    torch.manual_seed(0)
    x = torch.randn(B, Ci, H, H, device="cuda").contiguous(memory_format=CL)
    wk = (torch.randn(k, k, Ci, Co, device="cuda") * (2.0 / (k * k * Ci)) ** 0.5).contiguous()
    bias = torch.randn(Co, device="cuda")
    Ho = (H + 2 * pad - k) // stride + 1
    nbytes = x6.dsf_conv_x6_image_bytes(I(k), I(k), I(Ci), I(Co))
    img = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = x6.dsf_conv_x6_split_weights(P(wk), P(img), I(k), I(k), I(Ci), I(Co), I(0), st); assert rc == 0, rc
    y = torch.empty(B, Co, Ho, Ho, device="cuda").contiguous(memory_format=CL)
    def f6():
        rc = x6.dsf_conv_x6_forward(P(x), P(img), P(bias), P(y), I(B), I(H), I(H), I(Ci), I(Ho), I(Ho), I(Co), I(k), I(k),
                                    I(stride), I(1), I(pad), I(pad), I(splits), st)
        assert rc == 0, rc
    def f32():
        return nn_conv._fwd(x, wk, bias, (Ho, Ho), Co, k, k, stride, 1, (pad, pad))
    f6(); y32 = f32(); torch.cuda.synchronize()
    msg = ""
    if check:
        Bc = min(B, 2)
        ref = torch.nn.functional.conv2d(x[:Bc].double(), wk.permute(3, 2, 0, 1).double(), bias.double(), stride=stride, padding=pad)
        sc = ref.abs().mean()
        e6 = ((y[:Bc].double() - ref).abs().max() / sc).item()
        e32 = ((y32[:Bc].double() - ref).abs().max() / sc).item()
        msg = f"max err / mean|y|: x6 {e6:.2e}  fp32-mfma {e32:.2e}"
    def timeit(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t6, t32 = timeit(f6), timeit(f32)
    fl = 2.0 * B * Ho * Ho * Co * k * k * Ci
    print(f"B{B} {H}x{H}x{Ci}->{Co} k{k} s{stride} splits{splits}: x6 {t6:7.1f} us {fl/t6/1e6:6.1f} TF | fp32 {t32:7.1f} us {fl/t32/1e6:6.1f} TF | {msg}", flush=True)


if __name__ == "__main__":
    run(32, 64, 488, 256, 3, 1, 1)
    run(32, 64, 256, 256, 3, 1, 1)
    run(32, 64, 64, 64, 3, 1, 1)
    run(32, 32, 128, 128, 3, 1, 1)
    run(32, 16, 256, 256, 3, 1, 1)
    run(32, 16, 256, 256, 3, 1, 1, splits=4)
    run(32, 8, 512, 512, 3, 1, 1, splits=8)
    run(32, 64, 256, 84, 1, 1, 0)
    run(32, 64, 256, 256, 4, 2, 1)
    run(3, 17, 20, 36, 3, 2, 1)
    run(2, 9, 12, 130, 5, 1, 2, splits=3)
