"""A/B: the patch-staged 3 x 3 kernel (igemm_x6p_kernel) against the per-tap gather kernel (igemm_x6b_kernel, DSF_X6_PATCH=0) on the
64-wide layers of the B = 32 step: bitwise equality of the outputs and alternating timings (graph replay of 20 launches)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsf_amd import _lib as L
x6 = L.lib()
I = ctypes.c_int
P = lambda t: ctypes.c_void_p(t.data_ptr())
CL = torch.channels_last


def run(B, H, Ci, Co, bias_on=False):
    torch.manual_seed(0)
    W = 64
    x = torch.randn(B, Ci, H, W, device="cuda").contiguous(memory_format=CL)
    wk = (torch.randn(3, 3, Ci, Co, device="cuda") * (2.0 / (9 * Ci)) ** 0.5).contiguous()
    bias = torch.randn(Co, device="cuda") if bias_on else None
    nbytes = x6.dsf_conv_x6_image_bytes(I(3), I(3), I(Ci), I(Co))
    img = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    st = ctypes.c_void_p(s.cuda_stream)
    ys = {k: torch.empty(B, Co, H, W, device="cuda").contiguous(memory_format=CL) for k in ("0", "1")}
    torch.cuda.synchronize()

    def f(kind):
        os.environ["DSF_X6_PATCH"] = kind
        rc = x6.dsf_conv_x6_forward(P(x), P(img), P(bias) if bias_on else None, P(ys[kind]), I(B), I(H), I(W), I(Ci), I(H), I(W), I(Co),
                                    I(3), I(3), I(1), I(1), I(1), I(1), I(1), st)
        assert rc == 0, rc
    with torch.cuda.stream(s):
        rc = x6.dsf_conv_x6_split_weights(P(wk), P(img), I(3), I(3), I(Ci), I(Co), I(0), st); assert rc == 0, rc
        f("0"); f("1")
    torch.cuda.synchronize()
    same = torch.equal(ys["0"], ys["1"])
    ref = torch.nn.functional.conv2d(x[:1].double(), wk.permute(3, 2, 0, 1).double(), bias.double() if bias_on else None, padding=1)
    err = ((ys["1"][:1].double() - ref).abs().max() / ref.abs().mean()).item()
    graphs = {}
    for kind in ("0", "1"):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            f(kind)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(20):
                    f(kind)
        graphs[kind] = g
    best = {"0": 1e9, "1": 1e9}
    for rep in range(6):
        for kind in ("0", "1"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            graphs[kind].replay(); torch.cuda.synchronize()
            e0.record(); graphs[kind].replay(); e1.record(); torch.cuda.synchronize()
            best[kind] = min(best[kind], e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * B * H * W * Co * 9 * Ci
    print(f"B{B} {H}x{W}x{Ci}->{Co}{' bias' if bias_on else ''}: gather {best['0']:7.1f} us {fl/best['0']/1e6:6.1f} TF | patch {best['1']:7.1f} us "
          f"{fl/best['1']/1e6:6.1f} TF | bitwise equal {same} | err vs f64 {err:.2e}", flush=True)
    return same


if __name__ == "__main__":
    ok = True
    ok &= run(2, 8, 64, 64)                  # two tiles per image: top and bottom borders
    ok &= run(1, 12, 20, 36, bias_on=True)   # ragged chunk (20 channels), ragged n tile, three tiles per image
    ok &= run(6, 64, 36, 130)                # 128-wide n tiles, odd chunk count
    ok &= run(4, 64, 488, 256, bias_on=True)
    ok &= run(32, 64, 64, 64)
    ok &= run(32, 64, 256, 64)
    ok &= run(32, 64, 64, 256)
    ok &= run(32, 64, 488, 256)
    ok &= run(32, 64, 256, 488)
    ok &= run(32, 64, 256, 256)
    print("ALL BITWISE EQUAL" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
