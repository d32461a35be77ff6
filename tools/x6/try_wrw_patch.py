"""A/B: the row-staged 3 x 3 backward-weights kernel (igemm_wrw_x6p_kernel) against igemm_wrw_x6_kernel (DSF_X6_WRW_PATCH=0):
both against float64 on the small cases, agreement with each other, and alternating timings (graph replay of 10 launches)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsf_amd import _lib as L
lib = L.lib()
I = ctypes.c_int
P = lambda t: ctypes.c_void_p(t.data_ptr())
CL = torch.channels_last


def run(B, H, W, Ci, Co, ref=False):
    torch.manual_seed(0)
    x = torch.randn(B, Ci, H, W, device="cuda").contiguous(memory_format=CL)
    gy = torch.randn(B, Co, H, W, device="cuda").contiguous(memory_format=CL)
    dws = {k: torch.empty(3, 3, Ci, Co, device="cuda") for k in ("0", "1")}
    s = torch.cuda.Stream()
    st = ctypes.c_void_p(s.cuda_stream)
    torch.cuda.synchronize()

    def f(kind):
        os.environ["DSF_X6_WRW_PATCH"] = kind
        rc = lib.dsf_conv_x6_wrw(P(x), P(gy), P(dws[kind]), I(B), I(H), I(W), I(Ci), I(H), I(W), I(Co), I(3), I(3), I(1), I(1), I(1), I(0), st)
        assert rc == 0, rc
    with torch.cuda.stream(s):
        f("0"); f("1")
    torch.cuda.synchronize()
    sc = dws["0"].abs().mean()
    diff = ((dws["0"] - dws["1"]).abs().max() / sc).item()
    msg = ""
    if ref:
        w = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
        y = torch.nn.functional.conv2d(x.double(), w, None, padding=1)
        r, = torch.autograd.grad((y * gy.double()).sum(), [w])
        r = r.permute(2, 3, 1, 0)
        msg = " | err vs f64: old %.2e new %.2e" % tuple(((dws[k].double() - r).abs().max() / r.abs().mean()).item() for k in ("0", "1"))
    graphs = {}
    for kind in ("0", "1"):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            f(kind)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(10):
                    f(kind)
        graphs[kind] = g
    best = {"0": 1e9, "1": 1e9}
    for rep in range(5):
        for kind in ("0", "1"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            graphs[kind].replay(); torch.cuda.synchronize()
            e0.record(); graphs[kind].replay(); e1.record(); torch.cuda.synchronize()
            best[kind] = min(best[kind], e0.elapsed_time(e1) / 10 * 1e3)
    fl = 2.0 * B * H * W * Co * 9 * Ci
    print(f"B{B} {H}x{W}x{Ci}->{Co}: old {best['0']:7.1f} us {fl/best['0']/1e6:6.1f} TF | rows {best['1']:7.1f} us {fl/best['1']/1e6:6.1f} TF | "
          f"diff {diff:.1e}{msg}", flush=True)
    return diff < 2e-5


if __name__ == "__main__":
    ok = True
    ok &= run(2, 8, 64, 36, 132, ref=True)
    ok &= run(1, 5, 64, 20, 200, ref=True)
    ok &= run(3, 7, 32, 40, 72, ref=True)
    ok &= run(5, 16, 16, 36, 132, ref=True)
    ok &= run(3, 3, 16, 32, 128, ref=True)
    ok &= run(2, 8, 64, 36, 60, ref=True)
    ok &= run(3, 6, 32, 64, 64, ref=True)
    ok &= run(32, 64, 64, 64, 64)
    ok &= run(32, 64, 64, 256, 64)
    ok &= run(64, 64, 64, 64, 64)
    ok &= run(32, 64, 64, 488, 256)
    ok &= run(32, 64, 64, 256, 488)
    ok &= run(32, 64, 64, 256, 256)
    ok &= run(32, 64, 64, 64, 256)
    ok &= run(32, 32, 32, 128, 128)
    ok &= run(32, 16, 16, 256, 256)
    ok &= run(64, 32, 32, 128, 128)
    ok &= run(64, 16, 16, 256, 256)
    print("ALL AGREE" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
