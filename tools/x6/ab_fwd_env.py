"""A/B of an environment switch the convolution launcher reads per call (DSF_X6P_NSLOW, DSF_X6P_BD, ..) on single forward-type layers:
alternating graph replays of 20 launches on one box, outputs compared bit for bit.
    python tools/x6/ab_fwd_env.py VAR a b [c ..]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsf_amd import _lib as L
x6 = L.lib()
I = ctypes.c_int
P = lambda t: ctypes.c_void_p(t.data_ptr())
var, values = sys.argv[1], sys.argv[2:]
LAYERS = [  # B, H, Ci, Co, K, stride, pad
    (32, 64, 256, 256, 4, 2, 1), (32, 32, 256, 256, 4, 2, 1), (32, 16, 256, 512, 4, 2, 1), (192, 64, 256, 256, 4, 2, 1), (192, 32, 256, 256, 4, 2, 1), (64, 64, 256, 256, 4, 2, 1),
    (32, 64, 488, 256, 3, 1, 1), (32, 64, 256, 488, 3, 1, 1), (64, 64, 488, 256, 3, 1, 1), (32, 64, 64, 64, 3, 1, 1), (32, 32, 128, 128, 3, 1, 1),
    (32, 16, 256, 256, 3, 1, 1), (32, 8, 512, 512, 3, 1, 1), (32, 32, 256, 256, 3, 1, 1), (192, 32, 128, 128, 3, 1, 1), (192, 16, 256, 256, 3, 1, 1)]
for (B, H, Ci, Co, K, stride, pad) in LAYERS:
    torch.manual_seed(0)
    Ho = (H + 2 * pad - K) // stride + 1
    x = torch.randn(B, H, H, Ci, device="cuda")
    wk = (torch.randn(K, K, Ci, Co, device="cuda") * (2.0 / (K * K * Ci)) ** 0.5).contiguous()
    img = torch.empty(x6.dsf_conv_x6_image_bytes(I(K), I(K), I(Ci), I(Co)), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream(); st = ctypes.c_void_p(s.cuda_stream)
    ys = {v: torch.empty(B, Ho, Ho, Co, device="cuda") for v in values}

    def f(v):
        os.environ[var] = v
        rc = x6.dsf_conv_x6_forward(P(x), P(img), None, P(ys[v]), I(B), I(H), I(H), I(Ci), I(Ho), I(Ho), I(Co), I(K), I(K), I(stride), I(1), I(pad), I(pad), I(0), st)
        assert rc == 0, rc
    with torch.cuda.stream(s):
        assert x6.dsf_conv_x6_split_weights(P(wk), P(img), I(K), I(K), I(Ci), I(Co), I(0), st) == 0
        for v in values: f(v)
    torch.cuda.synchronize()
    graphs = {}
    for v in values:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            f(v); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(20): f(v)
        graphs[v] = g
    best = {v: 1e9 for v in values}
    for rep in range(6):
        for v in values:
            graphs[v].replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); graphs[v].replay(); e1.record(); torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * B * Ho * Ho * Co * K * K * Ci
    same = all(torch.equal(ys[values[0]], ys[v]) for v in values[1:])
    rel = max(((ys[values[0]] - ys[v]).abs().max() / ys[values[0]].abs().max()).item() for v in values[1:])
    print("B%d %dx%dx%d->%d k%d s%d: " % (B, H, H, Ci, Co, K, stride) + " | ".join("%s=%s %7.1f us %6.1f TF" % (var, v, best[v], fl / best[v] / 1e6) for v in values)
          + ("   bitwise equal" if same else "   outputs differ by %.1e of the largest" % rel), flush=True)
