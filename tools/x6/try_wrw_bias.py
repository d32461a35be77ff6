import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dsf_amd import _lib as L
lib = L.lib(); I = ctypes.c_int; P = lambda t: ctypes.c_void_p(t.data_ptr())
CL = torch.channels_last
os.environ["DSF_WRW_BIAS_MAX_SPLITS"] = "100000"
def run(B, H, W, Ci, Co, K, pad):
    x = torch.randn(B, Ci, H, W, device="cuda").contiguous(memory_format=CL)
    gy = torch.randn(B, Co, H, W, device="cuda").contiguous(memory_format=CL)
    dw = torch.zeros(K, K, Ci, Co, device="cuda"); db = torch.zeros(Co, device="cuda")
    s = torch.cuda.Stream(); st = ctypes.c_void_p(s.cuda_stream)
    def f(bias):
        if bias:
            rc = lib.dsf_conv_x6_wrw_bias(P(x), P(gy), P(dw), P(db), I(B), I(H), I(W), I(Ci), I(H), I(W), I(Co), I(K), I(K), I(1), I(pad), I(pad), I(1), st)
        else:
            rc = lib.dsf_conv_x6_wrw(P(x), P(gy), P(dw), I(B), I(H), I(W), I(Ci), I(H), I(W), I(Co), I(K), I(K), I(1), I(pad), I(pad), I(1), st)
        assert rc == 0, rc
    res = {}
    for bias in (False, True):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            f(bias); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(10): f(bias)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g.replay(); torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        res[bias] = best
    print(f"B{B} {H}x{W}x{Ci}->{Co} k{K}: wrw {res[False]:7.1f} us | wrw+bias {res[True]:7.1f} us", flush=True)
run(32, 64, 64, 256, 84, 1, 0)
run(32, 64, 64, 256, 256, 1, 0)
run(64, 32, 32, 128, 256, 1, 0)
run(64, 32, 32, 256, 128, 1, 0)
run(64, 32, 32, 128, 128, 3, 1)
run(64, 16, 16, 128, 128, 3, 1)
run(64, 4, 4, 128, 128, 3, 1)
run(32, 64, 64, 488, 256, 3, 1)
