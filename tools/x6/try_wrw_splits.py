"""Backward-weights launches of the small-map / 64-channel layers of config 2 alone (host-free graph replays) under different
workgroup targets of the pixel split (DSF_X6_WRW_WGS, read once per process: one child process per value) and in deterministic mode
(ordered partial tiles + the reduce launch instead of float atomics).   python tools/x6/try_wrw_splits.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from dsf_amd import nn_conv, _lib as L
CL = torch.channels_last
def replay_us(fn, n=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best
det = os.environ.get("DSF_DETERMINISTIC", "0") == "1"
out = []
for (B, Ci, Co, H, K, s, p) in [(32, 128, 128, 32, 3, 1, 1), (32, 256, 256, 16, 3, 1, 1), (32, 512, 512, 8, 3, 1, 1), (32, 64, 64, 64, 3, 1, 1), (32, 256, 256, 64, 4, 2, 1),
                                (32, 128, 256, 32, 3, 2, 1)]:
    Ho = (H + 2 * p - K) // s + 1
    x = torch.randn(B, Ci, H, H, device="cuda").contiguous(memory_format=CL)
    gy = torch.randn(B, Co, Ho, Ho, device="cuda").contiguous(memory_format=CL)
    us = replay_us(lambda: nn_conv._wrw(x, gy, K, K, s, (p, p)))
    fl = 2.0 * B * Ho * Ho * Co * Ci * K * K
    out.append("%%dx%%dx%%d->%%d k%%d s%%d: %%6.1f us %%6.1f TF" %% (H, H, Ci, Co, K, s, us, fl / us / 1e6))
print("WGS=%%s det=%%d | " %% (os.environ.get("DSF_X6_WRW_WGS", "512"), det) + " | ".join(out), flush=True)
''' % ROOT
for env in ({"DSF_X6_WRW_WGS": "256"}, {"DSF_X6_WRW_WGS": "384"}, {}, {"DSF_X6_WRW_WGS": "768"}, {"DSF_X6_WRW_WGS": "1024"}, {"DSF_X6_WRW_WGS": "2048"},
            {"DSF_DETERMINISTIC": "1"}, {"DSF_DETERMINISTIC": "1", "DSF_X6_WRW_WGS": "256"}, {"DSF_X6_WRW_PATCH": "2"}):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
    print((r.stdout.strip() or r.stderr[-500:]) + ("   [%s]" % env if env else ""), flush=True)
