"""BatchNorm passes of config 2's layer shapes in isolation: microseconds and algorithmic TB/s per C-ABI call.

    python tools/bn_bw.py [--sets 12]

Each shape is timed with HIP events over 30 calls, once on ONE set of tensors (stays in the 256 MB Infinity Cache) and once
rotating over --sets sets (HBM).  Calls: forward = reduce + apply (dsf_bn_forward_acc, acc_filled 0), apply = the apply pass
alone (acc_filled 1: the statistics came from the convolution epilogue), backward = reduce + apply (dsf_bn_backward_acc) with
the ReLU mask recomputed from x (mode 2) or read from y with a residual gradient written (mode 1).
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd import _lib as L  # noqa: E402

I, I64, CF = ctypes.c_int, ctypes.c_int64, ctypes.c_float
SHAPES = [(32 * 64 * 64, 64), (32 * 32 * 32, 128), (32 * 16 * 16, 256), (32 * 8 * 8, 512), (32 * 32 * 32, 256), (32 * 64 * 64, 256)]


def timed(fn, sets, n=30):
    for k in range(3):
        fn(k % sets)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(n):
        fn(k % sets)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", type=int, default=12)
    ap.add_argument("--only", type=int, default=-1, help="index of the one shape to run")
    a = ap.parse_args()
    lib = L.lib()
    dev = torch.device("cuda:0")
    rows = int(lib.dsf_bn_acc_rows())
    p, st = L.ptr, L.stream_ptr
    print("%-16s %-9s %10s %8s %10s %8s" % ("M x C (MB)", "call", "cache us", "TB/s", "hbm us", "TB/s"))
    for M, C in (SHAPES if a.only < 0 else SHAPES[a.only:a.only + 1]):
        mb = M * C * 4 / 1e6
        nset = max(2, min(a.sets, int(3000 // (5 * mb))))
        T = [[torch.randn(M, C, device=dev) for _ in range(5)] for _ in range(nset)]     # x, res / gy, y, gx, gres
        g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        gg, gb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        acc = torch.zeros(rows * 2 * C, device=dev, dtype=torch.float64)
        filled = torch.zeros(rows * 2 * C, device=dev, dtype=torch.float64)
        filled.view(rows, 2, C)[:, 1] = M / rows                                          # mean 0, variance 1

        def fwd(s, res=False, fill=0):
            x, r, y = T[s][0], T[s][1], T[s][2]
            L.check(lib.dsf_bn_forward_acc(p(x), p(r) if res else None, p(g), p(b), I64(M), I(C), CF(1e-5), CF(0.1), I(1), p(rm), p(rv),
                                           p(y), p(mean), p(invstd), p(filled if fill else acc), I(fill), st()), "fwd")

        def bwd(s, mode):
            x, gy, y, gx, gr = T[s]
            L.check(lib.dsf_bn_backward_acc(p(x), p(gy), p(y) if mode == 1 else None, p(g), p(b), p(mean), p(invstd), I64(M), I(C), I(mode),
                                            p(gx), p(gr) if mode == 1 else None, p(gg), p(gb), p(acc), st()), "bwd")

        calls = [("forward", lambda s: fwd(s), 3), ("fwd+res", lambda s: fwd(s, True), 4), ("apply", lambda s: fwd(s, False, 1), 2),
                 ("apply+res", lambda s: fwd(s, True, 1), 3), ("backward2", lambda s: bwd(s, 2), 5), ("backward1", lambda s: bwd(s, 1), 8)]
        for name, fn, passes in calls:
            t1, tn = timed(fn, 1), timed(fn, nset)
            print("%6d x %-4d %4.0f %-9s %10.1f %8.2f %10.1f %8.2f" % (M, C, mb, name, t1, passes * mb / t1, tn, passes * mb / tn))


if __name__ == "__main__":
    main()
