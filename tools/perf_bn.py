"""BatchNorm passes alone, host-free (a HIP-graph replay of N calls): microseconds and algorithmic TB/s per entry point, per tensor
size of configs 2 (B = 32) and 4 (B = 192), per streaming variant (DSF_BN_VAR bits: 1 nt loads, 2 nt stores, 4 prefetch) and
apply-grid cap.   python tools/perf_bn.py [--quick] [--vars 0 1 2 ...] [--caps 1024 2048]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd import _lib as L          # noqa: E402
from dsf_amd import nn_norm            # noqa: E402

I, I64, CF = ctypes.c_int, ctypes.c_int64, ctypes.c_float


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def replay_us(fn, n=10, reps=3):
    """fn() enqueues one call on the current stream; -> microseconds per call from a graph of n calls (best of reps replays)"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--vars", type=int, nargs="*", default=[0, 1, 2, 3, 4, 7])
    ap.add_argument("--caps", type=int, nargs="*", default=[1024])
    a = ap.parse_args()
    lib = L.lib()
    dev = torch.device("cuda", 0)
    cases = [(131072, 64), (32768, 128), (8192, 256), (131072, 256), (786432, 64), (196608, 512), (786432, 256)]
    if a.quick:
        cases = [(131072, 64), (786432, 256)]
    rows = nn_norm.acc_rows()
    print("M C op var cap us TB/s(algorithmic)")
    for M, C in cases:
        n = M * C
        x = torch.randn(M, C, device=dev)
        r = torch.randn(M, C, device=dev)
        ga, gb = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
        y = torch.empty(M, C, device=dev)
        gx, gr = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev)
        gsum = torch.empty(M, C, device=dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        gg, gbt = torch.empty(C, device=dev), torch.empty(C, device=dev)
        acc = torch.zeros(rows * 2 * C, device=dev, dtype=torch.float64)
        acc[:C] = 1.0; acc[C:2 * C] = float(M)                   # finite statistics for the folded prologue
        st = L.stream_ptr
        mb = n * 4 / 1e6

        def fwd_apply(res):
            return lambda: lib.dsf_bn_forward_acc(_p(x), _p(r if res else None), _p(gamma), _p(beta), I64(M), I(C), CF(1e-5), CF(0.1), I(1), None, None,
                                                  _p(y), _p(mean), _p(invstd), _p(acc), I(1), st())

        def fwd_full():
            return lambda: lib.dsf_bn_forward_acc(_p(x), None, _p(gamma), _p(beta), I64(M), I(C), CF(1e-5), CF(0.1), I(1), None, None,
                                                  _p(y), _p(mean), _p(invstd), _p(acc), I(0), st())

        def bwd(relu, res, pair):
            return lambda: lib.dsf_bn_backward_acc_pair(_p(x), _p(ga), _p(gb if pair else None), _p(y), _p(gamma), _p(beta), _p(mean), _p(invstd), I64(M), I(C),
                                                        I(relu), _p(gx), _p(gr if res else None), _p(gg), _p(gbt), I(0), _p(acc), st())

        def add_then_bwd():
            def f():
                torch.add(ga, gb, out=gsum)
                lib.dsf_bn_backward_acc(_p(x), _p(gsum), _p(y), _p(gamma), _p(beta), _p(mean), _p(invstd), I64(M), I(C), I(1), _p(gx), _p(gr), _p(gg), _p(gbt),
                                        _p(acc), st())
            return f

        ops = [("fwd_apply", fwd_apply(False), 2), ("fwd_apply_res", fwd_apply(True), 3), ("fwd_reduce+apply", fwd_full(), 3),
               ("bwd_relu2", bwd(2, False, False), 5), ("bwd_res_relu1", bwd(1, True, False), None), ("bwd_res_relu1_pair", bwd(1, True, True), None),
               ("add+bwd_res_relu1", add_then_bwd(), None)]
        for cap in a.caps:
            os.environ["DSF_BN_APPLY_WGS"] = str(cap)
            for var in a.vars:
                os.environ["DSF_BN_VAR"] = str(var)
                for name, fn, passes in ops:
                    for wg in ((1, 0) if "res_relu1" in name and not name.startswith("add") else (1,)):
                        os.environ["DSF_BN_WRITE_G"] = str(wg)
                        if passes is None:      # algorithmic passes over the activation of this call
                            pair = "pair" in name
                            p_ = ((3 + (1 if pair else 0) + 1) + 3) if wg else ((3 + (1 if pair else 0)) * 2 + 2)      # sums pass + apply pass
                            if name.startswith("add"):
                                p_ = 3 + 4 + 3
                        else:
                            p_ = passes
                        us = replay_us(fn)
                        print("%7d %5d %-20s var %d cap %5d wg %d  %8.1f us  %6.2f TB/s  (%d passes of %.0f MB)" %
                              (M, C, name, var, cap, wg, us, p_ * mb / us, p_, mb), flush=True)
        del x, r, ga, gb, y, gx, gr, gsum
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
