"""Config 2: eager step against a HIP-graph replay of the same step, captured three ways, on ONE box in ONE process
(alternating blocks, so clocks / boxes cancel) -- round-5 verdict item "a graph replay is slower than eager".

  forked   : as the eager step runs today (branch streams + the weight-gradient stream inside the capture)
  wrw_only : one stream + the weight-gradient stream
  single   : one stream, weight gradients inline

  python tools/graph_replay_ab.py [--block 10] [--rounds 6] [--modes forked wrw_only single]
Prints one line per (mode, eager | replay) and the graph's node counts; `--trace-mode M` runs ONLY a few replays of mode M after
a marker kernel (for `rocprofv3 --kernel-trace`: tools/lanes.py on the trace shows how many kernels the replay keeps in flight)."""
import argparse
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

MODES = {"forked": (True, True), "wrw_only": (False, True), "single": (False, False)}


def set_mode(mode):
    from dsf_amd import streams, nn_conv
    branches, wrw = MODES[mode]
    streams.ENABLED[0] = branches
    nn_conv.WRW_STREAM[0] = wrw and nn_conv.SIDE_API


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--block", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--modes", nargs="+", default=["forked", "wrw_only", "single"])
    ap.add_argument("--trace-mode", default="")
    ap.add_argument("--config", type=int, default=2)
    a = ap.parse_args()
    from dsf_amd.train_step import GraphedStep
    args = types.SimpleNamespace(config=a.config, batch=0, backbone="", graph=False, no_graph=False, cpu_steps=0, init="fresh")
    dev = torch.device("cuda", 0)
    w = bench.build_workload(args, dev, 0, 1)
    step, tgt = w["step"], w["tgt"]
    for _ in range(3):
        step(tgt)
    torch.cuda.synchronize()
    runs = {}
    modes = [a.trace_mode] if a.trace_mode else a.modes
    for m in modes:
        set_mode(m)
        for _ in range(2):
            step(tgt)
        g = GraphedStep(step, tgt, validate=False)
        print("graph %-8s: %s nodes %s" % (m, sum(g.node_types.values()), dict(g.node_types)), flush=True)
        runs[(m, "eager")] = (m, lambda: step(tgt))
        runs[(m, "replay")] = (m, (lambda g_: (lambda: g_()))(g))
    if a.trace_mode:
        m = a.trace_mode
        set_mode(m)
        for kind in ("eager", "replay"):
            torch.cuda.synchronize()
            torch.zeros(1 << 20, device=dev).fill_(1.0)            # marker between the phases in the trace
            for _ in range(6):
                runs[(m, kind)][1]()
            torch.cuda.synchronize()
        return
    tot = {k: [] for k in runs}
    keys = list(runs)
    for r in range(a.rounds):
        for k in (keys if r % 2 == 0 else keys[::-1]):
            set_mode(runs[k][0])
            fn = runs[k][1]
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.block):
                fn()
            torch.cuda.synchronize()
            tot[k].append((time.perf_counter() - t0) * 1e3 / a.block)
    for k in keys:
        xs = sorted(tot[k])
        print("config %d %-8s %-6s: median %.3f ms/step  min %.3f  max %.3f" % (a.config, k[0], k[1], xs[len(xs) // 2], xs[0], xs[-1]), flush=True)
    # host side: time to ENQUEUE one eager step / one replay (no synchronisation inside)
    for k in keys:
        set_mode(runs[k][0])
        fn = runs[k][1]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print("enqueue %-8s %-6s: %.2f ms host" % (k[0], k[1], (t1 - t0) * 1e3), flush=True)


if __name__ == "__main__":
    main()
