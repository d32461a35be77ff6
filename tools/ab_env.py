"""Whole-step A/B of an environment switch the library reads per call (e.g. DSF_X6_PATCH), on ONE box and in ONE process:
blocks of steps alternate between the values, so clock / box differences cancel.

  python tools/ab_env.py --config 2 --var DSF_X6_PATCH --values 0 1 [--block 10] [--rounds 6]

Config 3 replays a HIP graph: one GraphedStep is captured per value."""
import argparse, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--var", required=True)
ap.add_argument("--values", nargs="+", required=True)
ap.add_argument("--block", type=int, default=10)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--init", default="fresh")
a = ap.parse_args()
def sync_python_switches():
    """switches that the Python side reads once at import (module-level lists): follow the environment"""
    from dsf_amd import nn_conv, nn_norm, ops
    for name, cell in (("DSF_BN_TWIN", nn_norm.TWIN), ("DSF_BN_EPILOGUE", nn_norm.EPILOGUE_STATS), ("DSF_DECODE_CL", ops.DECODE_CL), ("DSF_CAT", ops.CAT_FUSED), ("DSF_BN_POOL", nn_norm.POOL_FUSED),
                       ("DSF_C1_STATS", nn_conv.C1_STATS), ("DSF_C1_BN", nn_norm.C1_BN)):
        if name in os.environ:
            cell[0] = os.environ[name] == "1"
    from dsf_amd.model import backbone
    if "DSF_FT_S1_FORK" in os.environ:
        from dsf_amd import train_step
        train_step.S1_FORK[0] = os.environ["DSF_FT_S1_FORK"] == "1"
    if "DSF_FT_SYN_FORK" in os.environ:
        from dsf_amd import train_step
        train_step.SYN_FORK[0] = os.environ["DSF_FT_SYN_FORK"] == "1"
    if "DSF_FT_STREAMS" in os.environ:
        from dsf_amd import train_step
        train_step.FT_STREAMS[0] = os.environ["DSF_FT_STREAMS"] == "1"
    if "DSF_HEAD_FORK" in os.environ:
        backbone._HEAD_FORK[0] = os.environ["DSF_HEAD_FORK"] == "1"
    if "DSF_BRIDGE_ORDER" in os.environ:
        backbone._BRIDGE_ORDER[0] = os.environ["DSF_BRIDGE_ORDER"]


args = types.SimpleNamespace(config=a.config, batch=0, backbone="", graph=False, no_graph=False, cpu_steps=0, init=a.init)
dev = torch.device("cuda", 0)
os.environ[a.var] = a.values[0]
w = bench.build_workload(args, dev, 0, 1)
runs = {}
for v in a.values:
    os.environ[a.var] = v
    sync_python_switches()
    if a.config == 3:
        from dsf_amd.train_step import GraphedStep
        g = GraphedStep(w["step"], w["tgt"])
        runs[v] = (lambda g_: (lambda: g_(w["tgt"])))(g)
    else:
        runs[v] = w["run"]
    for _ in range(4):
        runs[v]()
torch.cuda.synchronize()
tot = {v: [] for v in a.values}
for r in range(a.rounds):
    for v in (a.values if r % 2 == 0 else a.values[::-1]):
        os.environ[a.var] = v
        sync_python_switches()
        runs[v]()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.block):
            runs[v]()
        torch.cuda.synchronize()
        tot[v].append((time.perf_counter() - t0) * 1e3 / a.block)
for v in a.values:
    xs = sorted(tot[v])
    print("AB config %d %s=%s: median %.3f ms/step  min %.3f  max %.3f  (%d blocks of %d)" %
          (a.config, a.var, v, xs[len(xs) // 2], xs[0], xs[-1], len(xs), a.block))
