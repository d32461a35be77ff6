"""Bare training steps of one BASELINE config for kernel traces (no bench diagnostics, no roofline replays, no validation passes
in the traced region beyond the warm-up): `rocprofv3 --kernel-trace -- python3 tools/step_only.py [--config N] [--steps K]`.
The workload is bench.py's own (build_workload); config 3 replays its HIP graph as bench does.  Prints the number of TIMED steps
and their wall time; tools/kstats.py takes the last K steps' kernels by time stamp (--tail-ms)."""
import argparse, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--warmup", type=int, default=4)
ap.add_argument("--eager", action="store_true", help="config 3: eager instead of the HIP-graph replay")
a = ap.parse_args()
args = types.SimpleNamespace(config=a.config, batch=0, backbone="", graph=False, no_graph=a.eager, cpu_steps=0)
dev = torch.device("cuda", 0)
w = bench.build_workload(args, dev, 0, 1)
run = w["run"]
if a.config == 3 and not a.eager:
    from dsf_amd.train_step import GraphedStep
    g = GraphedStep(w["step"], w["tgt"])
    run = lambda: g(w["tgt"])
for _ in range(a.warmup):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    run()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("STEP_ONLY config %d steps %d wall_ms %.3f ms_per_step %.3f" % (a.config, a.steps, dt * 1e3, dt * 1e3 / a.steps))
