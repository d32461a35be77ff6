"""Host time to ENQUEUE one step (no synchronisation inside the timed region) against the step's GPU time: how close the eager
step is to being host-bound.   python tools/enqueue_time.py [--config N]"""
import argparse, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
a = ap.parse_args()
args = types.SimpleNamespace(config=a.config, batch=0, backbone="", graph=False, no_graph=True, cpu_steps=0, init="fresh")
w = bench.build_workload(args, torch.device("cuda", 0), 0, 1)
run = w["run"]
for _ in range(6):
    run()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
t0 = time.perf_counter()
for _ in range(20):
    run()
torch.cuda.synchronize()
steady = (time.perf_counter() - t0) * 1e3 / 20
print("config %d: enqueue %.2f ms (min %.2f) of a lone step's %.2f ms; steady state %.2f ms per step" %
      (a.config, sorted(enq)[5], min(enq), sorted(tot)[5], steady))
