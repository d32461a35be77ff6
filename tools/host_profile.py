"""cProfile of the host side of the eager step (top functions by own time): where the enqueue time of tools/enqueue_time.py goes."""
import argparse, cProfile, os, pstats, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
args = types.SimpleNamespace(config=a.config, batch=0, backbone="", graph=False, no_graph=True, cpu_steps=0, init="fresh")
w = bench.build_workload(args, torch.device("cuda", 0), 0, 1)
run = w["run"]
for _ in range(6):
    run()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
    run()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
