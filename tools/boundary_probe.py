"""Config 3: what the step boundary costs beside the replayed graph -- full step, replay + optimizer without the input copies,
replay alone (no optimizer: the weights stand still, same kernels)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsf_amd.train_step import GraphedStep
args = types.SimpleNamespace(config=3, batch=0, backbone="", graph=False, no_graph=False, cpu_steps=0)
w = bench.build_workload(args, torch.device("cuda", 0), 0, 1)
g = GraphedStep(w["step"], w["tgt"])


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


def replay_opt():
    g.graph.replay()
    g.step.opt.step()


for rep in range(2):
    print("full step %.3f ms | replay + optimizer (no input copies) %.3f | replay alone %.3f | optimizer alone %.3f"
          % (timed(lambda: g(w["tgt"])), timed(replay_opt), timed(g.graph.replay), timed(g.step.opt.step)))
