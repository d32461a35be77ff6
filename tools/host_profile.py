"""cProfile of the HOST side of the eager config-2 step (whole step, or --part opt for the optimizer call alone): where the Python time of
a step goes when the step is host-bound (tools/host_lag.py).   python tools/host_profile.py [--part step|fwd|opt] [--steps 20] [--top 35]"""
import argparse, cProfile, os, pstats, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--part", default="opt")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--top", type=int, default=35)
ap.add_argument("--sort", default="cumulative")
a = ap.parse_args()
args = types.SimpleNamespace(config=2, batch=0, backbone="", graph=False, no_graph=False, cpu_steps=0)
w = bench.build_workload(args, torch.device("cuda", 0), 0, 1)
step, tgt = w["step"], w["tgt"]
for _ in range(6):
    w["run"]()
torch.cuda.synchronize()
pr = cProfile.Profile()
for _ in range(a.steps):
    if a.part == "opt":
        step.forward_backward(tgt)
        torch.cuda.synchronize()                       # (so that nothing in the optimizer call waits for the GPU)
        pr.enable(); step.opt.step(); pr.disable()
    elif a.part == "fwd":                              # the forward pass alone (its graph is dropped unprofiled)
        from dsf_amd.train_step import _stat_pool
        step.opt.zero_grad(set_to_none=True)
        step.render.mano_layer.clear_cache()
        with _stat_pool(step, step.net):
            pr.enable(); loss, terms = step.loss(tgt); pr.disable()
        del loss, terms
        torch.cuda.synchronize()
    else:
        pr.enable(); step(tgt); pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.strip_dirs().sort_stats(a.sort).print_stats(a.top)
