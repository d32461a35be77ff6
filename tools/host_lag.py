"""Where the eager config-2 step waits for the HOST, measured without a profiler: events recorded around the stretches of a step in which
(almost) no kernel is enqueued -- between the end of backward() and the optimizer launch, around the optimizer, between the optimizer and
the first forward kernel of the next step.  The GPU time between two such events is the work enqueued between them plus the time the
GPU sat drained waiting for the next launch; e.query() at the moment the host records the next event tells whether the GPU had already
drained (host-bound right there).      python tools/host_lag.py [--steps 40]"""
import argparse, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsf_amd import nn_conv
from dsf_amd.train_step import _stat_pool
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
args = types.SimpleNamespace(config=2, batch=0, backbone="", graph=False, no_graph=False, cpu_steps=0)
dev = torch.device("cuda", 0)
w = bench.build_workload(args, dev, 0, 1)
step, tgt = w["step"], w["tgt"]
for _ in range(6):
    w["run"]()
torch.cuda.synchronize()
names = ["step start", "before loss()", "after loss() [forward enqueued]", "after backward()", "before opt.step()", "after opt.step()"]
recs = []
t0 = time.perf_counter()
for it in range(a.steps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in names]
    drained, host = [], []
    def mark(i):
        drained.append(ev[i - 1].query() if i else False)    # had the GPU already passed the previous mark when the host got here?
        host.append(time.perf_counter())
        ev[i].record()
    mark(0)
    step.opt.zero_grad(set_to_none=True)
    step.render.mano_layer.clear_cache()
    with _stat_pool(step, step.net):
        mark(1)
        loss, terms = step.loss(tgt)
        mark(2)
        with nn_conv.grad_pool(step._pool_floats, step._pool_dev, reducer=step.grad_sync):
            loss.backward()
        mark(3)
    mark(4)
    step.opt.step()
    mark(5)
    recs.append((ev, drained, host))
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / a.steps * 1e3
recs = recs[5:]
n = len(recs)
print("eager config 2, %d steps: %.3f ms per step (wall, with the six event records per step)" % (a.steps, wall))
print("%-34s %12s %12s %10s" % ("stretch", "GPU us", "host us", "drained"))
for i in range(1, len(names)):
    g = sorted(r[0][i - 1].elapsed_time(r[0][i]) * 1e3 for r in recs)[n // 2]
    h = sorted((r[2][i] - r[2][i - 1]) * 1e6 for r in recs)[n // 2]
    d = sum(1 for r in recs if r[1][i]) / n
    print("%-34s %12.1f %12.1f %9.0f%%" % (names[i - 1] + " ->", g, h, 100 * d))
g = sorted(recs[k][0][5].elapsed_time(recs[k + 1][0][0]) * 1e3 for k in range(n - 1))[(n - 1) // 2]
print("%-34s %12.1f" % ("after opt.step() -> next step start", g))
