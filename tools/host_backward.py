"""Host time inside the backward functions of the step (they run on autograd's device thread, which cProfile does not see):
every torch.autograd.Function subclass defined in dsf_amd gets its backward wrapped in a timer."""
import argparse, collections, importlib, inspect, os, pkgutil, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import dsf_amd
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
a = ap.parse_args()
args = types.SimpleNamespace(config=a.config, batch=0, backbone="", graph=False, no_graph=True, cpu_steps=0, init="fresh")
w = bench.build_workload(args, torch.device("cuda", 0), 0, 1)
acc = collections.defaultdict(lambda: [0, 0.0])
seen = set()
for name, mod in list(sys.modules.items()):
    if not name.startswith("dsf_amd") or mod is None:
        continue
    for cname, cls in inspect.getmembers(mod, inspect.isclass):
        if issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and cls not in seen and "backward" in cls.__dict__:
            seen.add(cls)
            orig = cls.__dict__["backward"].__func__ if isinstance(cls.__dict__["backward"], staticmethod) else cls.backward
            def make(orig, key):
                def timed(*args, **kw):
                    t0 = time.perf_counter()
                    try:
                        return orig(*args, **kw)
                    finally:
                        e = acc[key]; e[0] += 1; e[1] += time.perf_counter() - t0
                return timed
            cls.backward = staticmethod(make(orig, "%s.%s" % (name.split(".")[-1], cname)))
run = w["run"]
for _ in range(6):
    run()
torch.cuda.synchronize()
for v in acc.values():
    v[0] = 0; v[1] = 0.0
N = 10
for _ in range(N):
    run()
torch.cuda.synchronize()
tot = 0.0
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if n:
        print("%-44s %5.1f calls/step  %7.3f ms/step  (%.1f us each)" % (k, n / N, t * 1e3 / N, t * 1e6 / n))
        tot += t
print("backward functions of dsf_amd: %.2f ms of host time per step" % (tot * 1e3 / N))
