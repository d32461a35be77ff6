"""Which Python lines launch the torch 'glue' kernels of a step (strided adds, layout copies, cat, fills, scalar multiplies):
one eager step under torch.profiler with stacks, device kernels grouped by (kernel family, innermost dsf_amd / bench frame).
  python tools/glue_sources.py [--config N]"""
import argparse, collections, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
a = ap.parse_args()
args = types.SimpleNamespace(config=a.config, batch=0, backbone="", graph=False, no_graph=True, cpu_steps=0)
w = bench.build_workload(args, torch.device("cuda", 0), 0, 1)
for _ in range(3):
    w["run"]()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    w["run"]()
    torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fam = lambda n: ("strided add" if "manual_unroll" in n and "CUDAFunctor_add" in n else "layout copy" if "direct_copy" in n else
                 "cat" if "CatArray" in n else "fill" if "FillFunctor" in n else "mul" if "MulFunctor" in n else
                 "add" if "CUDAFunctor_add" in n else "other torch" if ("at::native" in n or "rocclr" in n) else None)
rows = collections.defaultdict(lambda: [0, 0.0, set()])
for e in prof.events():
    if not e.kernels:
        continue
    for k in e.kernels:
        f = fam(k.name)
        if f is None:
            continue
        frames = [s for s in (e.stack or []) if ("dsf_amd" in s or "bench.py" in s) and "site-packages" not in s]
        where = frames[0].replace(ROOT + "/", "") if frames else "(autograd engine: %s)" % e.name
        r = rows[(f, where, e.name)]
        r[0] += 1
        r[1] += k.duration
        if e.input_shapes:
            r[2].add(str(e.input_shapes)[:70])
print("config %d, one eager step: torch glue kernels by source" % a.config)
for (f, where, op), (n, us, shapes) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%7.1f us %3d x  %-12s %-22s %s   %s" % (us, n, f, op[:22], where[:90], ("; ".join(sorted(shapes))[:80] if shapes else "")))
