"""Which torch-native (non-dsf) kernels a BASELINE config's step still launches, with the aten op and input shapes that issue
them (torch.profiler): python tools/torch_ops_by_config.py 5"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5


class A: pass


a = A(); a.config = cfg; a.batch = 0; a.backbone = ""
dev = torch.device("cuda", 0)
w = bench.build_workload(a, dev, 0, 1)
for _ in range(3):
    w["run"]()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    w["run"]()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
    self_dt = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if self_dt > 0:
        rows.append((self_dt, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("config %d: ops by self device time of ONE step (us, calls, op, shapes); total %.1f ms" % (cfg, tot / 1e3))
for r in rows[:45]:
    print("%9.1f %5d  %-44s %s" % r)
