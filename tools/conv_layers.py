"""Per-layer convolution table of one step of a bench config: every convolution launch of the step (nn_conv.RECORD) replayed alone
and timed, grouped by shape.   python tools/conv_layers.py [--config N] [--top 40]"""
import argparse, collections, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsf_amd import nn_conv

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--top", type=int, default=40)
a = ap.parse_args()
args = types.SimpleNamespace(config=a.config, batch=0, backbone="", graph=False, no_graph=True, cpu_steps=0, init="fresh")
w = bench.build_workload(args, torch.device("cuda", 0), 0, 1)
run = w["run"]
run(); run()
nn_conv.RECORD = []
run()
torch.cuda.synchronize()
recs, nn_conv.RECORD = nn_conv.RECORD, None
agg = collections.OrderedDict()
for r in recs:
    agg.setdefault(r, 0)
    agg[r] += 1
rows = []
for r, n in agg.items():
    us, fl, _ = nn_conv.replay(r, iters=5)
    rows.append((us * n, n, us, fl / us / 1e6, r))
rows.sort(reverse=True)
tot = sum(x[0] for x in rows)
print("config %d: total conv us/step %.0f over %d launches" % (a.config, tot, len(recs)))
for t, n, us, tf, r in rows[:a.top]:
    print(f"{t:8.0f}us n={n} each {us:7.1f}us {tf:6.1f}TF {nn_conv.kernel_name(r):28s} {r[0]} B{r[1]} in{r[2]}x{r[3]}x{r[4]} out{r[5]}x{r[6]}x{r[7]} k{r[8]} s{r[10]} d{r[11]}")
