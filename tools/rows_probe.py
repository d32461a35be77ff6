import os, sys, types
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
args = types.SimpleNamespace(config=2, batch=0, backbone="", graph=False, no_graph=False, cpu_steps=0)
w = bench.build_workload(args, torch.device("cuda", 0), 0, 1)
step, tgt = w["step"], w["tgt"]
for _ in range(4):
    w["run"]()
seen, seq = {}, []
for it in range(24):
    step.forward_backward(tgt)
    rows = tuple(p.grad.data_ptr() for g in step.opt.param_groups for p in g["params"] if p.grad is not None)
    k = seen.setdefault(rows, len(seen))
    seq.append(k)
    step.opt.step()
print("distinct gradient-address sets over 24 steps:", len(seen), "sequence:", seq)
if len(seen) > 1:
    a, b = list(seen)[:2]
    d = [i for i, (x, y) in enumerate(zip(a, b)) if x != y]
    print("differing entries between the first two sets: %d of %d, first indices %s" % (len(d), len(a), d[:10]))
    ps = [p for g in step.opt.param_groups for p in g["params"] if p.grad is not None]
    for i in d[:6]:
        print("   ", i, tuple(ps[i].shape))
