"""Which modules of a step receive a 4-D gradient that is NOT channels_last (the HIP kernels then convert it, and torch's own
adds / upsampling take their strided paths):  python tools/grad_layouts.py [--config N]
One eager step of bench.py's workload with a backward pre-hook on every module of the network."""
import argparse, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
a = ap.parse_args()
args = types.SimpleNamespace(config=a.config, batch=0, backbone="", graph=False, no_graph=True, cpu_steps=0)
w = bench.build_workload(args, torch.device("cuda", 0), 0, 1)
net = w["step"].net
found, seen = [], [0]


def hook(name):
    def f(mod, gout):
        for g in gout:
            if g is not None and g.dim() == 4:
                seen[0] += 1
                if g.shape[1] > 1 and g.shape[2] * g.shape[3] > 1 and not g.is_contiguous(memory_format=torch.channels_last):
                    found.append((name, type(mod).__name__, tuple(g.shape), tuple(g.stride())))
    return f


for n, m in net.named_modules():
    if n:
        m.register_full_backward_pre_hook(hook(n))
w["run"]()
torch.cuda.synchronize()
print("config %d: %d of %d 4-D module-output gradients are not channels_last" % (a.config, len(found), seen[0]))
for f in found:
    print("  %-40s %-22s %s strides %s" % f)
