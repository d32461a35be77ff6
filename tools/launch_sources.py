"""Which Python line of this package issues which GPU launches: one step of a bench config under torch.profiler (with_stack),
every kernel attributed to the innermost frame inside dsf_amd/ (or bench.py) of the CPU op that launched it.

  python tools/launch_sources.py --config 5 [--top 60] [--filter elementwise,copy,fill,reduce]
Prints, per (source line, kernel family), the launches per step -- the list the launch-count work of round 6 was made from."""
import argparse
import collections
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def family(name):
    for key, fam in (("igemm", "conv"), ("conv_c1", "conv"), ("::bn_", "bn"), ("CUDAFunctor_add", "add"), ("FillFunctor", "fill"),
                     ("zero_kernel", "fill"), ("direct_copy", "copy"), ("copyBuffer", "copy"), ("CatArray", "cat"), ("MulFunctor", "mul"),
                     ("DivFunctor", "div"), ("reduce_kernel", "reduce"), ("arange", "arange"), ("rocsolver", "inverse"),
                     ("where_kernel", "where"), ("compare_scalar", "compare"), ("Cijk", "gemm"), ("x6_split", "split")):
        if key in name:
            return fam
    if "at::native" in name:
        return "torch:" + name.split("at::native::")[-1][:40]
    return "dsf:" + name.split("(")[0].split("::")[-1][:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=5)
    ap.add_argument("--top", type=int, default=80)
    ap.add_argument("--skip", default="conv,bn")
    a = ap.parse_args()
    args = types.SimpleNamespace(config=a.config, batch=0, backbone="", graph=False, no_graph=True, cpu_steps=0, init="fresh")
    w = bench.build_workload(args, torch.device("cuda", 0), 0, 1)
    run = w["run"]
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        run()
        torch.cuda.synchronize()
    ev = prof.events()
    # kernels carry no stack; their launching CPU op does: link by correlation through the profiler's own parent relation
    skip = set(a.skip.split(","))
    agg = collections.Counter()
    total = collections.Counter()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for e in ev:
        if e.device_type != torch.autograd.DeviceType.CPU:
            continue
        ks = [k for k in (e.kernels or [])]
        if not ks:
            continue
        where = "?"
        for fr in (e.stack or []):
            if ("dsf_amd/" in fr or "bench.py" in fr) and "site-packages" not in fr:
                where = fr.replace(root + "/", "").strip()
                break
        for k in ks:
            fam = family(k.name)
            total[fam] += 1
            if fam in skip:
                continue
            agg[(where[:110], fam, e.name[:28])] += 1
    print("config %d: launches per step by family: %s" % (a.config, dict(total.most_common())))
    print("%5s  %-12s %-28s %s" % ("n", "family", "op", "innermost dsf_amd frame"))
    for (where, fam, op), n in agg.most_common(a.top):
        print("%5d  %-12s %-28s %s" % (n, fam, op, where))


if __name__ == "__main__":
    main()
