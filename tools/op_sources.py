"""Which Python line of this package issues which torch operators in one step of a bench config: a TorchDispatchMode records every
aten op that reaches the GPU with the innermost frame inside dsf_amd/ (ops run by the autograd engine's own thread have no Python
frame: they are listed as <backward>).  Complements tools/launch_sources.py (kernel families per torch op; the ROCm profiler
delivers no Python stacks).   python tools/op_sources.py --config 5 [--top 80]"""
import argparse
import collections
import os
import sys
import traceback
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIEW_OPS = ("view", "reshape", "expand", "permute", "transpose", "slice", "select", "unsqueeze", "squeeze", "as_strided", "detach", "alias",
            "t.default", "unbind", "split", "_unsafe_view", "empty", "stride", "size", "numel", "is_", "sym_", "_local_scalar", "set_", "record_stream",
            "lift_fresh", "unfold", "narrow", "chunk", "_to_copy_meta", "new_empty", "empty_like", "empty_strided", "resize_", "storage_offset", "contiguous", "dim")


class Rec(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()

    def __torch_dispatch__(self, func, types_, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEW_OPS):
            where = "<backward>"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if (fr.filename.startswith(os.path.join(ROOT, "dsf_amd")) or fr.filename.endswith("bench.py")) and "op_sources" not in fr.filename \
                        and not fr.filename.endswith("_lib.py"):
                    where = "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, (fr.line or "").strip()[:90])
                    break
            self.n[(name.replace("aten.", ""), where)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=5)
    ap.add_argument("--top", type=int, default=100)
    a = ap.parse_args()
    args = types.SimpleNamespace(config=a.config, batch=0, backbone="", graph=False, no_graph=True, cpu_steps=0, init="fresh")
    w = bench.build_workload(args, torch.device("cuda", 0), 0, 1)
    run = w["run"]
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    rec = Rec()
    with rec:
        run()
    torch.cuda.synchronize()
    total = sum(rec.n.values())
    print("config %d: %d non-view aten ops in one step" % (a.config, total))
    by_op = collections.Counter()
    for (op, _), n in rec.n.items():
        by_op[op] += n
    print("by op:", dict(by_op.most_common(30)))
    for (op, where), n in rec.n.most_common(a.top):
        print("%4d  %-28s %s" % (n, op[:28], where))


if __name__ == "__main__":
    main()
