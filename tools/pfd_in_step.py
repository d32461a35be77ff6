"""The point-to-mesh launches of a config's step, replayed ALONE on the step's own tensors (bench.py's workloads, --init fresh and
fitted): microseconds per launch of every mesh_point_fwd_kernel call of one step -- what the kernel costs in the state the bench
measures, without the other streams of the step beside it.   python tools/pfd_in_step.py [config ...]   (GPU box)"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dsf_amd import ops

configs = [int(a) for a in sys.argv[1:]] or [3, 5]
for cfg in configs:
    for init in ("fresh", "fitted"):
        args = argparse.Namespace(config=cfg, batch=0, backbone="", init=init)
        w = bench.build_workload(args, torch.device("cuda"), 0, 1)
        for _ in range(2):
            w["run"]()                                                  # the state after two optimizer steps, as the bench's warm-up
        calls = []
        orig = ops.MeshPointDistance.forward

        def spy(ctx, verts, points, faces_cat, part_first, seg, n_parts):
            calls.append((verts.detach().clone(), points.detach().clone(), faces_cat, part_first, None if seg is None else seg.clone(), n_parts))
            return orig(ctx, verts, points, faces_cat, part_first, seg, n_parts)
        ops.MeshPointDistance.forward = staticmethod(spy)
        try:
            w["run"]()
        finally:
            ops.MeshPointDistance.forward = staticmethod(orig)
        torch.cuda.synchronize()
        out = []
        for c in calls:
            fn = lambda: ops.MeshPointDistance.apply(*c)
            with torch.no_grad():
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record(); torch.cuda.synchronize()
            v = c[0]
            ext = (v.amax(1) - v.amin(1)).norm(dim=-1).mean().item()
            out.append("%s B=%d P=%d: %.0f us (mesh extent %.3g)" % ("labelled" if c[4] is not None else "whole mesh", v.shape[0], c[1].shape[1], e0.elapsed_time(e1) * 50, ext))
        print("config %d, --init %s: %d point-to-mesh launches per step, alone: %s" % (cfg, init, len(calls), "; ".join(out)))
        del w
        torch.cuda.empty_cache()
