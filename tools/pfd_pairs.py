"""How lane-coherent is the point-to-triangle scan?  A -DPFD_STATS build of csrc/pfd.hip (tools/platform/_war/libdsf_hip_pfdstats.so,
built by hand: the shipped library has none of this) counts, per mesh_point_fwd_kernel launch, the (point, triangle) pairs a lane
NEEDED evaluated and the lane slots its wave SPENT on evaluations (a triangle is evaluated when any lane of the wave needs it) -- the
ratio bounds what compacting the surviving pairs across lanes could save.  Launches = those of a config's own step (bench.py's
workload, after `steps` optimizer steps from --init fresh / fitted).   python tools/pfd_pairs.py [config] [steps]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsf_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "platform", "_war", "libdsf_hip_pfdstats.so")
import torch
import bench
from dsf_amd import ops
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lib = _lib.lib()
buf = (ctypes.c_ulonglong * 4)()
for init in ("fresh", "fitted"):
    args = argparse.Namespace(config=cfg, batch=0, backbone="", init=init)
    w = bench.build_workload(args, torch.device("cuda"), 0, 1)
    for _ in range(steps):
        w["run"]()
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
    for c in calls:
        with torch.no_grad():
            for _ in range(2):
                ops.MeshPointDistance.apply(*c)
            torch.cuda.synchronize(); lib.dsf_pfd_stats(buf)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.MeshPointDistance.apply(*c); e1.record(); torch.cuda.synchronize()
            lib.dsf_pfd_stats(buf)
        need, spent, dense_ev, cand_ev = [int(x) for x in buf]
        v = c[0]
        B, P = v.shape[0], c[1].shape[1]
        F = int(c[3][-1]) if c[4] is None else 0
        ext = (v.amax(1) - v.amin(1)).norm(dim=-1).mean().item()
        print("config %d %-6s %-10s B=%d P=%d: %.0f us; pairs needed %.1f M, lane slots spent %.1f M (x%.2f); triangle evaluations: %.2f M in dense "
              "blocks, %.2f M by candidate tests%s; mesh extent %.3g" % (
                  cfg, init, "labelled" if c[4] is not None else "whole mesh", B, P, e0.elapsed_time(e1) * 1e3, need / 1e6, spent / 1e6,
                  spent / max(need, 1), dense_ev / 1e6, cand_ev / 1e6, (" (all pairs: %.1f M)" % (B * P * F / 1e6)) if F else "", ext), flush=True)
    del w
    torch.cuda.empty_cache()
