import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dsf_amd import nn_conv
for kind in ("c1_fwd", "c1_fwd_bn"):
    for cap in ("512", "1024", "2048", "4096"):
        os.environ["DSF_C1_STATS_WGS"] = cap
        rec = (kind, 32, 128, 128, 1, 128, 128, 64, 5, 5, 1, 1, 2, 2)
        t = min(nn_conv.replay(rec, iters=20)[0] for _ in range(3))
        print(kind, cap, "%.1f us" % t, flush=True)
        if kind == "c1_fwd": break
