"""Backward-weights launches of config 2's small-map layers: float-atomic epilogue (default) against per-split partial tiles +
ordered reduce (deterministic mode's path), microseconds per call INCLUDING the zero fill / reduce launch each needs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd import _lib as L, nn_conv  # noqa: E402

RECS = [("wrw", 32, 8, 8, 512, 8, 8, 512, 3, 3, 1, 1, 1, 1), ("wrw", 32, 16, 16, 256, 16, 16, 256, 3, 3, 1, 1, 1, 1),
        ("wrw", 32, 32, 32, 128, 32, 32, 128, 3, 3, 1, 1, 1, 1), ("wrw", 32, 16, 16, 256, 8, 8, 512, 3, 3, 2, 1, 1, 1),
        ("wrw", 32, 32, 32, 256, 16, 16, 256, 4, 4, 2, 1, 1, 1), ("wrw", 32, 64, 64, 256, 64, 64, 84, 1, 1, 1, 1, 0, 0),
        ("wrw", 32, 64, 64, 64, 64, 64, 64, 3, 3, 1, 1, 1, 1), ("wrw", 32, 64, 64, 488, 64, 64, 256, 3, 3, 1, 1, 1, 1)]
for r in RECS:
    out = []
    for det in (False, True):
        L.set_deterministic(det)
        us, fl, _ = nn_conv.replay(r, iters=30)
        out.append("%s %7.1f us %6.1f TF" % ("ordered" if det else "atomic ", us, fl / us / 1e6))
    L.set_deterministic(False)
    print("in %dx%dx%d out %dx%dx%d k%d s%d : %s" % (r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[10], "   ".join(out)))
