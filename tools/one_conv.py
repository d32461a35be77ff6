import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd import nn_conv
recs = [("fwd", 32, 32, 32, 128, 32, 32, 128, 3, 3, 1, 1, 1, 1), ("fwd", 32, 16, 16, 256, 16, 16, 256, 3, 3, 1, 1, 1, 1),
        ("fwd", 32, 8, 8, 512, 8, 8, 512, 3, 3, 1, 1, 1, 1), ("fwd", 32, 64, 64, 64, 32, 32, 128, 3, 3, 2, 1, 1, 1),
        ("fwd", 32, 64, 64, 256, 64, 64, 84, 1, 1, 1, 1, 0, 0), ("bwd_s1", 32, 16, 16, 256, 16, 16, 256, 3, 3, 1, 1, 1, 1)]
for r in recs:
    us, fl, nb = nn_conv.replay(r, iters=20)
    print(r[0], r[4], r[7], r[5], r[10], f"{us:.1f} us {fl/us/1e6:.1f} TF")
