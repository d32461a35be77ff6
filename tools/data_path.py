"""Depth data path: frames/s of dsf_depth_crop_normalize on the device vs the numpy oracle on one host core."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden_data as mgd
from oracle import data_ref
from dsf_amd import ops

depth, com, cube = mgd.frames(np.random.RandomState(5), 32)
d = torch.tensor(depth).cuda()
for B in (32, 256):
    dd = d.repeat(B // 32, 1, 1); cc = np.tile(com, (B // 32, 1)); cb = np.tile(cube, (B // 32, 1))
    cc_t = torch.tensor(cc, device="cuda"); cb_t = torch.tensor(cb, device="cuda")
    run = lambda: ops.depth_crop_normalize(dd, cc_t, cb_t, mgd.PARAS, 128)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"device: B={B} {us:.1f} us per batch = {B / us * 1e6:.0f} frames/s (output {B * 128 * 128 * 4 / us / 1e3:.1f} GB/s)")
t0 = time.perf_counter()
for i in range(32):
    data_ref.crop_and_normalize(depth[i], com[i], cube[i], (128, 128), mgd.PARAS)
dt = time.perf_counter() - t0
print(f"numpy oracle, 1 core: {32 / dt:.0f} frames/s")
