"""Stream-priority A/B of the weight-gradient side stream (verdict r2 item 5), interleaved rounds in ONE process:
  a  main = default stream, side priority 0 (shipped)        b  main = default stream, side = lowest priority the runtime offers
  c  main = a HIGH-priority stream, side priority 0          d  one stream (DSF_WRW_STREAM=0)
  e  main = default stream, side = a stream created with the runtime's LEAST priority (hipStreamCreateWithPriority)
Prints ms per step per variant (median and min over rounds) for the config-2 step at the given batch."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsf_amd import nn_conv
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rounds, steps = 6, 12
dev = torch.device("cuda")
torch.manual_seed(0)
net = MANO_OCR_stage("ResNet_stage_18", 21, True).to(dev)
render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).to(dev)
step = RenderSupervisedStep(net, render, Config)
p, c, cube = synthetic_batch(B, dev, seed=0)
tgt = step.make_targets(p, c, cube)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range (lowest, highest):", lo, hi)
high = torch.cuda.Stream(priority=hi)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
least, greatest = ctypes.c_int(0), ctypes.c_int(0)
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest))
print("hipDeviceGetStreamPriorityRange (least, greatest):", least.value, greatest.value)
raw = ctypes.c_void_p(0)
rc = hip.hipStreamCreateWithPriority(ctypes.byref(raw), ctypes.c_uint(1), ctypes.c_int(least.value))      # hipStreamNonBlocking
print("low-priority stream rc", rc)
low_ext = torch.cuda.ExternalStream(raw.value) if rc == 0 else None


def run(variant, n):
    nn_conv.join_side_streams(); torch.cuda.synchronize()
    nn_conv._SIDE.clear()
    if variant == "e" and low_ext is not None:
        nn_conv._SIDE[torch.device("cuda", torch.cuda.current_device())] = low_ext
    nn_conv.WRW_STREAM[0] = variant != "d"
    nn_conv.WRW_PRIORITY = lo if variant == "b" else 0
    ctx = torch.cuda.stream(high) if variant == "c" else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(6):                       # (a fresh side stream warms its own allocator pool)
            step(tgt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(tgt)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


V = "abcde" if low_ext is not None else "abcd"
res = {v: [] for v in V}
for r in range(rounds):
    for v in V:
        res[v].append(run(v, steps))
for v in V:
    print("variant %s: median %.3f ms  min %.3f ms  all %s" % (v, statistics.median(res[v]), min(res[v]), " ".join("%.2f" % x for x in res[v])))
