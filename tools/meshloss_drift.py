"""Why does the EAGER config-3 step get slower as it runs (tools/soak.py: 22.1 -> 24.5 -> 26.8 ms per step over 300 steps, kernel times
unchanged)?  Host issue time per step, sizes of the module-level containers, and what tracemalloc sees growing."""
import gc, os, sys, time, tracemalloc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsf_amd import nn_conv, nn_norm
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.hourglass import PoseNetMANO
from dsf_amd.train_step import MeshLossStep, synthetic_batch, Config
render = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).cuda()
torch.manual_seed(1)
mnet = PoseNetMANO(2, 21).cuda()
mstep = MeshLossStep(mnet, render, Config)
p, c, cube = synthetic_batch(64, "cuda", seed=9)
tgt = mstep.make_targets(p, c, cube)
for _ in range(30):
    mstep(tgt)
torch.cuda.synchronize()
TRACE = os.environ.get('TRACE', '0') == '1'
if TRACE:
    tracemalloc.start(10)
    snap0 = tracemalloc.take_snapshot()
n0 = len(gc.get_objects())
for blk in range(6):
    host = 0.0
    t0 = time.perf_counter()
    for it in range(100):
        h0 = time.perf_counter()
        mstep(tgt)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("steps %3d-%3d: %.2f ms/step wall, %.2f ms/step host issue | gc objects +%d | _HELD %d | modules of torch.autograd.graph hooks: n/a" % (blk * 100, blk * 100 + 99, dt * 10, host * 10, len(gc.get_objects()) - n0, len(nn_conv._HELD)), torch.cuda.memory_reserved() >> 20, 'MiB reserved', flush=True)
if not TRACE:
    raise SystemExit(0)
snap1 = tracemalloc.take_snapshot()
print("tracemalloc growth (top 8):")
for st in snap1.compare_to(snap0, "traceback")[:8]:
    print("  +%.1f KiB in %d blocks" % (st.size_diff / 1024, st.count_diff))
    for line in st.traceback.format()[-6:]:
        print("      " + line.strip()[:150])
