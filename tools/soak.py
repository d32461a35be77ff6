"""Soak run: many training steps of the bench configuration; prints loss, allocated / reserved device memory and step
time per interval (catches leaks in caches / pools and numerical blow-ups that short tests cannot)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = 'cuda'
torch.manual_seed(0)
net = MANO_OCR_stage('ResNet_stage_18', 21, True).to(dev)
render = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).to(dev)
step = RenderSupervisedStep(net, render, Config)
batches = []
for i in range(4):                                             # rotate over a few batches so the net keeps learning
    p, c, cube = synthetic_batch(32, dev, i)
    batches.append(step.make_targets(p, c, cube, seed=100 + i))
t0 = time.perf_counter()
for it in range(1, steps + 1):
    loss, terms = step(batches[it % 4])
    if it % 250 == 0 or it == 1:
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0; t0 = time.perf_counter()
        print(f"step {it:5d} loss {float(loss):9.4f} alloc {torch.cuda.memory_allocated()/2**20:8.1f} MiB reserved "
              f"{torch.cuda.memory_reserved()/2**20:8.1f} MiB  {dt*1e3/ (250 if it > 1 else 1):.2f} ms/step "
              f"finite={all(bool(torch.isfinite(v)) for v in terms.values())}", flush=True)
