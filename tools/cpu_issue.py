"""Host-side issue time of one training step (how far the CPU is from being the bottleneck): time of the first steps
after a device sync, before any queue back-pressure."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config
dev = 'cuda'
torch.manual_seed(0)
net = MANO_OCR_stage('ResNet_stage_18', 21, True).to(dev)
render = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).to(dev)
step = RenderSupervisedStep(net, render, Config)
p, c, cube = synthetic_batch(32, dev, 0); tgt = step.make_targets(p, c, cube)
for _ in range(10): step(tgt)
for trial in range(3):
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for _ in range(3):
        step(tgt); ts.append(time.perf_counter())
    torch.cuda.synchronize(); t_end = time.perf_counter()
    print("issue ms per step:", [round((b - a) * 1e3, 2) for a, b in zip(ts, ts[1:])], "  3 steps incl. drain:", round((t_end - ts[0]) * 1e3, 2))
from dsf_amd.train_step import GraphedStep
g = GraphedStep(step, tgt)
for _ in range(3): g(tgt)
for trial in range(2):
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for _ in range(3):
        g(tgt); ts.append(time.perf_counter())
    torch.cuda.synchronize(); t_end = time.perf_counter()
    print("GraphedStep issue ms per step:", [round((b - a) * 1e3, 2) for a, b in zip(ts, ts[1:])], "  3 steps incl. drain:", round((t_end - ts[0]) * 1e3, 2),
          " graph nodes:", g.node_types)
