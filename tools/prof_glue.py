"""Which torch ops (shapes) issue the elementwise / copy kernels of a step?  torch.profiler, grouped by op + input shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config
dev = 'cuda'
torch.manual_seed(0)
net = MANO_OCR_stage('ResNet_stage_18', 21, True).to(dev)
render = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).to(dev)
step = RenderSupervisedStep(net, render, Config)
p, c, cube = synthetic_batch(32, dev, 0); tgt = step.make_targets(p, c, cube)
for _ in range(4): step(tgt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3): step(tgt)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith("aten::") and e.device_time_total > 0 and any(k in e.key for k in ("add", "copy", "cat", "fill", "zero", "mul", "contiguous", "clone", "sum", "mean", "div", "where", "stack", "index")):
        rows.append((e.self_device_time_total / 3, e.count / 3, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
for r in rows[:45]:
    print("%8.1f us  x%5.1f  %-22s %s" % r)
