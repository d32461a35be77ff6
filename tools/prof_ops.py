import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for _ in range(3): step(tgt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(tgt); torch.cuda.synchronize()

ka = prof.key_averages(group_by_input_shape=True)
rows = [e for e in ka if e.key in ('aten::copy_', 'aten::fill_', 'aten::add_', 'aten::add', 'aten::mul', 'aten::normal_', 'aten::cat', 'aten::clone', 'aten::contiguous', 'aten::mul_', 'aten::sub', 'aten::div', 'aten::where', 'aten::clamp_min', 'aten::relu', 'aten::relu_', 'aten::sum', 'aten::mean')]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:70]:
    print(f"{e.key:16s} n={e.count:4d} cuda={e.self_device_time_total:9.1f}us  {str(e.input_shapes)[:150]}")
