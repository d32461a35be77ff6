import sys, torch
sys.path.insert(0, '.')
from dsf_amd import ops
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config
dev = 'cuda'
torch.manual_seed(0)
net = MANO_OCR_stage('ResNet_stage_18', 21, True).to(dev)
render = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).to(dev)
step = RenderSupervisedStep(net, render, Config)
p, c, cube = synthetic_batch(32, dev, 0); tgt = step.make_targets(p, c, cube)
calls = []
orig = ops.RenderCropFunction.apply
def spy(verts, *a):
    calls.append((verts.detach().clone(), a))
    return orig(verts, *a)
ops.RenderCropFunction.apply = spy
for it in range(30):
    calls.clear()
    step(tgt)
    if it in (0, 3, 10, 29):
        for ci, (v, a) in enumerate(calls):
            fx, fy = 588.03, 587.07
            u = v[..., 0] * fx / v[..., 2] + 320; w = v[..., 1] * fy / v[..., 2] + 240
            ext = torch.stack([u.amax(1) - u.amin(1), w.amax(1) - w.amin(1)], -1)
            zneg = (v[..., 2] <= 0).float().mean()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): orig(v, *a)
            e1.record(); torch.cuda.synchronize()
            print(f"it{it} call{ci}: bbox px mean {ext.mean(0).tolist()} max {ext.amax(0).tolist()} z<=0 frac {float(zneg):.3f} zmean {float(v[...,2].mean()):.1f} kernel {e0.elapsed_time(e1)*100:.1f} us")
