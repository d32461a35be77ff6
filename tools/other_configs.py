"""Timing of the other BASELINE configs' per-GPU steps on one GPU (not bench lines; sanity / regression numbers)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.train_step import MeshLossStep, RenderSupervisedStep, FinetuneStageStep, synthetic_batch, Config
from dsf_amd import ops
render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
def timeit(step, tgt, n=10, w=4):
    for _ in range(w): step(tgt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step(tgt)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
# config 3: batch 64, hourglass 2-stack + MANO head, full mesh loss + collision
from dsf_amd.model.hourglass import PoseNetMANO
torch.manual_seed(0)
net = PoseNetMANO(2, 21).cuda()
step = MeshLossStep(net, render, Config)
p, c, cube = synthetic_batch(64, "cuda", seed=9)
tgt = step.make_targets(p, c, cube)
ms = timeit(step, tgt)
print(f"config 3 (B=64 hourglass-2 + meshLoss + collision): {ms:.1f} ms/step, {64/ms*1e3:.0f} img/s")
# config 4 per-GPU share: 64 samples x 3 augmentView renders = 192 images through ResNet-50 2-stage (Trainer.Pretrain)
import numpy as np
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import PretrainStep
torch.manual_seed(0)
net = MANO_OCR_stage('ResNet_stage_50', 21, True).cuda()
step4 = PretrainStep(net, render, None, Config, views=3)
g = torch.Generator(device="cuda").manual_seed(4)
d = step4.draw(64, "cuda", g, np.random.default_rng(4))
ms = timeit(lambda _: step4(p, cube, d), None, n=5, w=2)
print(f"config 4 share (64 samples x 3 views, ResNet-50 2-stage, Pretrain step): {ms:.1f} ms/step, {64/ms*1e3:.0f} samples/s = {192/ms*1e3:.0f} images/s")
del step4, net
# config 5 per-GPU share: B = 64 synthetic + 64 real through the whole FinetuneStage step (frozen transfer net included)
from dsf_amd.render_model.transfer import define_G
torch.manual_seed(1)
net = MANO_OCR_stage('ResNet_stage_18', 21, True).cuda()
with torch.no_grad():
    for head in (net.mano_regress[2], net.mano_regress_s2[2]):
        head.bias[58] = 1.0
gen = define_G(1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier').cuda()
step5 = FinetuneStageStep(net, render, gen, Config)
pr, cr, cube_r = synthetic_batch(64, "cuda", seed=22)
with torch.no_grad():
    img_r = render.render(pr, cr, cube_r)[0]
    _, M_r, _, _ = ops.crop_setup(cr, cube_r, render.cam, 128)
g = torch.Generator(device="cuda").manual_seed(5)
ms = timeit(lambda _: step5(p, cube, img_r, cr, cube_r, M_r, generator=g), None, n=6, w=3)
print(f"config 5 share (B=64 synthetic + 64 real, FinetuneStage step): {ms:.1f} ms/step, {64/ms*1e3:.0f} pairs/s")
del step5, net, gen
# evaluation path: eval-mode forward + decode + MANO joints + mean joint error, B = 32 and 128
from dsf_amd.eval_step import EvalStep
net = MANO_OCR_stage('ResNet_stage_18', 21, True).cuda()
ev = EvalStep(net, render, Config)
for Bv in (32, 128):
    p, c, cube = synthetic_batch(Bv, "cuda", seed=3)
    with torch.no_grad():
        img, juvd, jxyz, _ = render.render(p, c, cube)
        _, M, _, _ = ops.crop_setup(c, cube, render.cam, 128)
    batch = (img, jxyz[:, render.mano_layer.transfer], juvd, c, M, cube)
    ev.test([batch] * 3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev.test([batch] * 20)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"eval test_iter B={Bv} ResNet-18 2-stage: {ms:.2f} ms/batch, {Bv/ms*1e3:.0f} img/s")
