"""Race check of the backward-weights side stream on the real config-2 step: twin networks, deterministic mode, one stepping with
the side stream and one without, N steps over rotating batches -> parameters, BatchNorm buffers and losses must be bitwise equal."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config
from dsf_amd import nn_conv, _lib as L
L.set_deterministic(True)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
def make():
    torch.manual_seed(0)
    net = MANO_OCR_stage('ResNet_stage_18', 21, True).cuda()
    render = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).cuda()
    return RenderSupervisedStep(net, render, Config)
a, b = make(), make()
tg = []
for s in range(3):
    p, c, cube = synthetic_batch(B, "cuda", seed=10 + s)
    tg.append(a.make_targets(p, c, cube))
bad = 0
for i in range(N):
    t = tg[i % 3]
    nn_conv.WRW_STREAM[0] = True
    la, _ = a(t)
    nn_conv.WRW_STREAM[0] = False
    lb, _ = b(t)
    if not torch.equal(la, lb):
        bad += 1
        if bad < 4: print("step", i, "loss differs", float(la), float(lb))
nn_conv.WRW_STREAM[0] = True
torch.cuda.synchronize()
pd = sum(int(not torch.equal(x, y)) for x, y in zip(a.net.state_dict().values(), b.net.state_dict().values()))
print("steps", N, "batch", B, "| steps whose loss differed:", bad, "| state tensors differing at the end:", pd, "of", len(a.net.state_dict()))
