"""How much of the stage-2 fusion convolution's input is exact zeros, per kernel tile?  (fraction of 16-channel x 128-pixel forward
A tiles and of 16-pixel x 128-channel backward-weights chunks that are all zero)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config
torch.manual_seed(0)
net = MANO_OCR_stage('ResNet_stage_18', 21, True).cuda()
render = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).cuda()
step = RenderSupervisedStep(net, render, Config)
p, c, cube = synthetic_batch(32, 'cuda', 0); tgt = step.make_targets(p, c, cube)
for _ in range(30): step(tgt)                      # a somewhat trained net: hands inside the crop
got = {}
def grab(m, i, o):
    got.setdefault("x", i[0].detach())
h = net.fusion[0].register_forward_hook(grab)
step(tgt); h.remove()
x = got["x"].permute(0, 2, 3, 1).reshape(-1, got["x"].shape[1])          # [pixels][488]
M, C = x.shape
print("fusion input", tuple(got["x"].shape), "zero fraction overall %.3f" % float((x == 0).float().mean()))
pad = (-C) % 16
xc = torch.nn.functional.pad(x, (0, pad)).reshape(M, -1, 16)                # [M][chunks][16]
nz = (xc != 0).any(-1)                                                      # [M][chunks]
fw = nz.reshape(M // 128, 128, -1).any(1)                                   # forward A tiles: 128 pixels x 16 channels
print("forward: all-zero (128 px x 16 ch) tiles: %.3f of %d" % (1 - float(fw.float().mean()), fw.numel()))
per_chunk = 1 - fw.float().mean(0)
print("  by channel chunk:", [round(float(v), 2) for v in per_chunk])
ww = (x.reshape(M // 16, 16, C) != 0).any(1)                                # wrw: 16-pixel chunks, per channel
ww128 = torch.nn.functional.pad(ww, (0, (-C) % 128)).reshape(M // 16, -1, 128).any(-1)
print("backward-weights: all-zero (16 px x 128 k-row) chunks for one tap: %.3f" % (1 - float(ww128.float().mean())))
