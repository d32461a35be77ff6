import sys, os, torch
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
for _ in range(12): step(tgt)
torch.cuda.synchronize()
