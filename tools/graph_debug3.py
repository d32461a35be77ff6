import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import RenderSupervisedStep, GraphedStep, synthetic_batch, Config
from dsf_amd import _lib as L
L.set_deterministic(True)
dev = 'cuda'
torch.manual_seed(0)
net = MANO_OCR_stage('ResNet_stage_18', 21, True).to(dev)
render = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).to(dev)
st = RenderSupervisedStep(net, render, Config)
p, c, cube = synthetic_batch(32, dev, 0)
tgt = st.make_targets(p, c, cube)
g = GraphedStep(st, tgt, warmup=2)
params = list(net.parameters()); names = [n for n, _ in net.named_parameters()]
def sig():
    torch.cuda.synchronize()
    return [float(g.loss)] + [q.grad.double().sum().item() for q in params]
g.graph.replay(); ref = sig()
bad = 0
for i in range(30):
    if i % 2: torch.cuda.synchronize()
    g.graph.replay()
    s = sig()
    d = [n for n, a, b in zip(["loss"] + names, ref, s) if a != b and not (a != a and b != b)]
    if d:
        bad += 1
        if i < 4: print("replay", i, "differs in", len(d), "of", len(s), [(n, a, b) for n, a, b in zip(["loss"] + names, ref, s) if a != b][:3], [(n, a, b) for n, a, b in zip(["loss"] + names, ref, s) if a != b][-3:])
print("pure replays differing from the first:", bad, "of 30")
# now with the optimizer between, against an eager twin
torch.manual_seed(0)
net2 = MANO_OCR_stage('ResNet_stage_18', 21, True).to(dev)
