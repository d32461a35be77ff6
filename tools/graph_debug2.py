"""eager vs graph trajectories in deterministic mode; SYNC=1 drains the stream before each replay."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import RenderSupervisedStep, GraphedStep, synthetic_batch, Config
from dsf_amd import _lib as L
L.set_deterministic(True)
dev = 'cuda'
B = int(os.environ.get("B", "32")); SYNC = os.environ.get("SYNC", "0")
def make():
    torch.manual_seed(0)
    net = MANO_OCR_stage('ResNet_stage_18', 21, True).to(dev)
    render = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).to(dev)
    return RenderSupervisedStep(net, render, Config)
st = make()
p, c, cube = synthetic_batch(B, dev, 0)
tgt = st.make_targets(p, c, cube)
if os.environ.get("MODE", "graph") == "eager":
    out = [float(st(tgt)[0]) for _ in range(8)]
else:
    g = GraphedStep(st, tgt, warmup=2)
    out = ["w", "w"]
    for _ in range(6):
        if SYNC == "1": torch.cuda.synchronize()
        if SYNC == "2": torch.cuda.current_stream().synchronize()
        l, _ = g(tgt)
        if SYNC == "3": torch.cuda.synchronize()
        out.append(float(l))
print(os.environ.get("MODE", "graph"), "SYNC", SYNC, out)
