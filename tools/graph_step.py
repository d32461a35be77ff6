"""Eager vs HIP-graph replay of the config-2 step: same parameters after n steps? ms/step of each.  usage: graph_step.py [B]"""
import sys, os, time, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import RenderSupervisedStep, GraphedStep, synthetic_batch, Config
from dsf_amd import nn_conv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
from dsf_amd import _lib as L
if os.environ.get('DET', '1') == '1': L.set_deterministic(True)
dev = 'cuda'
def make():
    torch.manual_seed(0)
    net = MANO_OCR_stage('ResNet_stage_18', 21, True).to(dev)
    render = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).to(dev)
    return RenderSupervisedStep(net, render, Config)
a, b = make(), make()
p, c, cube = synthetic_batch(B, dev, 0)
tgt = a.make_targets(p, c, cube)
g = GraphedStep(b, tgt, warmup=2)
for _ in range(2): a(tgt)
for i in range(5):
    la, _ = a(tgt); lb, _ = g(tgt)
    gd = max((x.grad - y.grad).abs().max().item() for x, y in zip(a.net.parameters(), b.net.parameters()) if x.grad is not None)
    pd = max((x - y).abs().max().item() for x, y in zip(a.net.parameters(), b.net.parameters()))
    bd = max((x.float() - y.float()).abs().max().item() for x, y in zip(a.net.buffers(), b.net.buffers()))
    print(i, float(la), float(lb), "grad diff", gd, "param diff", pd, "buffer diff", bd)
worst = max(((x - y).abs().max() / x.abs().max().clamp_min(1e-12)).item() for x, y in zip(a.net.parameters(), b.net.parameters()))
print("max rel param diff after 7 steps:", worst)
for name, f in (("eager", lambda: a(tgt)), ("graph", lambda: g(tgt))):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(30): f()
    torch.cuda.synchronize(); print(name, "%.3f ms/step" % ((time.perf_counter() - t) / 30 * 1e3))
