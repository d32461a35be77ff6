"""one-off: which tensor of the deterministic-mode step differs first between two runs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dsf_amd import _lib as L, ops
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config
render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
torch.manual_seed(0)
net = MANO_OCR_stage("ResNet_stage_18", 21, True).cuda()
with torch.no_grad():
    for head in (net.mano_regress[2], net.mano_regress_s2[2]):
        head.bias[58] = 1.0
step = RenderSupervisedStep(net, render, Config)
p, c, cube = synthetic_batch(8, "cuda", seed=2)
tgt = step.make_targets(p, c, cube)
L.set_deterministic(True)
log = []
orig_f, orig_b = ops.ManoPackedFunction.forward, ops.ManoPackedFunction.backward
def fwd(ctx, model, paras, k1, k2):
    out = orig_f(ctx, model, paras, k1, k2)
    log.append(("mano_fwd_in", paras.detach().clone())); log.append(("mano_fwd_verts", out[0].detach().clone())); log.append(("mano_fwd_joints", out[1].detach().clone()))
    return out
def bwd(ctx, gv, gj):
    out = orig_b(ctx, gv, gj)
    log.append(("mano_bwd_gv", None if gv is None else gv.detach().clone())); log.append(("mano_bwd_gj", None if gj is None else gj.detach().clone()))
    log.append(("mano_bwd_out", out[1].detach().clone()))
    return out
ops.ManoPackedFunction.forward = staticmethod(fwd)
ops.ManoPackedFunction.backward = staticmethod(bwd)
runs = []
for r in range(int(os.environ.get("RUNS", "3"))):
    log.clear()
    net.zero_grad(set_to_none=True)
    render.mano_layer.clear_cache()
    loss = step.loss(tgt)[0]
    loss.backward()
    torch.cuda.synchronize()
    runs.append((loss.detach().clone(), list(log), [(n, q.grad.clone()) for n, q in net.named_parameters() if q.grad is not None]))
for r in range(1, len(runs)):
    print("run", r, "vs 0: loss equal", bool(torch.equal(runs[0][0], runs[r][0])))
    for (n, a), (_, b) in zip(runs[0][1], runs[r][1]):
        eq = (a is None and b is None) or bool(torch.equal(a, b))
        print("   %-16s %s" % (n, "equal" if eq else "DIFFERS max %.3e" % float((a - b).abs().max())))
    bad = [n for (n, a), (_, b) in zip(runs[0][2], runs[r][2]) if not torch.equal(a, b)]
    print("   parameter gradients differing: %d of %d; last (= earliest in backward order) few: %s" % (len(bad), len(runs[0][2]), bad[-4:]))
