"""Per-parameter gradient comparison of a trainer step on the HIP path against its CPU oracle composition
(diagnostic for tests/test_gpu_steps.py):  python tools/step_diff.py ResNet_stage_50 3 2"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_steps as T
from oracle import step_ref, nets
from dsf_amd.assets import build_synthetic_mano
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.train_step import PretrainStep, synthetic_batch, draws_to, Config
backbone, views, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
refine = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
render = Render("synthetic", "nyu", T.CAM, (640, 480)).cuda()
orender = step_ref.OracleRender(build_synthetic_mano(0))
net_cpu, net_gpu = T._twin_pair(MANO_OCR_stage, backbone, 21, refine, seed=5)
step = PretrainStep(net_gpu, render, None, Config, views=views)
p, _, cube = synthetic_batch(B, "cpu", seed=31)
d = step.draw(B, "cpu", torch.Generator().manual_seed(32), np.random.default_rng(33))
loss_c = step_ref.pretrain_loss(net_cpu, orender, None, p, cube, d, Config, views=views)
loss_c.backward()
loss_g, terms = step.loss(p.cuda(), cube.cuda(), draws_to(d, "cuda"))
loss_g.backward()
print("loss", float(loss_c), float(loss_g))
rows = []
for (n, pc), (_, pg) in zip(net_cpu.named_parameters(), net_gpu.named_parameters()):
    if pc.grad is None: continue
    a, b = pc.grad.double().flatten(), pg.grad.cpu().double().flatten()
    cos = float((a*b).sum() / (a.norm()*b.norm() + 1e-300))
    rows.append((cos, n, float(a.norm()), float(b.norm())))
num = sum(((pc.grad.double() - pg.grad.cpu().double()) ** 2).sum() for (_, pc), (_, pg) in zip(net_cpu.named_parameters(), net_gpu.named_parameters()) if pc.grad is not None)
den = sum((pc.grad.double() ** 2).sum() for _, pc in net_cpu.named_parameters() if pc.grad is not None)
dot = sum((pc.grad.double() * pg.grad.cpu().double()).sum() for (_, pc), (_, pg) in zip(net_cpu.named_parameters(), net_gpu.named_parameters()) if pc.grad is not None)
ng = sum((pg.grad.cpu().double() ** 2).sum() for (_, pc), (_, pg) in zip(net_cpu.named_parameters(), net_gpu.named_parameters()) if pc.grad is not None)
print("overall cos %.6f rel %.4f" % (float(dot / (den * ng) ** 0.5), float((num / den) ** 0.5)))
bad = [r for r in rows if r[0] < 0.999]
print("tensors below 0.999: %d of %d; first 12:" % (len(bad), len(rows)))
for r in bad[:400]:
    print("%.5f %-40s |ref| %.3e |got| %.3e" % r)
# bridge check: stage-1 MANO estimate -> render on both sides: do any crop pixels flip?
with torch.no_grad():
    pp, cc = (p.repeat_interleave(views, 0), cube.repeat_interleave(views, 0)) if views > 1 else (p, cube)
    s_c = step_ref.synth_pass(orender, None, pp, cc, d, True)
    s_g = step.synth(pp.cuda(), cc.cuda(), draws_to(d, "cuda"))
    print("synthetic input image: pixels differing > 1e-4:", int(((s_c["img"] - s_g["img"].cpu()).abs() > 1e-4).sum()))
    c0 = net_cpu.pre(s_c["img_t"]); mano_c = net_cpu._run_trunk(c0, '')[3]
    g0 = net_gpu.pre(s_c["img_t"].cuda()); mano_g = net_gpu._run_trunk(g0, '')[3]
    print("stage-1 mano params max diff", float((mano_c - mano_g.cpu()).abs().max()))
    ic = orender.render(mano_c, s_c["center"], s_c["cube"])[0]
    ig = render.render(mano_g, s_c["center"].cuda(), s_c["cube"].cuda())[0].cpu()
    ig2 = render.render(mano_c.cuda(), s_c["center"].cuda(), s_c["cube"].cuda())[0].cpu()
    print("bridge render: pixels differing > 1e-4 (own params):", int(((ic - ig).abs() > 1e-4).sum()), " fg/bg flips:", int(((ic < 0.99) != (ig < 0.99)).sum()),
          "| same params:", int(((ic - ig2).abs() > 1e-4).sum()))
