"""Which fp32 path is closer to the float64 gradients of an ill-conditioned two-stage net?  (diagnostic for
tests/test_gpu_steps.py::test_pretrain_and_config4_multiview_step_vs_oracle)
  python tools/step_truth.py ResNet_stage_50 3 2 [frozen]     [DSF_CONV_MATH=f32] [DSF_FUSED_BN=0] [DSF_DETERMINISTIC=1]
``frozen``: BatchNorm with frozen statistics (taken from one batch-statistics pass over the same images), the
well-conditioned variant of the case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_steps as T
from oracle import step_ref
from dsf_amd.assets import build_synthetic_mano
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.train_step import PretrainStep, synthetic_batch, draws_to, Config
backbone, views, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
frozen = len(sys.argv) > 4 and sys.argv[4] == "frozen"
# TRUTH_DET=fwd|wrw: the deterministic (unsplit / ordered) form for the forward-type convolutions only, or for backward-weights only
_det = os.environ.get("TRUTH_DET", "")
if _det:
    from dsf_amd import nn_conv, _lib as L
    def _wrap(name):
        inner = getattr(nn_conv, name)
        def f(*a, **k):
            old = L.set_deterministic(True)
            try:
                return inner(*a, **k)
            finally:
                L.set_deterministic(old)
        setattr(nn_conv, name, f)
    for nm in (("_fwd_x6", "_fwd", "_fwd_wt", "_bwd_data_s1") if _det == "fwd" else ("_wrw",)):
        _wrap(nm)
render = Render("synthetic", "nyu", T.CAM, (640, 480)).cuda()
orender = step_ref.OracleRender(build_synthetic_mano(0))
net_cpu, net_gpu = T._twin_pair(MANO_OCR_stage, backbone, 21, True, seed=5)
step = PretrainStep(net_gpu, render, None, Config, views=views)
p, _, cube = synthetic_batch(B, "cpu", seed=31)
d = step.draw(B, "cpu", torch.Generator().manual_seed(32), np.random.default_rng(33))
if frozen:
    T._freeze_statistics(net_cpu, net_gpu, orender, p, cube, d, views)
rec = T._Recording(orender)
step_ref.pretrain_loss(net_cpu, rec, None, p, cube, d, Config, views=views).backward()
pinned = PretrainStep(net_gpu, T._PinnedBridge(render, rec.images), None, Config, views=views, optimizer=step.opt)
lg, _ = pinned.loss(p.cuda(), cube.cuda(), draws_to(d, "cuda"))
lg.backward()
net64 = T._Net64(net_cpu)
l64 = step_ref.pretrain_loss(net64, T._PinnedBridge(orender, rec.images), None, p, cube, d, Config, views=views)
l64.backward()
print("loss64 %.6f lossgpu %.6f" % (float(l64), float(lg)))
# forward distance from the float64 trunk: stage-1 feature map and pixel heads, CPU fp32 and GPU
with torch.no_grad():
    pp, cc = (p.repeat_interleave(views, 0), cube.repeat_interleave(views, 0)) if views > 1 else (p, cube)
    s_c = step_ref.synth_pass(orender, None, pp, cc, d, True)
    f64 = net64._run_trunk(net64.pre(s_c["img_t"]), '')
    f32 = net_cpu._run_trunk(net_cpu.pre(s_c["img_t"]), '')
    fg = net_gpu._run_trunk(net_gpu.pre(s_c["img_t"].cuda()), '')
    rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())
    print("forward distance from float64 (feat, pix): cpu32 %.2e %.2e | gpu %.2e %.2e" % (rel(f32[1], f64[1]), rel(f32[2], f64[2]), rel(fg[1], f64[1]), rel(fg[2], f64[2])))
print("cpu32 vs 64 (cos, rel):", T._grad_error(net64, net_cpu))
print("gpu   vs 64 (cos, rel):", T._grad_error(net64, net_gpu))
rows = []
for (n, p64), (_, pc), (_, pg) in zip(net64.named_parameters(), net_cpu.named_parameters(), net_gpu.named_parameters()):
    if p64.grad is None: continue
    r = p64.grad.double().flatten(); nr = float(r.norm()) + 1e-300
    rows.append((n, float((pc.grad.double().flatten() - r).norm()) / nr, float((pg.grad.cpu().double().flatten() - r).norm()) / nr))
print("per tensor rel error (cpu32, gpu), every 12th + the 8 with the largest gpu/cpu ratio:")
for r in rows[::12]: print("  %-36s %.4f %.4f" % r)
for r in sorted(rows, key=lambda r: -r[2] / max(r[1], 1e-9))[:8]: print("  worst ratio %-30s %.4f %.4f" % r)
