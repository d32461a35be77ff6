"""How sensitive is the two-stage ResNet-50 gradient (config-4 test case, B = 2 x 3 views) to the SIZE of the forward rounding
error?  CPU only.  The torch fp32 twin is run with every block's output multiplied by (1 + s * N(0,1)) -- an fp32 evaluation whose
per-block rounding is s instead of its own ~2e-7 -- and its whole-parameter gradient is compared with the float64 trunk's, next to
the distance of its stage-1 feature map from float64's.  Reads: where on this curve do torch-CPU (1.8e-5), torch-GPU and the HIP
path (2.9e-5) sit?
  python tools/r50_noise.py [frozen]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import test_gpu_steps as T
from oracle import step_ref, nets
from dsf_amd.assets import build_synthetic_mano
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import synthetic_batch, draw_augmentation, Config

views, B = 3, 2
torch.manual_seed(5)
net = nets.build(MANO_OCR_stage, "ResNet_stage_50", 21, True)
with torch.no_grad():                                                   # as tests/test_gpu_steps.py::_twin_pair
    for name in ("mano_regress", "mano_regress_s2"):
        head = getattr(net, name)
        head[2].bias[58] = 1.0
        head[2].bias[3:48] = 0.2 * torch.randn(45)
        head[2].bias[:3] = torch.tensor([0.3, -0.2, 0.1])
orender = step_ref.OracleRender(build_synthetic_mano(0))
p, _, cube = synthetic_batch(B, "cpu", seed=31)
d = draw_augmentation(B, "cpu", torch.Generator().manual_seed(32), np.random.default_rng(33), views=views, depth_range=(500, 1200),
                      view_scale=1.0, mask=True)
rec = T._Recording(orender)
step_ref.pretrain_loss(net, rec, None, p, cube, d, Config, views=views).backward()
pin = lambda: T._PinnedBridge(orender, rec.images)
net64 = T._Net64(net)
step_ref.pretrain_loss(net64, pin(), None, p, cube, d, Config, views=views).backward()
with torch.no_grad():
    pp, cc = p.repeat_interleave(views, 0), cube.repeat_interleave(views, 0)
    img = step_ref.synth_pass(orender, None, pp, cc, d, True)["img_t"]
    f64 = net64._run_trunk(net64.pre(img), '')[1]


def blocks(n):
    out = [n.pre, n.fusion]
    for s in ("", "_s2"):
        for l in range(1, 5):
            out += list(getattr(n, "layer%d%s" % (l, s)))
        out += [getattr(n, "deconv_layer%d%s" % (k, s)) for k in (4, 3, 2)]
    return out


print("per-block relative noise s | stage-1 feature distance from float64 | gradient vs float64 (cosine, relative L2)")
print("  s = 0 (torch CPU fp32 itself)   %.2e   %.4f %.4f" % ((float((net._run_trunk(net.pre(img), '')[1].detach().double() - f64).norm() / f64.norm()),) + T._grad_error(net64, net)))
for s in (1e-7, 2e-7, 4e-7, 8e-7, 1.6e-6):
    for seed in (0, 1, 2):
        g = torch.Generator().manual_seed(seed)
        hooks = [m.register_forward_hook(lambda mod, i, o: o * (1 + s * torch.randn(o.shape, generator=g))) for m in blocks(net)]
        for q in net.parameters():
            q.grad = None
        step_ref.pretrain_loss(net, pin(), None, p, cube, d, Config, views=views).backward()
        with torch.no_grad():
            dist = float((net._run_trunk(net.pre(img), '')[1].double() - f64).norm() / f64.norm())
        for h in hooks:
            h.remove()
        print("  s = %.1e seed %d            %.2e   %.4f %.4f" % ((s, seed, dist) + T._grad_error(net64, net)), flush=True)
