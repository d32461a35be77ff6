"""Two controls for the two-stage ResNet-50 gradient gap (round-3 verdict, weak #2): is the HIP path's 3.7-4.4 x larger
distance from the float64 gradient a property of the NETWORK (any other fp32 evaluation lands there too) or of the HIP path?

  python tools/r50_controls.py [frozen]        (the config-4 test case of tests/test_gpu_steps.py: B = 2 x 3 views)

(a) the torch twin (oracle/nets.py) moved to the SAME GPU on torch-ROCm's own convolution / BatchNorm / pooling kernels
    (MIOpen, and torch's native kernels with MIOpen switched off): two further fp32 evaluations with their own summation orders.
(b) teacher-forced backward: every block of the trunk (stem, each Bottleneck, the three transposed-convolution stages, the
    fusion layer, the heads) is run ALONE on the float64 run's own input (cast to fp32) and differentiated against the float64
    run's own upstream gradient (cast to fp32), on the HIP modules and on the torch-CPU fp32 modules; output, input gradient
    and parameter gradients are compared with float64's per block.  No error can travel from one block to the next, so the
    chaotic amplification through ~110 layers is gone and what is left is each block's own arithmetic.
"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import test_gpu_steps as T
from oracle import step_ref
from dsf_amd import nn_conv
from dsf_amd.assets import build_synthetic_mano
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.train_step import PretrainStep, synthetic_batch, draws_to, Config

frozen = "frozen" in sys.argv[1:]
backbone, views, B = os.environ.get("CTRL_BACKBONE", "ResNet_stage_50"), 3, 2


class NetOnGpu:
    """the torch.nn twin on the GPU (torch-ROCm kernels); geometry stays with the CPU oracle, as for _Net64"""

    def __init__(self, net):
        self.net = copy.deepcopy(net).cuda()
        self.refine = net.refine
        for q in self.net.parameters():
            q.grad = None

    def pre(self, x):
        return self.net.pre(x.cuda())

    def fusion(self, x):
        return self.net.fusion(x.cuda())

    def _run_trunk(self, x, suffix):
        c4, feat, pix, mano = self.net._run_trunk(x.cuda(), suffix)
        return c4, feat, pix.cpu(), mano.cpu()

    def named_parameters(self):
        return self.net.named_parameters()


def block_names(net):
    names = ["pre"]
    for suffix in ("", "_s2"):
        for l in range(1, 5):
            layer = getattr(net, "layer%d%s" % (l, suffix), None)
            if layer is not None:
                names += ["layer%d%s.%d" % (l, suffix, i) for i in range(len(layer))]
        for n in ("deconv_layer4", "deconv_layer3", "deconv_layer2", "mano_regress"):
            if hasattr(net, n + suffix):
                names.append(n + suffix)
    if hasattr(net, "fusion"):
        names.append("fusion")
    return names


def sub(net, name):
    m = net
    for part in name.split("."):
        m = m[int(part)] if part.isdigit() else getattr(m, part)
    return m


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
pin = lambda r: T._PinnedBridge(r, rec.images)

# ---- float64 truth, with every block's input / output / upstream gradient / input gradient recorded ----
net64 = T._Net64(net_cpu)
cap = {}
heads_cap = {}


def hook_block(name, mod):
    def fwd(m, inp, out):
        cap.setdefault(name, {})["x"] = inp[0].detach().clone()
        cap[name]["y"] = out.detach().clone()

    def bwd(m, gin, gout):
        cap[name]["gy"] = gout[0].detach().clone()
        cap[name]["gx"] = None if gin[0] is None else gin[0].detach().clone()

    mod.register_forward_hook(fwd)
    mod.register_full_backward_hook(bwd)


names = block_names(net64.net)
for n in names:
    hook_block(n, sub(net64.net, n))
for suffix in ("", "_s2"):
    for i, h in enumerate(getattr(net64.net, "finals" + suffix)):
        hook_block("finals%s.%d" % (suffix, i), h)
l64 = step_ref.pretrain_loss(net64, pin(orender), None, p, cube, d, Config, views=views)
l64.backward()
print("case: %s two-stage, %d x %d views%s; float64 loss %.6f" % (backbone, B, views, ", frozen statistics" if frozen else "", float(l64)))

# ---- (a) other fp32 evaluations of the whole step ----
print("\n(a) whole-step gradient against the float64 trunk (cosine, relative L2 error):")
print("  torch CPU fp32                          %.4f %.4f" % T._grad_error(net64, net_cpu))
lg, _ = PretrainStep(net_gpu, pin(render), None, Config, views=views, optimizer=step.opt).loss(p.cuda(), cube.cuda(), draws_to(d, "cuda"))
lg.backward()
print("  HIP path (dsf_amd)                      %.4f %.4f" % T._grad_error(net64, net_gpu))
for label, miopen in (("torch twin on the GPU, MIOpen           ", True), ("torch twin on the GPU, native (no MIOpen)", False)):
    torch.backends.cudnn.enabled = miopen
    try:
        tw = NetOnGpu(net_cpu)
        lt = step_ref.pretrain_loss(tw, pin(orender), None, p, cube, d, Config, views=views)
        lt.backward()
        torch.cuda.synchronize()
        print("  %s %.4f %.4f   (loss %.6f)" % ((label,) + T._grad_error(net64, tw) + (float(lt),)))
    except Exception as e:                                              # a missing MIOpen solver must not hide control (b)
        print("  %s failed: %s" % (label, str(e).splitlines()[0][:160]))
torch.backends.cudnn.enabled = True


# ---- (b) teacher-forced blocks ----
def rel(a, ref):
    ref = ref.double().cpu()
    return float((a.double().cpu() - ref).norm() / (ref.norm() + 1e-300))


def pgrads(mod):
    g = [q.grad.double().cpu().flatten() for q in mod.parameters() if q.grad is not None]
    return torch.cat(g) if g else torch.zeros(0, dtype=torch.float64)


def run_block(mod, x64, gy64, dev, call=None):
    for q in mod.parameters():
        q.grad = None
    x = x64.float().to(dev).requires_grad_(x64.is_floating_point())
    y = (call or mod)(x)
    y.backward(gy64.float().to(dev))
    if dev == "cuda":
        torch.cuda.synchronize()
    return y.detach(), (x.grad if x.grad is not None else None), pgrads(mod)


print("\n(b) teacher-forced blocks: relative L2 error against float64 of (output | input gradient | parameter gradients), torch-CPU fp32 then HIP:")
print("  %-22s %-30s %-30s %s" % ("block", "torch CPU fp32", "HIP", "ratio HIP/CPU (gx, gW)"))
worst = []
rows = [(n, sub(net_cpu, n), sub(net_gpu, n), None, None) for n in names]
for suffix in ("", "_s2"):
    hc, hg = getattr(net_cpu, "finals" + suffix), getattr(net_gpu, "finals" + suffix)
    a, b = cap["finals%s.0" % suffix], cap["finals%s.1" % suffix]
    cap["finals" + suffix] = {"x": a["x"], "y": torch.cat([a["y"], b["y"]], 1), "gy": torch.cat([a["gy"], b["gy"]], 1), "gx": a["gx"] + b["gx"]}
    rows.append(("finals" + suffix, hc, hg, (lambda x, h=hc: torch.cat([m(x) for m in h], 1)), (lambda x, h=hg: nn_conv.fused_heads(x, h))))
for n, mc, mg, call_c, call_g in rows:
    c = cap[n]
    p64 = pgrads(sub(net64.net, n))
    yc, gxc, gwc = run_block(mc, c["x"], c["gy"], "cpu", call_c)
    yg, gxg, gwg = run_block(mg, c["x"], c["gy"], "cuda", call_g)
    e = lambda y, gx, gw: (rel(y, c["y"]), rel(gx, c["gx"]) if (gx is not None and c["gx"] is not None) else float("nan"), rel(gw, p64) if p64.numel() else float("nan"))
    ec, eg = e(yc, gxc, gwc), e(yg, gxg, gwg)
    ratio = (eg[1] / ec[1] if ec[1] == ec[1] and ec[1] > 0 else float("nan"), eg[2] / ec[2] if ec[2] == ec[2] and ec[2] > 0 else float("nan"))
    worst.append((max(r for r in ratio if r == r) if any(r == r for r in ratio) else 0.0, n, ec, eg))
    print("  %-22s %.2e %.2e %.2e    %.2e %.2e %.2e    %.2f %.2f" % ((n,) + ec + eg + ratio))
print("\nworst blocks by HIP / CPU error ratio:")
for r, n, ec, eg in sorted(worst, reverse=True)[:6]:
    print("  %-22s ratio %.2f   cpu (%.2e %.2e %.2e)  hip (%.2e %.2e %.2e)" % ((n, r) + ec + eg))
