"""Soak run: many training steps of the bench configuration; prints loss, allocated / reserved device memory and step
time per interval (catches leaks in caches / pools and numerical blow-ups that short tests cannot)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = 'cuda'
torch.manual_seed(0)
net = MANO_OCR_stage('ResNet_stage_18', 21, True).to(dev)
render = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).to(dev)
step = RenderSupervisedStep(net, render, Config)
batches = []
for i in range(4):                                             # rotate over a few batches so the net keeps learning
    p, c, cube = synthetic_batch(32, dev, i)
    batches.append(step.make_targets(p, c, cube, seed=100 + i))
t0 = time.perf_counter()
for it in range(1, steps + 1):
    loss, terms = step(batches[it % 4])
    if it % 250 == 0 or it == 1:
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0; t0 = time.perf_counter()
        print(f"step {it:5d} loss {float(loss):9.4f} alloc {torch.cuda.memory_allocated()/2**20:8.1f} MiB reserved "
              f"{torch.cuda.memory_reserved()/2**20:8.1f} MiB  {dt*1e3/ (250 if it > 1 else 1):.2f} ms/step "
              f"finite={all(bool(torch.isfinite(v)) for v in terms.values())}", flush=True)

# ---- config-5 style step (FinetuneStage with the frozen generator), B = 16, and config-3 step (MeshLossStep), B = 64 ----
if len(sys.argv) > 2 and sys.argv[2] == "all":
    from dsf_amd.render_model.transfer import define_G
    from dsf_amd.train_step import FinetuneStageStep, MeshLossStep
    from dsf_amd.model.hourglass import PoseNetMANO
    from dsf_amd import ops
    torch.manual_seed(1)
    net = MANO_OCR_stage("ResNet_stage_18", 21, True).cuda()
    with torch.no_grad():
        for head in (net.mano_regress[2], net.mano_regress_s2[2]):
            head.bias[58] = 1.0
    gen = define_G(1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier').cuda()
    fstep = FinetuneStageStep(net, render, gen, Config)
    B = 16
    p, c, cube = synthetic_batch(B, "cuda", seed=21)
    pr, cr, cube_r = synthetic_batch(B, "cuda", seed=22)
    with torch.no_grad():
        img_r = render.render(pr, cr, cube_r)[0]
        _, M_r, _, _ = ops.crop_setup(cr, cube_r, render.cam, 128)
    g = torch.Generator(device="cuda").manual_seed(5)
    t0 = time.perf_counter()
    for it in range(1, 301):
        loss, terms = fstep(p, cube, img_r, cr, cube_r, M_r, generator=g)
        if it % 100 == 0:
            torch.cuda.synchronize(); dt = time.perf_counter() - t0; t0 = time.perf_counter()
            print(f"finetune step {it:4d} loss {float(loss):9.4f} alloc {torch.cuda.memory_allocated()/2**20:8.1f} MiB {dt*10:.2f} ms/step "
                  f"finite={all(bool(torch.isfinite(v)) for v in terms.values())}", flush=True)
    mnet = PoseNetMANO(2, 21).cuda()
    mstep = MeshLossStep(mnet, render, Config)
    p, c, cube = synthetic_batch(64, "cuda", seed=9)
    tgt = mstep.make_targets(p, c, cube)
    t0 = time.perf_counter()
    for it in range(1, 301):
        loss, terms = mstep(tgt)
        if it % 100 == 0:
            torch.cuda.synchronize(); dt = time.perf_counter() - t0; t0 = time.perf_counter()
            print(f"meshloss step {it:4d} loss {float(loss):9.4f} alloc {torch.cuda.memory_allocated()/2**20:8.1f} MiB {dt*10:.2f} ms/step "
                  f"finite={all(bool(torch.isfinite(v)) for v in terms.values())}", flush=True)
