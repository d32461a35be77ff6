import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.optim import FusedAdamW
net = MANO_OCR_stage('ResNet_stage_18', 21, True).cuda()
net(torch.randn(2, 1, 128, 128, device='cuda')) if False else None
for p in net.parameters(): p.grad = torch.randn_like(p)
for name, opt in (("torch", torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=0.01)), ("fused", FusedAdamW(net.parameters(), lr=1e-3, weight_decay=0.01))):
    for _ in range(3): opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(20): opt.step()
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print(name, f"gpu {e0.elapsed_time(e1)/20*1e3:.0f} us/step, cpu issue {(t1-t0)/20*1e6:.0f} us/step", sum(p.numel() for p in net.parameters()))
