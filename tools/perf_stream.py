"""Achievable streaming bandwidth on this GPU (torch copy / add / our BN kernels) for the activation sizes of the step."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.nn_norm import FusedBatchNorm2d
def bench(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for (B,C,H) in [(32,64,64),(32,256,64),(32,128,32),(32,256,16),(32,512,8)]:
    x=torch.randn(B,C,H,H,device='cuda').contiguous(memory_format=torch.channels_last)
    y=torch.empty_like(x); r=torch.randn_like(x)
    nb=x.numel()*4
    t=bench(lambda: y.copy_(x)); print(f"B{B} C{C} H{H} {nb/1e6:.1f}MB: copy {t:.1f}us {2*nb/t/1e6:.2f} TB/s", end=' | ')
    t=bench(lambda: torch.add(x,r,out=y)); print(f"add {t:.1f}us {3*nb/t/1e6:.2f} TB/s", end=' | ')
    bn=FusedBatchNorm2d(C).cuda()
    xg=x.clone().requires_grad_(True)
    t=bench(lambda: bn(xg, None, True)); print(f"bn fwd(3 kernels, 2R+1W) {t:.1f}us {3*nb/t/1e6:.2f} TB/s", end=' | ')
    out=bn(xg,None,True); g=torch.randn_like(out)
    t=bench(lambda: torch.autograd.grad(out,xg,g,retain_graph=True)); print(f"bn bwd(3 kernels, 4R+1W) {t:.1f}us {5*nb/t/1e6:.2f} TB/s")
