import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.nn_norm import FusedBatchNorm2d
import torch.nn.functional as F
def bench(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for (C,H) in [(64,64),(128,32),(256,16),(512,8),(256,64)]:
    x=torch.randn(32,C,H,H,device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r=torch.randn_like(x).requires_grad_(True)
    fb=FusedBatchNorm2d(C).cuda(); tb=torch.nn.BatchNorm2d(C).cuda()
    mb=x.numel()*4/1e6
    def ff():
        y=fb(x,r,True); y.backward(x.detach(), retain_graph=False)
    def tf():
        y=F.relu(tb(x)+r); y.backward(x.detach(), retain_graph=False)
    with torch.no_grad():
        f1=bench(lambda: fb(x,r,True)); t1=bench(lambda: F.relu(tb(x)+r))
    f2=bench(ff); t2=bench(tf)
    print(f'C{C} H{H} ({mb:.0f}MB): fused fwd {f1:.0f}us fwd+bwd {f2:.0f}us | torch fwd {t1:.0f}us fwd+bwd {t2:.0f}us')
