import time, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.nn_conv import Conv2dFunction, ConvTranspose2dFunction
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
B=32
for (ci,co,k,s,p,H) in [(64,64,3,1,1,64),(128,128,3,1,1,32),(256,256,3,1,1,16),(512,512,3,1,1,8),(64,128,3,2,1,64),(488,256,3,1,1,64),(256,64,3,1,1,64),(1,64,5,1,2,128),(256,84,1,1,0,64)]:
    x=torch.randn(B,ci,H,H,device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w=torch.randn(co,ci,k,k,device='cuda',requires_grad=True)
    Ho=(H+2*p-k)//s+1
    fl=2*B*Ho*Ho*co*ci*k*k
    with torch.no_grad():
        tf=bench(lambda: Conv2dFunction.apply(x,w,None,s,(p,p)))
    y=Conv2dFunction.apply(x,w,None,s,(p,p)); gy=torch.randn_like(y)
    tb=bench(lambda: torch.autograd.grad(y,[x,w],gy,retain_graph=True))
    print(f'conv {ci}->{co} k{k} s{s} H{H}: fwd {tf*1e3:.0f} us {fl/tf/1e9:.1f} TF | bwd(data+wrw) {tb*1e3:.0f} us {2*fl/tb/1e9:.1f} TF',flush=True)
x=torch.randn(B,512,8,8,device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
w=torch.randn(512,256,4,4,device='cuda',requires_grad=True)
with torch.no_grad(): tf=bench(lambda: ConvTranspose2dFunction.apply(x,w,None,2,(1,1),(0,0)))
print('convT 512->256 8->16 fwd us', tf*1e3)
