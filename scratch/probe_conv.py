import time, torch, os
torch.backends.cudnn.benchmark = False
dev='cuda'
def t(f, n=5):
    torch.cuda.synchronize(); t0=time.perf_counter(); f(); torch.cuda.synchronize(); first=time.perf_counter()-t0
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return first, (time.perf_counter()-t0)/n
import torch.nn.functional as F
for (cin,cout,k,s,H) in [(64,64,3,1,64),(1,64,5,1,128),(128,128,3,1,32),(256,256,3,1,16),(512,512,3,1,8),(64,128,1,2,64)]:
    x=torch.randn(32,cin,H,H,device=dev,requires_grad=True); w=torch.randn(cout,cin,k,k,device=dev,requires_grad=True)
    def fb():
        y=F.conv2d(x,w,stride=s,padding=k//2); y.sum().backward()
    first,avg=t(fb)
    fl=2*32*cout*cin*k*k*(H//s)**2*3
    print(f'conv {cin}->{cout} k{k} s{s} H{H}: first {first:.2f}s avg {avg*1e3:.3f} ms  {fl/avg/1e12:.1f} TF', flush=True)
# matmul
a=torch.randn(4096,4096,device=dev); b=torch.randn(4096,4096,device=dev)
first,avg=t(lambda: a@b)
print(f'matmul 4096^3 fp32 first {first:.2f}s avg {avg*1e3:.3f} ms {2*4096**3/avg/1e12:.1f} TF')
x=torch.randn(32,64,64,64,device=dev)
first,avg=t(lambda: F.unfold(x,3,padding=1))
print(f'unfold first {first:.2f}s avg {avg*1e3:.3f} ms')
ct=torch.nn.ConvTranspose2d(512,256,4,2,1,bias=False).to(dev); x=torch.randn(32,512,8,8,device=dev,requires_grad=True)
first,avg=t(lambda: ct(x).sum().backward())
print(f'convT first {first:.2f}s avg {avg*1e3:.3f} ms')
bn=torch.nn.BatchNorm2d(64).to(dev); x=torch.randn(32,64,64,64,device=dev,requires_grad=True)
first,avg=t(lambda: bn(x).sum().backward())
print(f'bn first {first:.2f}s avg {avg*1e3:.3f} ms')
mp=torch.nn.MaxPool2d(3,2,1); x=torch.randn(32,64,128,128,device=dev,requires_grad=True)
first,avg=t(lambda: mp(x).sum().backward())
print(f'maxpool first {first:.2f}s avg {avg*1e3:.3f} ms')
