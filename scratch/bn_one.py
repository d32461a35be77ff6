import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.nn_norm import FusedBatchNorm2d
x = torch.randn(32, 64, 64, 64, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
fb = FusedBatchNorm2d(64).cuda()
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y = fb(x, None, True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    y.backward(x.detach())
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(i, f"fwd {1e3*(t1-t0):.2f} ms bwd {1e3*(t2-t1):.2f} ms", flush=True)
