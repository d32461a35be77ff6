"""Do captured device-to-device copies replay?  y.copy_(x) (contiguous: hipMemcpyAsync node), x changes between replays."""
import torch
for n in (1, 64, 1000, 74688, 1 << 20, (1 << 20) + 3):
    x = torch.zeros(n, device="cuda"); y = torch.full((n,), -1.0, device="cuda"); z = torch.zeros(n, device="cuda")
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y.copy_(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y.copy_(x)
        z.copy_(y); z.add_(1)
    out = []
    for k in range(4):
        x.fill_(float(k + 1))
        if k == 2: torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize(); out.append((y.min().item(), y.max().item(), z.min().item(), z.max().item()))
    print(n, out)
