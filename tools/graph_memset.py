"""Does a hipMemsetAsync captured in a HIP graph replay?  (x = 0 by memset; x += 1 by a kernel) x 3 replays -> expect 1, 1, 1."""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
for n in (1, 64, 1000, 74688, 1 << 20, (1 << 20) + 3):
    x = torch.full((n,), 5.0, device="cuda")
    y = torch.zeros(n, device="cuda")
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        x.add_(1)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y.copy_(x)                                      # reader before the memset (must see the previous replay's x)
        rc = hip.hipMemsetAsync(x.data_ptr(), 0, n * 4, torch.cuda.current_stream().cuda_stream)
        x.add_(1)
    out = []
    for _ in range(3):
        g.replay(); torch.cuda.synchronize(); out.append((x.min().item(), x.max().item(), y.max().item()))
    print(n, "rc", rc, out)
