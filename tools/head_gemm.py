"""The MANO regression head's GEMMs (nn.Linear(512 or 2048 -> 62) on B = 32 .. 192 rows): round 2's profile of the other configs
shows a hipBLASLt kernel (Cijk_Alik_Bjlk ... MT256x16x16) at 656 us per call.  Times forward / input-gradient / weight-gradient
under hipBLASLt and rocBLAS."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for lib in ("hipblaslt", "hipblas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print("cannot select", lib, e); continue
    for B, K in ((32, 512), (64, 512), (128, 512), (192, 2048), (64, 2048)):
        x = torch.randn(B, K, device="cuda"); w = torch.randn(62, K, device="cuda"); b = torch.randn(62, device="cuda"); gy = torch.randn(B, 62, device="cuda")
        print("%-9s B=%3d K=%4d  fwd %7.1f us  dX %7.1f us  dW %7.1f us" % (lib, B, K, t(lambda: torch.nn.functional.linear(x, w, b)),
              t(lambda: gy @ w), t(lambda: gy.t() @ x)))
