"""Does a kernel's packed-FP32 arithmetic survive beside bf16-MFMA workgroups on the same CU?  (round 4: the 256-thread MANO
backward, whose skinning loop the compiler had vectorised into v_pk_fma_f32, returned wrong first components in lanes 48-63
whenever conv_x6 workgroups ran on a second stream; alone, or beside fp32-MFMA / rocBLAS / elementwise kernels, never.)
  hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/platform/pk_fp32_beside_mfma.hip -o /tmp/libpk_victim.so
  python tools/platform/pk_fp32_beside_mfma.py /tmp/libpk_victim.so
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dsf_amd import nn_conv
lib = ctypes.CDLL(sys.argv[1])
blocks, iters = 512, 20000
out = torch.empty(blocks * 256 * 4, device="cuda")
x = torch.randn(32, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
gy = torch.randn(32, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
side = torch.cuda.Stream()
def run():
    assert lib.pk_victim_launch(ctypes.c_void_p(out.data_ptr()), blocks, iters, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    return out.clone().view(-1, 4)
ref = run()
print("alone: packed chain == plain chain: %s" % bool(torch.equal(ref[:, :2], ref[:, 2:])))
for load in ("none", "x6_wrw"):
    bad_pk = bad_plain = 0
    lanes = set()
    for it in range(50):
        if load == "x6_wrw":
            with torch.cuda.stream(side):
                for _ in range(3):
                    nn_conv._wrw(x, gy, 3, 3, 1, (1, 1))
        got = run()
        d_pk = (got[:, :2] != ref[:, :2]).any(1)
        d_pl = (got[:, 2:] != ref[:, 2:]).any(1)
        bad_pk += int(d_pk.any()); bad_plain += int(d_pl.any())
        lanes |= set((d_pk.nonzero().flatten() % 64).tolist())
    print("side load %-7s: launches with a wrong packed result %d / 50, with a wrong plain result %d / 50; lanes hit: %s" % (load, bad_pk, bad_plain, sorted(lanes)))
