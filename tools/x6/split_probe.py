"""The whole-network weight split (x6_split_weights_multi_kernel, once per optimizer step) alone, on a job list of config 2's shape (30 M
weights, both directions).  History: a thread wrote its four consecutive 16-byte granules per plane directly -- one store instruction of
a wave touched 64 separate 64-byte segments -- 154.3 us = 3.9 TB/s; a probe build with lane-contiguous stores (wrong placement, same bytes)
ran 106.9 us; the shipped kernel now exchanges the granules through LDS so that every store instruction writes 1 KB: 132 us, and 124-128 us
with its registers held to four waves per SIMD (-DX6_SPLIT_OCC: 2 137.5, 3 136.0, 4 124-128, 5 119.3 with scratch spills, 6 174.8 us).
    python tools/x6/split_probe.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
cs = os.path.join(ROOT, "dsf_amd", "csrc")


def build(name, flags):
    so = os.path.join(ROOT, "tools", "x6", "_stamp", "libx6_%s.so" % name)
    if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(cs, "conv_x6.hip")):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
                               "-fno-fast-math", "-fno-slp-vectorize", "-fno-vectorize"] + flags +
                              ["-I" + cs, "-I" + os.path.join(ROOT, "include"), os.path.join(cs, "conv_x6.hip"), os.path.join(cs, "api.hip"), "-o", so],
                              stderr=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    lib.dsf_conv_x6_image_bytes.restype = ctypes.c_int64
    lib.dsf_conv_x6_image_granules.restype = ctypes.c_int64
    return lib


if __name__ == "__main__":
    I = ctypes.c_int
    layers = [(3, 3, 64, 64)] * 8 + [(3, 3, 128, 128)] * 8 + [(3, 3, 256, 256)] * 8 + [(3, 3, 512, 512)] * 6 + [(4, 4, 512, 256), (4, 4, 256, 256), (4, 4, 256, 256)] * 2 + [(3, 3, 488, 256)]
    for name, flags in (("split_ship", []), ("split_occ2", ["-DX6_SPLIT_OCC=2"]), ("split_occ3", ["-DX6_SPLIT_OCC=3"]), ("split_occ5", ["-DX6_SPLIT_OCC=5"]), ("split_occ6", ["-DX6_SPLIT_OCC=6"]), ("split_ship", [])):
        lib = build(name, flags)
        rows, keep, total = [], [], 0
        for (KH, KW, Ci, Co) in layers:
            w = torch.randn(KH, KW, Ci, Co, device="cuda")
            for mode in (0, 1):
                Ck, Cn = (Co, Ci) if mode else (Ci, Co)
                img = torch.empty(lib.dsf_conv_x6_image_bytes(I(KH), I(KW), I(Ck), I(Cn)) + 65536, device="cuda", dtype=torch.uint8)
                rows.append((w.data_ptr(), img.data_ptr(), KH, KW, Ci, Co, mode, total))
                keep.append((w, img))
                total += lib.dsf_conv_x6_image_granules(I(KH), I(KW), I(Ck), I(Cn))
        table = torch.tensor(rows + [(0, 0, 0, 0, 0, 0, 0, total)], dtype=torch.int64).cuda()
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        run = lambda: lib.dsf_conv_x6_split_weights_multi(ctypes.c_void_p(table.data_ptr()), I(len(rows)), ctypes.c_int64(total), st)
        for _ in range(3):
            assert run() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        nw = sum(k * l * a * b for k, l, a, b in layers)
        us = e0.elapsed_time(e1) * 1e3 / 20
        print("%-12s %d jobs, %.1f M weights, %.1f M granules: %.1f us  (%.2f TB/s of 4 B read + 6 B written per weight and direction)" % (
            name, len(rows), nw / 1e6, total / 1e6, us, 2 * nw * 10 / us / 1e6), flush=True)
