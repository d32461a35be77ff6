"""Where a 50-70 us convolution launch's time goes, per WORKGROUP: a diagnostic build of dsf_amd/csrc/conv_x6.hip (-DX6_STAMP: every
workgroup of igemm_wrw_x6_kernel / igemm_x6p_kernel stamps s_memrealtime at its start, after its prologue, after its main loop and
after its epilogue, plus HW_ID / XCC_ID; the shipped library contains none of this) run on the small-map layers of config 2.
Answers: how long after the first workgroup does the last one START, how long does each one live, how many share a CU, and what
share of the launch's span x resident slots is actually occupied.
    python tools/x6/wrw_stamps.py        (GPU box; uses tools/x6/_stamp/libx6_stamp.so when it travelled with the snapshot, else builds it)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="stamp", help="stamp | ko_load | ko_addr | ko_split | ko_addr_split (knock-outs of igemm_wrw_x6_kernel's loop: "
                "X6_KO_LOAD no buffer loads, X6_KO_ADDR neither loads nor their address arithmetic, X6_KO_SPLIT tiles stored unsplit)")
ap.add_argument("--wgs", default=None, help="DSF_X6_WRW_WGS: workgroup target of the pixel split")
ap.add_argument("--wrw-only", action="store_true")
ap.add_argument("--fwd-only", action="store_true")
args = ap.parse_args()
VARIANTS = {"stamp": [], "ko_load": ["-DX6_KO_LOAD=1"], "ko_addr": ["-DX6_KO_ADDR=1"], "ko_split": ["-DX6_KO_SPLIT=1"],
            "ko_addr_split": ["-DX6_KO_ADDR=1", "-DX6_KO_SPLIT=1"],
            # knock-outs of igemm_x6p_kernel's loop: patch loads / weight-fragment loads issued out of range after the first chunk (the
            # instruction stays, the memory access does not), the patch stored unsplit
            "kop_a": ["-DX6P_KO_A=1"], "kop_b": ["-DX6P_KO_B=1"], "kop_ab": ["-DX6P_KO_A=1", "-DX6P_KO_B=1"], "kop_split": ["-DX6P_KO_SPLIT=1"]}
so = os.path.join(ROOT, "tools", "x6", "_stamp", "libx6_%s.so" % args.variant)
cs = os.path.join(ROOT, "dsf_amd", "csrc")
if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(cs, "conv_x6.hip")):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
                           "-fno-fast-math", "-fno-slp-vectorize", "-fno-vectorize", "-DX6_STAMP"] + VARIANTS[args.variant] +
                          ["-I" + cs, "-I" + os.path.join(ROOT, "include"), os.path.join(cs, "conv_x6.hip"), os.path.join(cs, "api.hip"), "-o", so],
                          stderr=subprocess.DEVNULL)
lib = ctypes.CDLL(so)
lib.dsf_conv_x6_image_bytes.restype = ctypes.c_int64
I = ctypes.c_int
P = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
STREAM = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ROWS = 16384
st = torch.zeros(ROWS, 8, device="cuda", dtype=torch.int64)
assert lib.dsf_x6_stamp_buffer(P(st)) == 0
if args.wgs:
    os.environ["DSF_X6_WRW_WGS"] = args.wgs


def nhwc(B, C, H):
    return torch.randn(B, H, H, C, device="cuda")


def report(tag, flops):
    torch.cuda.synchronize()
    s = st.cpu().numpy()
    s = s[s[:, 0] != 0]
    n = len(s)
    t = s[:, :4].astype(np.float64) * 0.01                 # us (100 MHz)
    t0 = t[:, 0].min()
    span = t[:, 3].max() - t0
    start, dur = t[:, 0] - t0, t[:, 3] - t[:, 0]
    hw, xcc = s[:, 6], s[:, 7] & 0xF
    cu = (xcc << 16) | (hw & 0xFF00)                       # (XCC, SE / SH / CU id)
    ids, counts = np.unique(cu, return_counts=True)
    # concurrency: for every CU the time during which >= 1 / >= 2 of its workgroups are alive
    one = two = 0.0
    for c in ids:
        sel = cu == c
        ev = sorted([(a, 1) for a in t[sel, 0]] + [(b, -1) for b in t[sel, 3]])
        live, last = 0, None
        for x, d in ev:
            if last is not None:
                if live >= 1: one += x - last
                if live >= 2: two += x - last
            live += d; last = x
    print("%s: %d workgroups on %d CUs (per CU: %s), launch span %.1f us -> %.1f TF" % (
        tag, n, len(ids), " ".join("%dx%d" % (k, (counts == k).sum()) for k in sorted(set(counts))), span, flops / span / 1e6))
    print("   start after the first workgroup: median %.1f, 90 %% %.1f, max %.1f us;  lifetime: min %.1f median %.1f max %.1f us"
          % (np.median(start), np.percentile(start, 90), start.max(), dur.min(), np.median(dur), dur.max()))
    print("   phases (median us): prologue %.2f  main loop %.1f  epilogue %.2f;   end of the first workgroup %.1f us, of the last %.1f us after the start"
          % (np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]), np.median(t[:, 3] - t[:, 2]), (t[:, 3] - t0).min(), span))
    print("   CU time with >= 1 workgroup alive: %.2f of span x 256;  with >= 2: %.2f;  workgroup-time / (span x 512 slots) = %.2f"
          % (one / (span * 256), two / (span * 256), dur.sum() / (span * 512)), flush=True)


def run(tag, fn, flops):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st.zero_(); torch.cuda.synchronize()
    fn()
    report(tag, flops)


B = 32
print("backward-weights, igemm_wrw_x6_kernel (variant %s, DSF_X6_WRW_WGS=%s)" % (args.variant, os.environ.get("DSF_X6_WRW_WGS", "512")))
for (Ci, Co, H, K, s, p) in ([] if args.fwd_only else [(128, 128, 32, 3, 1, 1), (256, 256, 16, 3, 1, 1), (512, 512, 8, 3, 1, 1), (256, 256, 64, 4, 2, 1)]):
    Ho = (H + 2 * p - K) // s + 1
    x, gy = nhwc(B, Ci, H), nhwc(B, Co, Ho)
    dw = torch.zeros(K, K, Ci, Co, device="cuda")
    def f():
        rc = lib.dsf_conv_x6_wrw(P(x), P(gy), P(dw), I(B), I(H), I(H), I(Ci), I(Ho), I(Ho), I(Co), I(K), I(K), I(s), I(p), I(p), I(1), STREAM())
        assert rc == 0, rc
    os.environ["DSF_X6_WRW_PATCH"] = "0"                   # the gather kernel, as the small maps run
    run("wrw %dx%dx%d->%d k%d s%d" % (H, H, Ci, Co, K, s), f, 2.0 * B * Ho * Ho * Co * Ci * K * K)

print("forward, igemm_x6p_kernel (variant %s)" % args.variant)
for (Ci, Co, H) in ([] if args.wrw_only else [(128, 128, 32), (256, 256, 16), (512, 512, 8), (64, 64, 64), (488, 256, 64)]):
    x, y = nhwc(B, Ci, H), torch.empty(B, H, H, Co, device="cuda")
    w = torch.randn(3, 3, Ci, Co, device="cuda")
    img = torch.empty(lib.dsf_conv_x6_image_bytes(I(3), I(3), I(Ci), I(Co)), device="cuda", dtype=torch.uint8)
    assert lib.dsf_conv_x6_split_weights(P(w), P(img), I(3), I(3), I(Ci), I(Co), I(0), STREAM()) == 0
    def f():
        rc = lib.dsf_conv_x6_forward(P(x), P(img), P(None), P(y), I(B), I(H), I(H), I(Ci), I(H), I(H), I(Co), I(3), I(3), I(1), I(1), I(1), I(1), I(0), STREAM())
        assert rc == 0, rc
    run("fwd %dx%dx%d->%d k3" % (H, H, Ci, Co), f, 2.0 * B * H * H * Co * Ci * 9)
