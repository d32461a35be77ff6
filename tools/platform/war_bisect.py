"""Bisect of the round-4 co-residency finding on the GPU box (victim side by assembly edits, aggressor side by synthetic loops).

  python tools/platform/war_variants.py && hipcc ... war_kernels.hip      (CPU, see those files)
  python tools/platform/war_bisect.py [calls]                              (GPU box)

Part 1: every assembly variant of the SLP-vectorised `mano_skin_bwd_kernel` (tools/platform/_war/*.hsaco, loaded with hipModuleLoad and
launched with the product's own arguments) alone and beside conv_x6 backward-weights launches on a second stream; bitwise against
the same variant's run on an idle GPU.  Reports how many calls differ, which lanes and which component of d/d(v_posed).
Part 2: the synthetic victims of war_kernels.hip beside conv_x6 backward-weights and beside the synthetic aggressors.
Part 3: the `asis` MANO variant beside the synthetic aggressors."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dsf_amd import _lib as L, nn_conv
from dsf_amd._lib import F, I, ptr, stream_ptr
from dsf_amd.render_model.mano_layer import Render

CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
WAR = os.path.join(ROOT, "tools", "platform", "_war")
KERNEL = b"_ZN12_GLOBAL__N_120mano_skin_bwd_kernelE14dsf_mano_modelPKfS2_S2_S2_iffPfS3_"
hip = ctypes.CDLL("libamdhip64.so")
war = ctypes.CDLL(os.path.join(WAR, "libwar.so"))


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s -> %d" % (what, rc))


class Module:
    def __init__(self, path):
        self.mod, self.fn = ctypes.c_void_p(), ctypes.c_void_p()
        check(hip.hipModuleLoad(ctypes.byref(self.mod), path.encode()), "hipModuleLoad " + path)
        check(hip.hipModuleGetFunction(ctypes.byref(self.fn), self.mod, KERNEL), "hipModuleGetFunction")

    def launch(self, argbuf, B, stream):
        size = ctypes.c_size_t(len(argbuf))
        extra = (ctypes.c_void_p * 5)(1, ctypes.cast(argbuf, ctypes.c_void_p), 2, ctypes.cast(ctypes.byref(size), ctypes.c_void_p), 3)
        check(hip.hipModuleLaunchKernel(self.fn, B, 1, 1, 256, 1, 1, 0, stream, None, extra), "hipModuleLaunchKernel")


render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
cstruct = render.mano_layer._native().c_struct
B, W = 32, 62
g = torch.Generator(device="cuda").manual_seed(1)
paras = torch.randn(B, W, device="cuda", generator=g) * 0.4
paras[:, 58] = 1.0
col = lambda t_, c: ctypes.c_void_p(t_.data_ptr() + 4 * c)
verts, joints, save = torch.empty(B, 779, 3, device="cuda"), torch.empty(B, 21, 3, device="cuda"), torch.empty(B, 5248, device="cuda")
assert L.lib().dsf_mano_forward(ctypes.byref(cstruct), col(paras, 48), col(paras, 3), col(paras, 0), col(paras, 58), I(B), I(45), I(3), I(W), F(1000.0), F(1.0), ptr(verts), ptr(joints), ptr(None), ptr(save), stream_ptr()) == 0
gV, gJ = torch.randn(B, 779, 3, device="cuda", generator=g), torch.randn(B, 21, 3, device="cuda", generator=g)
x = torch.randn(32, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
gy = torch.randn(32, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
ops = torch.randint(0, 2 ** 31 - 1, (4 * 1024 * 1024,), device="cuda", dtype=torch.int32)      # 16 MB of operand bits
agg_out = torch.empty(4096 * 256, device="cuda")
side = torch.cuda.Stream()
AGG = {"valu": 0, "mfma+valu": 1, "mfma+lds": 2, "lds+vmem+valu": 3, "mfma+lds reads": 4, "mfma+lds writes": 5, "lds only": 6, "fp32 mfma+lds": 7}
PARTS = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else {"1", "2", "3"}
DUMP = {}


def side_load(kind):
    if kind == "none":
        return
    with torch.cuda.stream(side):
        s = ctypes.c_void_p(side.cuda_stream)
        if kind == "conv_x6 wrw":
            for _ in range(2):
                nn_conv._wrw(x, gy, 3, 3, 1, (1, 1))
        else:
            check(war.war_aggressor(AGG[kind], ctypes.c_void_p(ops.data_ptr()), ctypes.c_size_t(ops.numel() // 4), ctypes.c_void_p(agg_out.data_ptr()),
                                    2048, {"valu": 3000, "lds+vmem+valu": 1500, "lds only": 1500}.get(kind, 600), s), "aggressor " + kind)


def mano_args(scratch, g_cam):
    buf = (ctypes.c_char * 176)()
    ctypes.memmove(buf, ctypes.byref(cstruct), 112)
    def put(off, val, ty):
        ctypes.memmove(ctypes.byref(buf, off), ctypes.byref(ty(val)), ctypes.sizeof(ty))
    put(112, paras.data_ptr() + 4 * 58, ctypes.c_uint64)       # cam
    put(120, save.data_ptr(), ctypes.c_uint64)
    put(128, gV.data_ptr(), ctypes.c_uint64)
    put(136, gJ.data_ptr(), ctypes.c_uint64)
    put(144, W, ctypes.c_int32)
    put(148, 1000.0, ctypes.c_float)
    put(152, 1.0, ctypes.c_float)
    put(160, g_cam.data_ptr(), ctypes.c_uint64)
    put(168, scratch.data_ptr(), ctypes.c_uint64)
    return buf


def run_mano(mod, kinds):
    def call():
        scratch, g_cam = torch.zeros(B, 2560, device="cuda"), torch.zeros(B, W, device="cuda")
        mod.launch(mano_args(scratch, g_cam), B, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        return scratch[:, :2334].clone()
    ref = call()
    res = {}
    for kind in kinds:
        bad, lanes, comps = 0, set(), set()
        for _ in range(CALLS):
            side_load(kind)
            sc = call()
            if not torch.equal(sc, ref):
                bad += 1
                if "bad" not in DUMP:
                    DUMP.update(bad=sc.cpu().numpy(), ref=ref.cpu().numpy(), kind=kind)
                nz = (sc != ref).nonzero()
                lanes |= set(((nz[:, 1] // 3) % 64).tolist())
                comps |= set((nz[:, 1] % 3).tolist())
        res[kind] = (bad, sorted(lanes), sorted(comps))
        torch.cuda.synchronize()
    return ref, res


def fmt(r):
    bad, lanes, comps = r
    return "%3d of %d" % (bad, CALLS) + ("" if not bad else "  lanes %s components %s" % (("%d-%d" % (lanes[0], lanes[-1]) if lanes == list(range(lanes[0], lanes[-1] + 1)) else lanes), comps))


print("== part 1: assembly variants of mano_skin_bwd_kernel (SLP build), %d calls each ==" % CALLS)
refs = {}
for name in ("noslp", "asis", "wait0", "wait0_nop", "b128_split", "scalar_all", "scalar_loop", "scalar_final", "scalar_first"):
    if "1" not in PARTS:
        break
    mod = Module(os.path.join(WAR, name + ".hsaco"))
    refs[name], res = run_mano(mod, ("none", "conv_x6 wrw", "mfma+lds"))
    print("   %-12s alone: %s | beside conv_x6 backward-weights: %s | beside the synthetic mfma+lds loop: %s" % (name, fmt(res["none"]), fmt(res["conv_x6 wrw"]), fmt(res["mfma+lds"])))
    sys.stdout.flush()
if DUMP:
    import numpy as np
    np.savez(os.path.join(ROOT, "gpurun_out", "war_dump.npz"), save=save.cpu().numpy(), weights=render.mano_layer.weights.detach().cpu().numpy() if hasattr(render.mano_layer, "weights") else np.zeros(1),
             gV=gV.cpu().numpy(), gJ=gJ.cpu().numpy(), paras=paras.cpu().numpy(), **{k: v for k, v in DUMP.items() if k != "kind"})
    nz = np.argwhere(DUMP["bad"] != DUMP["ref"])
    print("   first damaged call (%s): samples %s, vertices %s" % (DUMP["kind"], sorted(set(nz[:, 0].tolist()))[:8], sorted(set((nz[:, 1] // 3).tolist()))[:40]))
for name in refs:
    if name != "noslp":
        d = (refs[name] - refs["noslp"]).abs().max().item()
        print("   idle-GPU result of %-12s vs the scalar build: max abs difference %.3g (SLP changes no rounding: expected 0)" % (name, d))

print("== part 2: synthetic victims (acc += 1.0 * src, src overwritten behind the op), 1024 workgroups x n = 20000 ==")
VICT = {0: "v_pk_fma, source overwritten by the next instruction", 1: "v_pk_fma, s_nop 0, overwrite", 2: "two v_fma (control)",
        3: "v_pk_fma, s_waitcnt lgkmcnt(0), overwrite from the LDS data", 4: "v_pk_fma, high half overwritten first", 5: "v_pk_fma, one independent VALU, overwrite"}
out = torch.empty(1024 * 512, device="cuda")
N = 20000
for mode, what in VICT.items():
    if "2" not in PARTS:
        break
    line = []
    for kind in ("none", "conv_x6 wrw", "valu", "mfma+valu", "mfma+lds", "lds+vmem+valu"):
        bad, lanes, halves = 0, set(), set()
        for _ in range(max(CALLS // 5, 10)):
            side_load(kind)
            out.zero_()
            check(war.war_victim(mode, ctypes.c_void_p(out.data_ptr()), 1024, N, ctypes.c_float(1.0), ctypes.c_float(3.0), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "victim")
            torch.cuda.synchronize()
            nz = (out != float(N)).nonzero().flatten()
            if nz.numel():
                bad += 1
                lanes |= set(((nz // 2) % 64).tolist())
                halves |= set((nz % 2).tolist())
        line.append("%s: %d%s" % (kind, bad, "" if not bad else " (lanes %s halves %s)" % (sorted(lanes), sorted(halves))))
        torch.cuda.synchronize()
    print("   victim %d (%s), launches with a wrong lane out of %d -- %s" % (mode, what, max(CALLS // 5, 10), "; ".join(line)))
    sys.stdout.flush()

print("== part 3: the SLP build as compiled, beside the synthetic aggressors ==")
mod = Module(os.path.join(WAR, "asis.hsaco"))
_, res = run_mano(mod, ("valu", "mfma+valu", "mfma+lds", "lds+vmem+valu", "mfma+lds reads", "mfma+lds writes", "lds only", "fp32 mfma+lds") if "3" in PARTS else ())
for k, r in res.items():
    print("   beside %-16s %s" % (k + ":", fmt(r)))
