"""Reproducer of the round-4 finding: the 256-thread MANO backward, compiled WITH the SLP vectoriser (the compiler's default),
returns wrong bits while conv_x6 workgroups share its CUs; compiled with -fno-slp-vectorize it never does.

  python tools/platform/mano_beside_conv_x6.py            (GPU box; builds both variants of dsf_amd/csrc/mano.hip into /tmp)

For each build: one backward call alone = the reference bits; then 200 calls, each launched right after backward-weights (or
forward) conv_x6 launches were queued on a SECOND stream, compared bitwise with the reference.  Also run with other kernels
as the side load (the fp32-MFMA backward-weights kernel, a rocBLAS GEMM, elementwise adds) -- none of them disturbs either
build.  The damage, where it happens, is always the FIRST component of d/d(v_posed) of 16 consecutive vertices = lanes 48-63 of
one wave, i.e. the low half of a v_pk_fma_f32 / v_pk_mul_f32 result fed by the first dword of a broadcast ds_read_b128.
(A chain of bare v_pk_fma_f32 -- tools/platform/pk_fp32_beside_mfma.* -- is NOT disturbed: the trigger is narrower than
"packed FP32 beside bf16 MFMA", and was not isolated further.)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dsf_amd import _lib as L, nn_conv
from dsf_amd._lib import F, I, ptr, stream_ptr
from dsf_amd.render_model.mano_layer import Render

FLAGS = "--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -shared".split()
src = open(os.path.join(ROOT, "dsf_amd", "csrc", "mano.hip")).read().replace('#include "common.h"', '#include "%s"' % os.path.join(ROOT, "dsf_amd", "csrc", "common.h"))
open("/tmp/mano_variant.hip", "w").write(src)
libs = {}
for tag, extra in (("slp (compiler default)", []), ("-fno-slp-vectorize (shipped)", ["-fno-slp-vectorize"])):
    out = "/tmp/libmano_%d.so" % len(libs)
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["/tmp/mano_variant.hip", "-o", out])
    libs[tag] = ctypes.CDLL(out)

render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
cs = ctypes.byref(render.mano_layer._native().c_struct)
B, W = 32, 62
g = torch.Generator(device="cuda").manual_seed(1)
paras = torch.randn(B, W, device="cuda", generator=g) * 0.4
paras[:, 58] = 1.0
col = lambda t_, c: ctypes.c_void_p(t_.data_ptr() + 4 * c)
verts, joints, save = torch.empty(B, 779, 3, device="cuda"), torch.empty(B, 21, 3, device="cuda"), torch.empty(B, 5248, device="cuda")
assert L.lib().dsf_mano_forward(cs, col(paras, 48), col(paras, 3), col(paras, 0), col(paras, 58), I(B), I(45), I(3), I(W), F(1000.0), F(1.0), ptr(verts), ptr(joints), ptr(None), ptr(save), stream_ptr()) == 0
gV, gJ = torch.randn(B, 779, 3, device="cuda", generator=g), torch.randn(B, 21, 3, device="cuda", generator=g)
x = torch.randn(32, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
gy = torch.randn(32, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
a_mm = torch.randn(8192, 8192, device="cuda")
mf_ops = (torch.randint(0, 256, (4096,), device="cuda", dtype=torch.int32) | 0x3f00)
mf_ops = (mf_ops | (mf_ops.roll(1) << 16)).contiguous()
mf_out = torch.empty(512 * 256, device="cuda")
conv = nn_conv.Conv2d(256, 256, 3, 1, 1, bias=False).cuda()
big = torch.nn.Sequential(*[nn_conv.Conv2d(512, 512, 3, 1, 1, bias=False) for _ in range(8)]).cuda()
nn_conv.manage_weights(big.parameters())
with torch.no_grad():
    big(torch.randn(1, 512, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last))      # creates the cached images
side = torch.cuda.Stream()


def side_load(kind):
    if kind == "conv_x6 backward-weights":
        for _ in range(2):
            nn_conv._wrw(x, gy, 3, 3, 1, (1, 1))
    elif kind == "conv_x6 forward":
        with torch.no_grad():
            for _ in range(2):
                conv(x)
    elif kind == "fp32-MFMA backward-weights":
        dw = torch.zeros(3, 3, 256, 256, device="cuda")
        for _ in range(2):
            assert L.lib().dsf_conv_igemm_wrw(nn_conv.ptr_nhwc(x), nn_conv.ptr_nhwc(gy), ptr(dw), I(32), I(64), I(64), I(256), I(64), I(64), I(256), I(3), I(3), I(1), I(1), I(1), I(1), stream_ptr()) == 0
    elif kind == "conv_x6 weight split (no MFMA, no LDS)":
        for _ in range(40):
            nn_conv.refresh_images(big.parameters(), owner=big)
    elif kind == "bare bf16 MFMA loop":
        assert L.lib().dsf_mfma_bf16_probe(ctypes.c_void_p(mf_ops.data_ptr()), ctypes.c_void_p(mf_out.data_ptr()), ctypes.c_int(512), ctypes.c_int(3000), stream_ptr()) > 0
    elif kind == "rocBLAS GEMM":
        torch.mm(a_mm, a_mm)
    elif kind == "elementwise":
        for _ in range(8):
            torch.add(x, gy)


for tag, lib in libs.items():
    def bwd():
        gp, scratch = torch.zeros(B, W, device="cuda"), torch.empty(B, 2560, device="cuda")
        assert lib.dsf_mano_backward(cs, col(paras, 3), col(paras, 0), col(paras, 58), ptr(save), ptr(gV), ptr(gJ), I(B), I(45), I(3), I(W), F(1000.0), F(1.0), col(gp, 48), col(gp, 3), col(gp, 0), col(gp, 58), ptr(scratch), stream_ptr()) == 0
        torch.cuda.synchronize()
        return gp, scratch[:, :2334].clone()
    ref, ref_s = bwd()
    print("build: %s" % tag)
    for kind in ("none", "conv_x6 backward-weights", "conv_x6 forward", "conv_x6 weight split (no MFMA, no LDS)", "bare bf16 MFMA loop", "fp32-MFMA backward-weights", "rocBLAS GEMM", "elementwise"):
        bad, lanes, comps = 0, set(), set()
        for it in range(200):
            if kind != "none":
                with torch.cuda.stream(side):
                    side_load(kind)
            gp, sc = bwd()
            if not torch.equal(gp, ref):
                bad += 1
                nz = (sc != ref_s).nonzero()
                lanes |= set(((nz[:, 1] // 3) % 64).tolist())
                comps |= set((nz[:, 1] % 3).tolist())
        print("   side load %-28s %3d of 200 calls differ from the call that ran alone%s" % (kind + ":", bad, "" if not bad else "; lanes of d/d(v_posed) hit: %s, components: %s" % (sorted(lanes), sorted(comps))))
