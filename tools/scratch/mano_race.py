"""one-off: is the MANO backward deterministic while backward-weights kernels run on a second stream?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dsf_amd import _lib as L, nn_conv
from dsf_amd._lib import F, I, ptr, stream_ptr
from dsf_amd.render_model.mano_layer import Render
render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
cs = ctypes.byref(render.mano_layer._native().c_struct)
B, W = int(os.environ.get("B", "8")), 62
g = torch.Generator(device="cuda").manual_seed(1)
paras = torch.randn(B, W, device="cuda", generator=g) * 0.4
paras[:, 58] = 1.0
col = lambda t_, c: ctypes.c_void_p(t_.data_ptr() + 4 * c)
verts, joints, save = torch.empty(B, 779, 3, device="cuda"), torch.empty(B, 21, 3, device="cuda"), torch.empty(B, 5248, device="cuda")
assert L.lib().dsf_mano_forward(cs, col(paras, 48), col(paras, 3), col(paras, 0), col(paras, 58), I(B), I(45), I(3), I(W), F(1000.0), F(1.0), ptr(verts), ptr(joints), ptr(None), ptr(save), stream_ptr()) == 0
gV, gJ = torch.randn(B, 779, 3, device="cuda", generator=g), torch.randn(B, 21, 3, device="cuda", generator=g)
x = torch.randn(32, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
gy = torch.randn(32, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
side = torch.cuda.Stream()
v1 = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmano_v1.so"))
USE_V1 = os.environ.get("V1", "0") == "1"
VAR = os.environ.get("VAR")
var = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmano_var%s.so" % VAR)) if VAR else None
LOAD = os.environ.get("LOAD", "wrw")
a_mm = torch.randn(8192, 8192, device="cuda")
conv = nn_conv.Conv2d(256, 256, 3, 1, 1, bias=False).cuda()
def side_load():
    if LOAD == "wrw":
        for _ in range(2): nn_conv._wrw(x, gy, 3, 3, 1, (1, 1))
    elif LOAD == "wrw_f32":
        dw = torch.zeros(3, 3, 256, 256, device="cuda")
        for _ in range(2):
            assert L.lib().dsf_conv_igemm_wrw(nn_conv.ptr_nhwc(x), nn_conv.ptr_nhwc(gy), ptr(dw), I(32), I(64), I(64), I(256), I(64), I(64), I(256), I(3), I(3), I(1), I(1), I(1), I(1), stream_ptr()) == 0
    elif LOAD == "fwd":
        with torch.no_grad():
            for _ in range(2): conv(x)
    elif LOAD == "matmul":
        torch.mm(a_mm, a_mm)
    elif LOAD == "elementwise":
        for _ in range(8): torch.add(x, gy)
def bwd():
    gp = torch.zeros(B, W, device="cuda"); scratch = torch.full((B, 2560), float("nan"), device="cuda")
    if USE_V1:
        assert v1.dsf_mano_backward_v1(cs, col(paras, 3), col(paras, 0), col(paras, 58), ptr(save), ptr(gV), ptr(gJ), I(B), I(45), I(3), I(W), F(1000.0), F(1.0), col(gp, 48), col(gp, 3), col(gp, 0), col(gp, 58), stream_ptr()) == 0
        return gp, torch.zeros(B, 2560, device="cuda")
    fn = var.dsf_mano_backward_var if var is not None else L.lib().dsf_mano_backward
    assert fn(cs, col(paras, 3), col(paras, 0), col(paras, 58), ptr(save), ptr(gV), ptr(gJ), I(B), I(45), I(3), I(W), F(1000.0), F(1.0), col(gp, 48), col(gp, 3), col(gp, 0), col(gp, 58), ptr(scratch), stream_ptr()) == 0
    return gp, scratch
ref, ref_s = bwd(); torch.cuda.synchronize()
for load in (False, True):
    bad = bad_s = 0
    worst = 0.0
    for it in range(200):
        if load:
            with torch.cuda.stream(side):
                side_load()
        gp, sc = bwd()
        torch.cuda.synchronize()
        if not torch.equal(gp, ref):
            bad += 1; worst = max(worst, float((gp - ref).abs().max()))
            if bad <= 3:
                d = (gp - ref).abs()
                print("   it %d: differing columns %s rows %s" % (it, sorted(set(d.nonzero()[:, 1].tolist())), sorted(set(d.nonzero()[:, 0].tolist()))))
        if not torch.equal(sc[:, :2526], ref_s[:, :2526]):
            bad_s += 1
            if bad_s <= 3:
                d = (sc[:, :2526] - ref_s[:, :2526]).abs()
                nz = d.nonzero()
                print("   it %d: scratch differs at %d entries; column range %d..%d" % (it, nz.shape[0], int(nz[:, 1].min()), int(nz[:, 1].max())))
                for r_, c_ in nz[:4].tolist():
                    print("        [%d, %d] ref %.9g got %.9g ratio %.6g" % (r_, c_, float(ref_s[r_, c_]), float(sc[r_, c_]), float(sc[r_, c_]) / float(ref_s[r_, c_])))
    print("mano %s, side-stream load %s %s: %d of 200 backward calls differ from the first (max abs %.3e); scratch differs %d" % (("v1" if USE_V1 else "var" + VAR if VAR else "new"), LOAD, load, bad, worst, bad_s))
