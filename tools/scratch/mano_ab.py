"""one-off: new 256-thread MANO kernels vs the round-3 1024-thread kernels (bitwise forward, gradients, time alone)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dsf_amd import _lib as L, ops
from dsf_amd._lib import F, I, ptr, stream_ptr
from dsf_amd.render_model.mano_layer import Render
v1 = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmano_v1.so"))
render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
model = render.mano_layer
cs = ctypes.byref(model._native().c_struct)
for B, ncomp, rot_dim, packed in ((32, 45, 3, True), (1, 45, 3, True), (7, 45, 4, True), (33, 30, 3, False), (192, 45, 3, True)):
    g = torch.Generator(device="cuda").manual_seed(B)
    W = rot_dim + 59
    paras = torch.randn(B, W, device="cuda", generator=g) * 0.4
    paras[:, rot_dim + 55] = 1 + 0.1 * torch.randn(B, device="cuda", generator=g)
    if packed:
        beta, theta, rot, cam, ps = paras[:, rot_dim + 45:], paras[:, rot_dim:], paras, paras[:, rot_dim + 55:], W
        colp = lambda t_: ctypes.c_void_p(t_.data_ptr())
    else:
        beta, theta, rot, cam, ps = paras[:, rot_dim + 45:rot_dim + 55].contiguous(), paras[:, rot_dim:rot_dim + ncomp].contiguous(), paras[:, :rot_dim].contiguous(), paras[:, rot_dim + 55:].contiguous(), 0
        colp = ptr
    k1, k2 = 1000.0, 1.0 / 1.3
    out = {}
    gV, gJ = torch.randn(B, 779, 3, device="cuda", generator=g), torch.randn(B, 21, 3, device="cuda", generator=g)
    for tag in ("v1", "new"):
        verts, joints, Rs = torch.empty(B, 779, 3, device="cuda"), torch.empty(B, 21, 3, device="cuda"), torch.empty(B, 15, 3, 3, device="cuda")
        save = torch.empty(B, 5248, device="cuda")
        gp = torch.zeros(B, W, device="cuda")
        gb, gt, gr, gc = (gp[:, rot_dim + 45:], gp[:, rot_dim:], gp, gp[:, rot_dim + 55:]) if packed else (torch.empty(B, 10, device="cuda"), torch.empty(B, ncomp, device="cuda"), torch.empty(B, rot_dim, device="cuda"), torch.empty(B, 4, device="cuda"))
        scratch = torch.empty(B, 2560, device="cuda")
        def fwd():
            f = v1.dsf_mano_forward_v1 if tag == "v1" else L.lib().dsf_mano_forward
            assert f(cs, colp(beta), colp(theta), colp(rot), colp(cam), I(B), I(ncomp), I(rot_dim), I(ps), F(k1), F(k2), ptr(verts), ptr(joints), ptr(Rs), ptr(save), stream_ptr()) == 0
        def bwd():
            if tag == "v1":
                assert v1.dsf_mano_backward_v1(cs, colp(theta), colp(rot), colp(cam), ptr(save), ptr(gV), ptr(gJ), I(B), I(ncomp), I(rot_dim), I(ps), F(k1), F(k2), colp(gb), colp(gt), colp(gr), colp(gc), stream_ptr()) == 0
            else:
                assert L.lib().dsf_mano_backward(cs, colp(theta), colp(rot), colp(cam), ptr(save), ptr(gV), ptr(gJ), I(B), I(ncomp), I(rot_dim), I(ps), F(k1), F(k2), colp(gb), colp(gt), colp(gr), colp(gc), ptr(scratch), stream_ptr()) == 0
        fwd(); bwd(); torch.cuda.synchronize()
        t = []
        for fn in (fwd, bwd):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5): fn()
            e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) * 20)
        grads = gp.clone() if packed else torch.cat([gr, gt, gb, gc], 1)
        out[tag] = (verts.clone(), joints.clone(), Rs.clone(), save[:, :5163].clone(), grads, t)
    a, b = out["v1"], out["new"]
    same = [bool(torch.equal(x, y)) for x, y in zip(a[:4], b[:4])]
    gerr = float((a[4] - b[4]).abs().max() / a[4].abs().max())
    print("B %3d ncomp %d rot %d packed %d: forward bitwise (verts, joints, Rs, save) %s | grad max rel diff %.2e | us fwd %.1f -> %.1f  bwd %.1f -> %.1f" % (B, ncomp, rot_dim, packed, same, gerr, a[5][0], b[5][0], a[5][1], b[5][1]))
