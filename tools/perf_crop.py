"""Timing of the fused crop rasteriser (dsf_render_crop_forward) at B = 32: default inputs, and a hand pushed out of the crop
(every tile empty: the per-workgroup prologue + the empty-tile walk).  DSF_CROP_WG_TARGET varies the workgroups per sample."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd import ops
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.train_step import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
p, c, cube = synthetic_batch(B, "cuda", seed=123)
mano = render.mano_layer
with torch.no_grad():
    v, _ = mano.get_mano_vertices(p[:, :3], p[:, 3:48], p[:, 48:58], p[:, 58:62], 1 / 125)
    verts = (v * cube.unsqueeze(1) / 2 + c.unsqueeze(1)).contiguous()
    c2, M, _, _ = ops.crop_setup(c, cube, render.cam, 128)
    minv = torch.linalg.inv_ex(M)[0].contiguous()
    cz, cbz = c2[:, 2].contiguous(), cube[:, 2].contiguous()
    def t(vv, n=100):
        run = lambda: ops.RenderCropFunction.apply(vv, mano.faces_i32, minv, render.resize_rowmap, cz, cbz, render.cam, 640, 128)
        for _ in range(10): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    far = verts.clone(); far[..., 0] += 2000.0
    print("B=%d  WG_TARGET=%s: hand in crop %.1f us | hand outside %.1f us" % (B, os.environ.get("DSF_CROP_WG_TARGET", "1024"), t(verts), t(far)))
