"""Where the crop rasteriser's 78 us go (B = 32): a diagnostic build of dsf_amd/csrc/raster.hip (-DCROP_STAMP: s_memtime stamps per
8x8 tile, the number of candidate faces and the longest per-lane pixel loop; the shipped library contains none of this) run on
bench.py's config-2 inputs.   python tools/crop_stamps.py   (GPU box; builds /tmp/libraster_stamp.so)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from dsf_amd import ops
from dsf_amd._lib import I, ptr, stream_ptr
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.train_step import synthetic_batch
so = "/tmp/libraster_stamp.so"
cs = os.path.join(ROOT, "dsf_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                       "-fno-slp-vectorize", "-fno-vectorize", "-DCROP_STAMP", "-I" + cs, "-I" + os.path.join(ROOT, "include"),
                       os.path.join(cs, "raster.hip"), os.path.join(cs, "api.hip"), "-o", so], stderr=subprocess.DEVNULL)
lib = ctypes.CDLL(so)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
p, c, cube = synthetic_batch(B, "cuda", seed=123)
if os.environ.get("CROP_STAMP_POSE") == "rest":            # what a freshly initialised MANO head predicts: rest pose, no rotation, unit scale
    p = torch.zeros_like(p); p[:, 58] = 1.0
mano = render.mano_layer
with torch.no_grad():
    v, _ = mano.get_mano_vertices(p[:, :3], p[:, 3:48], p[:, 48:58], p[:, 58:62], 1 / 125)
    verts = (v * cube.unsqueeze(1) / 2 + c.unsqueeze(1)).contiguous()
    c2, M, _, _ = ops.crop_setup(c, cube, render.cam, 128)
    minv = torch.linalg.inv_ex(M)[0].contiguous()
    cz, cbz = c2[:, 2].contiguous(), cube[:, 2].contiguous()
img = torch.empty(B, 1, 128, 128, device="cuda")
p2f = torch.empty(B, 128, 128, device="cuda", dtype=torch.int32)
st = torch.zeros(B, 256, 4, device="cuda", dtype=torch.int64)
assert lib.dsf_crop_stamp_buffer(ctypes.c_void_p(st.data_ptr())) == 0
for _ in range(3):
    rc = lib.dsf_render_crop_forward(ptr(verts), ptr(mano.faces_i32), ptr(minv), ptr(render.resize_rowmap), ptr(cz), ptr(cbz), ctypes.byref(render.cam),
                                     I(B), I(779), I(mano.faces_i32.shape[0]), I(640), I(128), ptr(img), ptr(p2f), stream_ptr())
    assert rc == 0, rc
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
cyc, cand, coop, start = s[..., 0], s[..., 1], s[..., 2] > 0, s[..., 3]
end = start + cyc
print("B = %d, 256 tiles per sample; shader-clock cycles (s_memtime) of the wave that writes the tile:" % B)
for name, sel in (("tiles a wave handles alone", ~coop), ("heavy tiles (> 64 candidate faces), all four waves", coop)):
    if sel.any():
        c = cyc[sel]
        print("  %-52s n = %5d (%.1f per sample): mean %6.0f, median %6.0f, 99 %% %6.0f, max %6.0f cycles; candidates (this wave's) mean %.1f max %.0f"
              % (name, sel.sum(), sel.sum() / B, c.mean(), np.median(c), np.percentile(c, 99), c.max(), cand[sel].mean(), cand[sel].max()))
print("  last tile of a sample finishes %.0f cycles after its workgroup started (mean over samples), worst sample %.0f cycles = %.1f us at 2.4 GHz"
      % (end.max(1).mean(), end.max(), end.max() / 2400))
