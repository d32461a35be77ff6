#!/usr/bin/env python
"""Route to PINNING oracle/p3d_ref.c against the real wheel.

Run on ANY machine that has ``pytorch3d==0.4.0`` with CUDA (the reference's environment: README.md:39-42); neither this
build container nor the GPU box has it, which is why the p3d oracle is "parity unpinned":

    python tests/golden/make_golden_p3d.py            # writes tests/golden/p3d_golden.npz (arrays only)

It feeds fixed, seeded inputs -- posed hand meshes of the repo's synthetic MANO asset, lattice-snapped triangle soups
(edges through pixel centres, exact z ties, degenerate / behind-camera / camera-plane-crossing faces), point clouds near
the meshes -- through the ops the reference calls and dumps inputs + outputs:

  * ``pytorch3d._C.rasterize_meshes`` (render_model/mano_layer.py:946-952 settings: image 640, blur 0, 1 face / pixel)
    TWICE: ``bin_size=0`` (naive path) and ``bin_size=64`` (the coarse-to-fine path ``bin_size=None`` selects on CUDA
    for a 640-pixel image -- the one the reference actually runs), and ``rasterize_meshes_backward`` for a seeded
    ``grad_zbuf``;
  * ``MeshRasterizer(PerspectiveCameras(...))`` exactly as mano_layer.py:935-952 builds it, on world-space vertices
    (pins the camera transform of SURVEY Appendix A.1);
  * ``pytorch3d._C.point_face_dist_forward / _backward`` (metric/meshLoss.py:52, 63).

tests/test_oracle_p3d.py::test_oracle_against_pytorch3d_golden consumes the file when it is present (and is skipped
otherwise): index outputs must agree everywhere except on pixels that oracle/p3d_exact.py classifies as decided by
float32 rounding (nvcc contracts a*b - c*d into FMAs, so bit-equality with a CUDA build is not attainable), float
outputs to 1e-5.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
CAM = (588.03, 587.07, 320.0, 240.0)


def fixed_inputs():
    """Seeded inputs shared by this script and the consuming test (pure numpy / torch-CPU, no pytorch3d)."""
    import torch
    from dsf_amd.assets import build_synthetic_mano
    from dsf_amd.train_step import synthetic_batch
    from oracle import hand_ref as H
    hm = H.HandModel(build_synthetic_mano(0))
    p, c, cube = synthetic_batch(4, "cpu", seed=101)
    with torch.no_grad():
        v, _ = H.mano_vertices(hm, p[:, :3], p[:, 3:48], p[:, 48:58], p[:, 58:62], 1 / 125)
    world = (v * cube[:, None] / 2 + c[:, None]).numpy().astype(np.float32)
    faces = hm.faces.numpy().astype(np.int64)
    rng = np.random.default_rng(202)
    S = 64
    lattice = (-1 + (2 * np.arange(S) + 1) / S).astype(np.float32)
    soups = []
    for k in range(3):
        n = 160
        tri = np.empty((n, 3, 3), dtype=np.float32)
        tri[..., 0] = rng.choice(lattice, (n, 3)) if k < 2 else rng.uniform(-1.2, 1.2, (n, 3))
        tri[..., 1] = rng.choice(lattice, (n, 3)) if k < 2 else rng.uniform(-1.2, 1.2, (n, 3))
        tri[..., 2] = rng.choice(np.array([1.0, 2.0, 2.0, 3.0], dtype=np.float32), (n, 1)) if k == 0 else \
            rng.uniform(0.5, 4.0, (n, 3))
        tri[:8, :, 2] = -1.0                                  # behind the camera
        tri[8:16, 0, 2] = -0.5                                # crossing the camera plane (coarse path: zmin < kEpsilon skip)
        tri[16:20, 1] = tri[16:20, 0]                         # degenerate
        tri[20:24] = tri[24:28]                               # duplicated faces: exact ties
        soups.append(tri)
    pts = (world[:, ::3][:, :256] + rng.normal(size=(4, 256, 3)).astype(np.float32) * 4.0).astype(np.float32)
    gz = rng.normal(size=(4, 640, 640)).astype(np.float32)
    return {"world": world, "faces": faces, "soups": np.stack(soups), "soup_size": S, "points": pts, "grad_zbuf_seed": 202,
            "grad_zbuf_head": gz[:, :4, :4].copy()}, gz


def main():
    import torch
    import pytorch3d
    from pytorch3d import _C
    from pytorch3d.renderer import PerspectiveCameras, RasterizationSettings, MeshRasterizer
    from pytorch3d.structures import Meshes
    if not pytorch3d.__version__.startswith("0.4.0"):
        print("WARNING: pytorch3d %s, the reference pins 0.4.0" % pytorch3d.__version__)
    dev = torch.device("cuda")
    inp, gz = fixed_inputs()
    out = {"pytorch3d_version": np.array(pytorch3d.__version__), "torch_version": np.array(torch.__version__)}
    out.update({k: np.asarray(v) for k, v in inp.items()})
    world, faces = torch.tensor(inp["world"], device=dev), torch.tensor(inp["faces"], device=dev)
    B, nf = world.shape[0], faces.shape[0]

    # (1) MeshRasterizer as the reference builds it (mano_layer.py:935-952)
    R = torch.eye(3).unsqueeze(0)
    R[:, 0, 0] = -1
    R[:, 1, 1] = -1
    cams = PerspectiveCameras(focal_length=((CAM[0], CAM[1]),), principal_point=((CAM[2], CAM[3]),), image_size=((640, 480),),
                              device=dev, R=R.to(dev), T=torch.zeros(1, 3, device=dev))
    rast = MeshRasterizer(cameras=cams, raster_settings=RasterizationSettings(image_size=640, blur_radius=0, faces_per_pixel=1))
    meshes = Meshes(verts=world, faces=faces.unsqueeze(0).repeat(B, 1, 1))
    frag = rast(meshes)
    out["mr_pix_to_face"], out["mr_zbuf"] = frag.pix_to_face.cpu().numpy(), frag.zbuf.cpu().numpy()
    out["mr_bary"], out["mr_dists"] = frag.bary_coords.cpu().numpy()[:, ::4, ::4], frag.dists.cpu().numpy()[:, ::4, ::4]
    screen = rast.transform(meshes).verts_padded()                                   # (x_ndc, y_ndc, z_view)
    out["mr_screen_verts"] = screen.cpu().numpy()

    # (2) _C.rasterize_meshes on explicit face_verts: naive and coarse-to-fine; backward
    def run(fv, first, cnt, S, tag):
        for name, bin_size in (("naive", 0), ("binned", 64 if S == 640 else 16)):
            p2f, zbuf, bary, dists = _C.rasterize_meshes(fv, first, cnt, S, 0.0, 1, bin_size, 10000, False, False, False)
            out["%s_%s_p2f" % (tag, name)], out["%s_%s_zbuf" % (tag, name)] = p2f.cpu().numpy(), zbuf.cpu().numpy()
            out["%s_%s_bary" % (tag, name)] = bary.cpu().numpy()
        return p2f, zbuf
    fv = screen[:, faces].reshape(-1, 3, 3).contiguous()
    first = torch.arange(B, device=dev) * nf
    cnt = torch.full((B,), nf, device=dev, dtype=torch.int64)
    p2f, zbuf = run(fv, first, cnt, 640, "hand")
    g = torch.tensor(gz, device=dev).unsqueeze(-1)
    grad = _C.rasterize_meshes_backward(fv, p2f, g, torch.zeros_like(g).unsqueeze(-1).expand(-1, -1, -1, -1, 3).contiguous(),
                                        torch.zeros_like(g), False, False)
    out["hand_grad_face_verts"] = grad.cpu().numpy()
    S = int(inp["soup_size"])
    for k, soup in enumerate(inp["soups"]):
        t = torch.tensor(soup, device=dev)
        run(t, torch.zeros(1, device=dev, dtype=torch.int64), torch.tensor([t.shape[0]], device=dev), S, "soup%d" % k)

    # (3) point-face distance
    pts = torch.tensor(inp["points"], device=dev).reshape(-1, 3).contiguous()
    tris = world[:, faces].reshape(-1, 3, 3).contiguous()
    P = inp["points"].shape[1]
    pfirst = torch.arange(B, device=dev) * P
    d, idx = _C.point_face_dist_forward(pts, pfirst, tris, first, P)
    out["pfd_dists"], out["pfd_idxs"] = d.cpu().numpy(), idx.cpu().numpy()
    gp, gt = _C.point_face_dist_backward(pts, tris, idx, torch.ones_like(d))
    out["pfd_grad_points"], out["pfd_grad_tris"] = gp.cpu().numpy(), gt.cpu().numpy()
    np.savez_compressed(os.path.join(HERE, "p3d_golden.npz"), **out)
    print("p3d_golden.npz", os.path.getsize(os.path.join(HERE, "p3d_golden.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
