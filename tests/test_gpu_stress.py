"""Adversarial bit-exactness stress of the index-producing kernels against the C oracle: random triangle soups (not hand
meshes) with vertices snapped to the pixel-centre lattice (exact edge hits and exact z ties), duplicated faces, slivers,
degenerate faces, faces behind the camera, huge and tiny triangles; point clouds with points exactly on vertices, edges
and duplicated triangles for the argmin ties."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CAM = (588.03, 587.07, 320.0, 240.0)


def _soup(seed, F, S, snap):
    """NDC-space triangle soup (F,3,3): x, y in about [-1.2, 1.2], z in [1, 5] with adversarial structure."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1.1, 1.1, (F, 1, 2))
    size = 10.0 ** rng.uniform(-2.5, -0.1, (F, 1, 1))
    xy = c + rng.normal(size=(F, 3, 2)) * size
    z = rng.uniform(1.0, 5.0, (F, 3, 1))
    if snap:                                                     # vertices on the pixel-centre lattice: edges through samples
        xy = (np.round((xy + 1.0) * S / 2.0 - 0.5) + 0.5) * 2.0 / S - 1.0
        z = np.round(z * 4.0) / 4.0                              # few distinct depths: exact z ties between faces
    fv = np.concatenate([xy, z], -1).astype(np.float32)
    fv[F // 10:F // 10 + 5] = fv[:5]                             # duplicated faces (ties -> lowest index)
    fv[F // 5, 2] = fv[F // 5, 1]                                # degenerate (zero area)
    fv[F // 4, :, 2] = -1.0                                      # behind the camera
    fv[F // 3, :, :2] *= 40.0                                    # huge triangle covering everything
    fv[F // 3, :, 2] = 4.75
    return fv


@pytest.mark.parametrize("seed,snap", [(1, False), (2, True), (3, True), (4, False)])
def test_full_raster_bit_exact_on_triangle_soups(seed, snap):
    from dsf_amd import ops
    from oracle import p3d
    S, F, N = 96, 400, 2
    fv = np.concatenate([_soup(seed * 10 + i, F, S, snap) for i in range(N)])
    first = np.arange(N, dtype=np.int64) * F
    cnt = np.full(N, F, dtype=np.int64)
    p2f_o, z_o, bary_o, d_o = p3d.rasterize_meshes(fv, first, cnt, S)
    T = lambda a: torch.tensor(a, device="cuda")
    p2f, zbuf, bary, dists = ops.RasterizeMeshesFunction.apply(T(fv), T(first), T(cnt), S)
    assert (p2f_o >= 0).mean() > 0.5
    assert np.array_equal(p2f.cpu().numpy()[..., 0], p2f_o)
    assert np.array_equal(zbuf.cpu().numpy()[..., 0], z_o)
    assert np.array_equal(bary.cpu().numpy()[..., 0, :], bary_o)


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_point_face_argmin_bit_exact_with_ties(seed):
    from dsf_amd.metric.meshLoss import point_face_distance
    from oracle import p3d
    rng = np.random.default_rng(seed)
    Tn, P = 300, 900
    tris = rng.normal(size=(Tn, 3, 3)).astype(np.float32)
    tris[50:60] = tris[:10]                                      # duplicated triangles: argmin ties -> lowest index
    tris[70, 2] = tris[70, 1]                                    # degenerate
    tris = np.round(tris * 8.0) / 8.0 if seed % 2 else tris      # lattice vertices: many exactly equal distances
    pts = rng.normal(size=(P, 3)).astype(np.float32)
    pts[:100] = tris[rng.integers(0, Tn, 100), rng.integers(0, 3, 100)]                   # exactly on vertices
    e = rng.integers(0, Tn, 100)
    pts[100:200] = (tris[e, 0] + tris[e, 1]) / 2                                          # on edges
    pts[200:300] = tris[rng.integers(0, Tn, 100)].mean(1)                                 # inside faces
    pf = np.array([0, 500], dtype=np.int64)
    tf = np.array([0, 160], dtype=np.int64)
    d_o, i_o = p3d.point_face_dist_forward(pts, pf, tris, tf)
    Tt = lambda a: torch.tensor(a, device="cuda")
    d = point_face_distance(Tt(pts), Tt(pf), Tt(tris), Tt(tf), 500)
    assert np.array_equal(d.cpu().numpy(), d_o)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_crop_render_bit_exact_on_triangle_soups(seed):
    """the fused crop rasteriser (candidate rounds, face-parallel path, pixel-parallel fallback) on arbitrary geometry"""
    from dsf_amd import ops
    from dsf_amd.render_model.mano_layer import Render
    from test_gpu_parity import _oracle_crop
    render = Render("synthetic", "nyu", CAM, (640, 480)).cuda()
    rng = np.random.default_rng(seed)
    B, F = 3, 270
    c = np.stack([rng.uniform(-40, 40, B), rng.uniform(-40, 40, B), rng.uniform(500, 1100, B)], 1).astype(np.float32)
    cube = np.full((B, 3), 250.0, dtype=np.float32)
    verts = np.zeros((B, 3 * F, 3), dtype=np.float32)
    for b in range(B):
        ctr = rng.uniform(-110, 110, (F, 1, 2))
        size = 10.0 ** rng.uniform(-0.5, 2.0, (F, 1, 1))          # 0.3 mm .. 100 mm triangles around the crop centre
        xy = c[b, None, None, :2] + ctr + rng.normal(size=(F, 3, 2)) * size
        z = c[b, 2] + rng.uniform(-120, 120, (F, 3, 1))
        if seed % 2:
            z = np.round(z / 20.0) * 20.0                           # few distinct depths -> exact z ties
        tri = np.concatenate([xy, z], -1)
        tri[10:15] = tri[:5]                                        # duplicated faces
        tri[20, 2] = tri[20, 1]                                     # degenerate
        tri[30, :, 2] = -50.0                                       # behind the camera
        verts[b] = tri.reshape(-1, 3)
    faces = np.arange(3 * F, dtype=np.int32).reshape(F, 3)
    T = lambda a: torch.tensor(a, device="cuda")
    c2, M, _, _ = ops.crop_setup(T(c), T(cube), render.cam, 128)
    Minv = torch.inverse(M.cpu())
    img, p2f = ops.RenderCropFunction.apply(T(verts), T(faces), Minv.cuda(), render.resize_rowmap, c2[:, 2].contiguous(),
                                            T(cube)[:, 2].contiguous(), render.cam, 640, 128)
    exp_img, exp_f, _, _ = _oracle_crop(verts, faces.astype(np.int64), Minv.numpy(), render.resize_rowmap.cpu().numpy().astype(np.int64),
                                        c2[:, 2].cpu().numpy(), cube[:, 2])
    assert (exp_f >= 0).mean() > 0.2
    assert np.array_equal(p2f.cpu().numpy(), exp_f)
    assert np.array_equal(img.cpu().numpy(), exp_img)


def _mesh_soup(rng, V, F, scale, snap):
    """vertex pool + faces (not a manifold): lattice-snapped vertices (exactly equal distances), duplicated faces (ties -> the
    lowest index), degenerate faces, slivers; ``scale`` shrinks the whole thing -- at 0.02 the reference's `+ 1e-8` terms are no
    longer negligible against the triangles' Gram determinants (its `inside` region grows by up to tens of per cent), which
    is the regime the cull's scaled bounding spheres exist for"""
    verts = rng.normal(size=(V, 3)).astype(np.float32)
    if snap:
        verts = np.round(verts * 4.0) / 4.0
    verts = (verts * scale).astype(np.float32)
    # faces from nearby vertices (small triangles, so that the cull has something to skip) + some arbitrary ones
    order = np.argsort(verts[:, 0] + 0.37 * verts[:, 1])
    near = np.stack([order[np.clip(np.arange(F) % (V - 8) + rng.integers(0, 8, F), 0, V - 1)] for _ in range(3)], 1)
    faces = near.astype(np.int32)
    wild = rng.integers(0, F, F // 10)
    faces[wild] = rng.integers(0, V, (F // 10, 3))
    faces[F // 7:F // 7 + 10] = faces[:10]                        # duplicates
    faces[F // 5, 2] = faces[F // 5, 1]                           # degenerate: two equal vertices
    faces[F // 4] = faces[F // 4, 0]                              # degenerate: a point
    return verts, faces


@pytest.mark.parametrize("seed,scale,snap,P,mode", [(21, 1.0, True, 2048, "mixed"), (22, 1.0, False, 3000, "mixed"),
                                                    (23, 0.02, False, 2048, "mixed"), (24, 0.02, True, 1000, "mixed"),
                                                    (25, 0.004, False, 777, "mixed"), (26, 1.0, True, 5000, "mixed"),
                                                    (27, 1.0, True, 2048, "far"), (28, 1.0, False, 2048, "blob"),
                                                    (29, 0.02, True, 3000, "blob")])
def test_culled_point_to_mesh_kernel_bit_exact_on_soups(seed, scale, snap, P, mode):
    """`dsf_mesh_point_dist_forward` (ICPLoss / JointICPLoss) since round 4 skips triangles whose bounding sphere cannot hold a
    minimiser, visits points in Morton-cell order and triangles per wave quarter: distances AND argmin indices must still be
    those of the exhaustive scan in index order (the C oracle), on geometry built to break a cull -- exact ties between
    duplicated / lattice triangles, degenerate triangles, tiny triangles where the reference's epsilon terms enlarge the
    `inside` region, points on vertices / edges / inside faces / far away, clouds above the LDS list capacity (P = 5000: plain
    index ranges), and a labelled (per-part) run over a random partition of the faces.  ``far``: the cloud sits beside the soup;
    ``blob``: the soup is collapsed to 1 % of the cloud's size (what a freshly initialised MANO head predicts) -- every triangle
    is then a near-minimiser of every point, whole blocks survive the cull and the kernel takes its plain loop over them."""
    from dsf_amd import ops
    from oracle import p3d
    rng = np.random.default_rng(seed)
    B, V, F = 3, 600, 1554
    soups = [_mesh_soup(rng, V, F, scale, snap) for _ in range(B)]
    faces = soups[0][1]                                           # one face table for the batch (as MANO's)
    verts = np.stack([s[0] for s in soups])
    pts = (rng.normal(size=(B, P, 3)) * scale).astype(np.float32)
    for b in range(B):
        tri = verts[b][faces]                                     # (F,3,3)
        k = P // 5
        pts[b, :k] = tri[rng.integers(0, F, k), rng.integers(0, 3, k)]                       # exactly on vertices
        e = rng.integers(0, F, k)
        pts[b, k:2 * k] = (tri[e, 0] + tri[e, 1]) / 2                                        # on edges
        f = rng.integers(0, F, k)
        pts[b, 2 * k:3 * k] = tri[f].mean(1) + (0.05 * scale) * rng.normal(size=(k, 3)).astype(np.float32)   # near faces
        g = rng.integers(0, F, k)
        w = rng.uniform(-0.2, 1.2, (k, 2)).astype(np.float32)                                # in the plane, just outside edges
        pts[b, 3 * k:4 * k] = tri[g, 0] + w[:, :1] * (tri[g, 1] - tri[g, 0]) + w[:, 1:] * (tri[g, 2] - tri[g, 0])
    pts[:, -1] = 50.0 * scale                                      # far away
    if mode == "far":
        pts[:, :, 0] += np.float32(6.0 * scale)
    elif mode == "blob":
        verts = (verts * np.float32(0.01)).astype(np.float32)
    T = lambda a: torch.tensor(a, device="cuda")
    first = torch.tensor([0, F], dtype=torch.int32, device="cuda")
    dis, idx = ops.MeshPointDistance.apply(T(verts), T(pts), T(faces), first, None, 1)
    dis, idx = dis.cpu().numpy(), idx.cpu().numpy()
    for b in range(B):
        d_o, i_o = p3d.point_face_dist_forward(pts[b], np.array([0], np.int64), verts[b][faces], np.array([0], np.int64))
        assert np.array_equal(idx[b], i_o.astype(np.int32)), (b, int((idx[b] != i_o).sum()))
        assert np.array_equal(dis[b], d_o), b
    # labelled form: a random partition of the faces into 7 parts, labels 0 (no part) .. 7
    n_parts = 7
    cut = np.sort(rng.choice(np.arange(1, F), n_parts - 1, replace=False))
    pfirst = np.concatenate([[0], cut, [F]]).astype(np.int32)
    seg = rng.integers(0, n_parts + 1, (B, P)).astype(np.int64)
    dis, idx = ops.MeshPointDistance.apply(T(verts), T(pts), T(faces), T(pfirst), T(seg), n_parts)
    dis, idx = dis.cpu().numpy(), idx.cpu().numpy()
    for b in range(B):
        for part in range(n_parts):
            sel = np.nonzero(seg[b] == part + 1)[0]
            if sel.size == 0:
                continue
            f0, f1 = int(pfirst[part]), int(pfirst[part + 1])
            d_o, i_o = p3d.point_face_dist_forward(pts[b][sel], np.array([0], np.int64), verts[b][faces[f0:f1]], np.array([0], np.int64))
            assert np.array_equal(idx[b][sel], (i_o + f0).astype(np.int32)), (b, part)
            assert np.array_equal(dis[b][sel], d_o), (b, part)
        none = seg[b] == 0
        assert (idx[b][none] == -1).all() and (dis[b][none] == 0).all()
