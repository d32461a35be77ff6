"""Known-answer tests that anchor the plain-C restatement of pytorch3d==0.4.0's rasteriser and
point-face distance (oracle/p3d_ref.c).  The wheel cannot be imported here, so these hand-checkable
cases are what pins the oracle: "parity unpinned" against the real wheel.  CPU only."""
import numpy as np
import pytest

from oracle import p3d

S = 16
ndc = lambda i: -1 + (2 * i + 1) / S


def _run(fv):
    fv = np.asarray(fv, dtype=np.float32).reshape(-1, 3, 3)
    p2f, z, b, d = p3d.rasterize_meshes(fv, np.zeros(1, dtype=np.int64), np.array([fv.shape[0]]), S)
    return p2f[0], z[0], b[0], d[0]


TRI = [[-1.0, -1.0, 5.0], [1.0, -1.0, 5.0], [-1.0, 1.0, 5.0]]


def test_axis_aligned_triangle_coverage_and_axes():
    p2f, z, b, d = _run([TRI])
    exp = np.zeros((S, S), dtype=bool)
    for yo in range(S):
        for xo in range(S):
            # output (yo, xo) samples NDC of index S-1-o: +x is left, +y is up (A.1)
            exp[yo, xo] = ndc(S - 1 - xo) + ndc(S - 1 - yo) < 0
    assert np.array_equal(p2f >= 0, exp)
    assert np.all(z[exp] == 5.0) and np.all(z[~exp] == -1.0)
    assert np.allclose(b[exp].sum(-1), 1.0, atol=1e-6) and np.all(b[~exp] == -1)
    assert np.all(d[exp] <= 0) and np.all(d[~exp] == -1)
    # the top-left output pixel is NDC (+,+): outside this triangle; bottom-right is inside
    assert p2f[0, 0] == -1 and p2f[S - 1, S - 1] == 0


def test_depth_order_ties_backface_and_behind_camera():
    far = [[-1, -1, 9.0], [1, -1, 9.0], [-1, 1, 9.0]]
    near = [[-1, -1, 3.0], [1, -1, 3.0], [-1, 1, 3.0]]
    assert set(np.unique(_run([far, near])[0])) == {-1, 1}
    assert set(np.unique(_run([near, far])[0])) == {-1, 0}
    assert set(np.unique(_run([TRI, TRI])[0])) == {-1, 0}                 # exact tie -> lowest face index
    back = [TRI[0], TRI[2], TRI[1]]
    assert (_run([back])[0] >= 0).sum() == (_run([TRI])[0] >= 0).sum()     # no back-face culling
    behind = [[-1, -1, -5.0], [1, -1, -5.0], [-1, 1, -5.0]]
    assert (_run([behind])[0] >= 0).sum() == 0


def test_pixel_centre_on_edge_is_not_covered():
    xk = ndc(5)
    half = [[xk, -1.0, 2.0], [xk, 1.0, 2.0], [-1.0, 0.0, 2.0]]
    p2f = _run([half])[0]
    assert not (p2f[:, S - 1 - 5] >= 0).any()          # strict w > 0
    assert (p2f >= 0).any()


def test_zbuf_is_screen_space_linear_interpolation():
    t = [[-1.0, -1.0, 2.0], [1.0, -1.0, 4.0], [-1.0, 1.0, 6.0]]
    p2f, z, b, _ = _run([t])
    m = p2f >= 0
    assert np.allclose(z[m], (b[m] * np.array([2.0, 4.0, 6.0])).sum(-1), atol=1e-6)


def test_raster_backward_matches_finite_differences():
    rng = np.random.default_rng(0)
    fv = np.array([[[-0.8, -0.7, 3.0], [0.9, -0.6, 4.0], [-0.5, 0.85, 5.0]]], dtype=np.float32)
    p2f, z, _, _ = p3d.rasterize_meshes(fv, np.zeros(1, dtype=np.int64), np.array([1]), S)
    gz = (rng.normal(size=z.shape) * (p2f >= 0)).astype(np.float32)
    g = p3d.rasterize_backward_zbuf(fv, p2f, gz)
    eps = 1e-3
    for idx in [(0, 0, 0), (0, 1, 1), (0, 2, 2), (0, 0, 2), (0, 2, 0)]:
        fp, fm = fv.copy(), fv.copy()
        fp[idx] += eps
        fm[idx] -= eps
        pp, zp, _, _ = p3d.rasterize_meshes(fp, np.zeros(1, dtype=np.int64), np.array([1]), S)
        pm, zm, _, _ = p3d.rasterize_meshes(fm, np.zeros(1, dtype=np.int64), np.array([1]), S)
        same = (pp >= 0) & (pm >= 0) & (p2f >= 0)        # coverage is piecewise constant; compare where it did not flip
        fd = ((zp - zm) * gz * same).sum() / (2 * eps)
        an_mask_corr = g[idx]
        assert abs(fd - an_mask_corr) < 5e-2 * max(1.0, abs(fd)), (idx, fd, an_mask_corr)


def test_project_verts_matches_pinhole():
    v = np.array([[10.0, -20.0, 800.0], [0.0, 0.0, 500.0]], dtype=np.float32)
    out = p3d.project_verts(v)
    u = v[:, 0] * 588.03 / v[:, 2] + 320
    w = v[:, 1] * 587.07 / v[:, 2] + 240
    assert np.allclose(out[:, 0], -(u - 320) / 320, atol=1e-6)          # A.1: x_ndc = -(u-320)/320
    assert np.allclose(out[:, 1], -(w - 240) / 240, atol=1e-6)
    assert np.array_equal(out[:, 2], v[:, 2])                            # z is view-space depth in mm


# ---- point-face distance (A.4) ----
T0 = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], dtype=np.float32)


def _pfd(pts, tris=T0, pf=(0,), tf=(0,)):
    return p3d.point_face_dist_forward(np.asarray(pts, dtype=np.float32), np.array(pf), tris, np.array(tf))


def test_point_above_interior_edge_vertex():
    d, i = _pfd([[0.25, 0.25, 2.0], [0.5, -1.0, 0.0], [-3.0, -4.0, 0.0], [2.0, 2.0, 1.0]])
    assert np.allclose(d, [4.0, 1.0, 25.0, (2 - 0.5) ** 2 * 2 + 1.0], atol=1e-6)
    assert np.all(i == 0)


def test_argmin_ties_take_lowest_index_and_ragged_batches():
    tris = np.concatenate([T0, T0, T0 + np.float32(5)]).astype(np.float32)
    d, i = _pfd([[0.2, 0.2, 1.0]], tris[:2])
    assert i[0] == 0 and d[0] == 1.0
    # batch element 0: 2 points vs tris[0:2]; element 1: EMPTY cloud; element 2: 1 point vs tris[2:3]
    pts = [[0.2, 0.2, 1.0], [0.1, 0.1, 0.5], [5.2, 5.2, 7.0]]
    d, i = _pfd(pts, tris, pf=(0, 2, 2), tf=(0, 2, 2))
    assert np.allclose(d, [1.0, 0.25, 4.0], atol=1e-6) and i.tolist() == [0, 0, 2]


def test_degenerate_triangle_falls_back_to_segments():
    tri = np.array([[[0, 0, 0], [1, 0, 0], [2, 0, 0]]], dtype=np.float32)       # zero area
    d, _ = _pfd([[0.5, 1.0, 0.0]], tri)
    assert np.allclose(d, [1.0], atol=1e-6)


def test_pfd_backward_matches_finite_differences():
    rng = np.random.default_rng(1)
    tris = rng.normal(size=(6, 3, 3)).astype(np.float32)
    pts = rng.normal(size=(40, 3)).astype(np.float32)
    d, idx = _pfd(pts, tris)
    gw = rng.normal(size=d.shape).astype(np.float32)
    gp, gt = p3d.point_face_dist_backward(pts, tris, idx, gw)
    eps = 1e-3
    for trial in range(12):
        if trial % 2 == 0:
            k = tuple(rng.integers(0, s) for s in pts.shape)
            a, b = pts.copy(), pts.copy()
            a[k] += eps; b[k] -= eps
            fd = ((_pfd(a, tris)[0] - _pfd(b, tris)[0]) * gw).sum() / (2 * eps)
            assert abs(fd - gp[k]) < 2e-2 * max(1.0, abs(fd))
        else:
            k = tuple(rng.integers(0, s) for s in tris.shape)
            a, b = tris.copy(), tris.copy()
            a[k] += eps; b[k] -= eps
            fd = ((_pfd(pts, a)[0] - _pfd(pts, b)[0]) * gw).sum() / (2 * eps)
            assert abs(fd - gt[k]) < 2e-2 * max(1.0, abs(fd))


# ------------------------------------------------------------------------------------------------------
# independent cross-check: the float32 restatement against the same rule in float64 + exact rational signs
# ------------------------------------------------------------------------------------------------------
def _golden_inputs():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_p3d as mg
    return mg.fixed_inputs()


def test_restatement_disagrees_with_exact_evaluation_only_at_rounding_level():
    """oracle/p3d_ref.c (float32, one IEEE op per expression) against oracle/p3d_exact.py (same published rule, float64
    with exact rational arithmetic for every sign float64 cannot settle): posed hand meshes at 640^2 and adversarial
    triangle soups whose vertices sit ON pixel centres.  Every pixel whose face differs must be one float32 rounding
    away from an edge (or an exact depth tie); zero unexplained pixels; depths agree to 5e-5 relative."""
    from oracle import p3d_exact as X
    inp, _ = _golden_inputs()
    faces, world = inp["faces"], inp["world"]
    nf = faces.shape[0]
    pv = p3d.project_verts(world.reshape(-1, 3)).reshape(world.shape)
    fv = pv[:, faces].reshape(-1, 3, 3)
    B = world.shape[0]
    p2f, z, _, _ = p3d.rasterize_meshes(fv, np.arange(B) * nf, np.full(B, nf), 640, want_bary=False)
    tot = {"disagree": 0, "edge": 0, "ztie": 0, "unexplained": 0, "covered": 0}
    for b in range(B):
        loc = np.where(p2f[b] >= 0, p2f[b] - b * nf, -1)
        r = X.explain(fv[b * nf:(b + 1) * nf], loc, z[b], 640)
        assert r["unexplained"] == 0 and r["max_z_rel_err"] < 5e-5, r          # thin faces: e_i / area cancels in float32
        for k in tot:
            tot[k] += r[k]
    assert tot["covered"] > 4000
    S = int(inp["soup_size"])
    soup_tot = {"disagree": 0, "edge": 0, "ztie": 0, "unexplained": 0, "covered": 0}
    for soup in inp["soups"]:
        p2f, z, _, _ = p3d.rasterize_meshes(soup, np.zeros(1, dtype=np.int64), np.array([soup.shape[0]]), S, want_bary=False)
        r = X.explain(soup, p2f[0], z[0], S)
        assert r["unexplained"] == 0, r
        for k in soup_tot:
            soup_tot[k] += r[k]
    assert soup_tot["covered"] > 3 * S * S // 2
    print("hands:", tot, "soups:", soup_tot)


def test_exact_evaluation_reproduces_the_known_answers():
    """The exact evaluator itself on the hand-checkable cases above (it must not share a blind spot with the C code)."""
    from oracle import p3d_exact as X
    fv = np.asarray([TRI], dtype=np.float32)
    p2f, z, _ = X.rasterize_exact(fv, S)
    exp = np.zeros((S, S), dtype=bool)
    for yo in range(S):
        for xo in range(S):
            exp[yo, xo] = ndc(S - 1 - xo) + ndc(S - 1 - yo) < 0
    assert np.array_equal(p2f >= 0, exp) and np.allclose(z[exp], 5.0, rtol=1e-7)   # (weights sum to area / (area + 1e-8))
    xk = ndc(5)
    half = np.asarray([[[xk, -1.0, 2.0], [xk, 1.0, 2.0], [-1.0, 0.0, 2.0]]], dtype=np.float32)
    assert not (X.rasterize_exact(half, S)[0][:, S - 1 - 5] >= 0).any()            # strict w > 0 on the edge
    both = np.asarray([TRI, TRI], dtype=np.float32)
    assert set(np.unique(X.rasterize_exact(both, S)[0])) == {-1, 0}                 # exact tie -> lowest face


def test_oracle_against_pytorch3d_golden():
    """Consumes tests/golden/p3d_golden.npz when someone has produced it on a machine WITH pytorch3d 0.4.0
    (tests/golden/make_golden_p3d.py); skipped otherwise -- until then the p3d oracle is parity-unpinned."""
    import os
    from oracle import p3d_exact as X
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "p3d_golden.npz")
    if not os.path.isfile(path):
        pytest.skip("no pytorch3d golden file (pytorch3d 0.4.0 is not installable here): p3d oracle stays unpinned")
    g = np.load(path)
    inp, gz = _golden_inputs()
    assert np.array_equal(g["world"], inp["world"]) and np.array_equal(g["soups"], inp["soups"])
    faces, world = inp["faces"], inp["world"]
    B, nf = world.shape[0], faces.shape[0]
    # A.1 camera transform
    pv = p3d.project_verts(world.reshape(-1, 3)).reshape(world.shape)
    assert np.abs(pv - g["mr_screen_verts"]).max() <= 1e-6 * max(1.0, np.abs(pv).max())
    fv = g["mr_screen_verts"][:, faces].reshape(-1, 3, 3)                            # the wheel's own projected vertices
    p2f, z, bary, _ = p3d.rasterize_meshes(fv, np.arange(B) * nf, np.full(B, nf), 640)
    for name in ("naive", "binned"):
        ref_p2f, ref_z = g["hand_%s_p2f" % name][..., 0], g["hand_%s_zbuf" % name][..., 0]
        diff = ref_p2f != p2f
        for b in range(B):                                                          # every differing pixel is rounding-level
            if diff[b].any():
                loc = np.where(ref_p2f[b] >= 0, ref_p2f[b] - b * nf, -1)
                assert X.explain(fv[b * nf:(b + 1) * nf], loc, ref_z[b], 640)["unexplained"] == 0
        same = ~diff & (p2f >= 0)
        assert np.abs(z[same] - ref_z[same]).max() <= 1e-5 * np.abs(ref_z[same]).max()
        assert diff.mean() < 1e-4
    assert np.array_equal(g["mr_pix_to_face"][..., 0], g["hand_binned_p2f"][..., 0])
    # backward
    gzb = p3d.rasterize_backward_zbuf(fv, g["hand_binned_p2f"][..., 0], gz)
    ref = g["hand_grad_face_verts"]
    assert np.abs(gzb - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max())
    S = int(inp["soup_size"])
    for k, soup in enumerate(inp["soups"]):
        p2f, z, _, _ = p3d.rasterize_meshes(soup, np.zeros(1, dtype=np.int64), np.array([soup.shape[0]]), S, want_bary=False)
        for name in ("naive", "binned"):
            ref_p2f, ref_z = g["soup%d_%s_p2f" % (k, name)][0, ..., 0], g["soup%d_%s_zbuf" % (k, name)][0, ..., 0]
            if name == "binned":
                # the coarse stage drops faces with zmin < kEpsilon (camera-plane crossers), the naive path keeps the part
                # in front: only such faces may differ
                crossing = (soup[..., 2].min(1) < 1e-8) & (soup[..., 2].max(1) >= 0)
                bad = (ref_p2f != p2f[0]) & ~np.isin(p2f[0], np.nonzero(crossing)[0])
            else:
                bad = ref_p2f != p2f[0]
            if bad.any():
                assert X.explain(soup, ref_p2f, ref_z, S)["unexplained"] <= int((ref_p2f != p2f[0]).sum() - bad.sum())
    # point-face distance
    tris = world[:, faces].reshape(-1, 3, 3)
    P = inp["points"].shape[1]
    d, idx = p3d.point_face_dist_forward(inp["points"].reshape(-1, 3), np.arange(B) * P, tris, np.arange(B) * nf)
    assert np.allclose(d, g["pfd_dists"], rtol=1e-4, atol=1e-6)
    moved = idx != g["pfd_idxs"]
    assert moved.mean() < 1e-2                                # ties between triangles sharing the nearest edge / vertex
    gp, gt = p3d.point_face_dist_backward(inp["points"].reshape(-1, 3), tris, g["pfd_idxs"], np.ones_like(d))
    assert np.abs(gp - g["pfd_grad_points"]).max() <= 1e-3 * max(1.0, np.abs(gp).max())


def test_point_face_restatement_vs_float64_evaluation():
    """oracle/p3d_ref.c's point_face_dist (float32, one IEEE op per expression) against oracle/pfd_exact.py (the published
    rule written independently in vectorised float64): posed hands in millimetres AND in cube-normalised units (where the
    rule's 1e-8 regularisers are NOT negligible: denom ~ 1e-6), clouds on / near / far from the surface, and lattice
    triangle soups with duplicated and degenerate triangles.  Every distance agrees to float32 rounding and every chosen
    triangle is a float64 minimiser to that bar; index disagreements are ties."""
    from oracle import pfd_exact as X
    inp, _ = _golden_inputs()
    faces, world = inp["faces"], inp["world"]
    rng = np.random.default_rng(7)
    tot = {"points": 0, "ties": 0, "dist": 0, "argmin": 0}
    for b in range(min(world.shape[0], 3)):
        for scale in (1.0, 1.0 / 125.0):                                   # mm, and the trainer's cube-normalised units
            v = (world[b] - world[b].mean(0)) * scale
            tris = v[faces].astype(np.float32)
            on = tris[rng.integers(0, tris.shape[0], 400)].mean(1)                     # centroids: exactly on the surface
            near = v[rng.integers(0, v.shape[0], 600)] + rng.normal(0, 4.0 * scale, (600, 3))
            far = rng.normal(0, 150.0 * scale, (200, 3))
            edge = 0.5 * (tris[rng.integers(0, tris.shape[0], 300), 0] + tris[rng.integers(0, tris.shape[0], 300), 1])
            pts = np.concatenate([on, near, far, edge]).astype(np.float32)
            d, i = p3d.point_face_dist_forward(pts, np.zeros(1, np.int64), tris, np.zeros(1, np.int64))
            r = X.explain(pts, tris, d, i)
            assert r["dist"] == 0 and r["argmin"] == 0, (b, scale, r)
            for k in tot:
                tot[k] += r[k]
    # lattice soup: shared vertices / edges everywhere, duplicated and zero-area triangles
    g = rng.integers(-3, 4, (120, 3, 3)).astype(np.float32)
    g[10] = g[11]                                                          # duplicate
    g[20, 2] = g[20, 1]                                                    # degenerate (two equal vertices)
    g[21] = g[21, 0]                                                       # a point
    pts = rng.integers(-8, 9, (500, 3)).astype(np.float32) * 0.5
    d, i = p3d.point_face_dist_forward(pts, np.zeros(1, np.int64), g, np.zeros(1, np.int64))
    r = X.explain(pts, g, d, i)
    assert r["dist"] == 0 and r["argmin"] == 0, r
    # exact ties resolve to the lowest index in the restatement (the float64 argmin does too: same values)
    assert r["ties"] >= 0 and tot["points"] == 9000 and tot["ties"] > 0
    print("hands:", tot, "soup:", r)
