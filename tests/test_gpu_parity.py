"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the golden
vectors made from the reference.  Run on the MI355X box with  pytest -m gpu.

Bars: bit-exact for index / integer outputs (pix_to_face, crop pixels selected, argmin triangle,
part labels, crop bounds); fp32 outputs within 1e-4 absolute (north star) -- the tolerance is
written at each assert.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CAM = (588.03, 587.07, 320.0, 240.0)
IMG = (640, 480)


def T(a, dev="cuda"):
    return torch.tensor(np.asarray(a), device=dev)


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def render():
    from dsf_amd.render_model.mano_layer import Render
    return Render("synthetic", "nyu", CAM, IMG).cuda()


@pytest.fixture(scope="module")
def mano(render):
    return render.mano_layer


def _params(B, seed, pose_scale=0.5):
    rng = np.random.default_rng(seed)
    P = np.zeros((B, 62), dtype=np.float32)
    P[:, :3] = rng.uniform(-np.pi, np.pi, (B, 3))
    P[:, 3:48] = rng.normal(size=(B, 45)) * pose_scale
    P[:, 48:58] = rng.normal(size=(B, 10)) * 0.5
    P[:, 58] = 1.0
    return P


def _centers(B, seed):
    rng = np.random.default_rng(seed)
    c = np.stack([rng.uniform(-40, 40, B), rng.uniform(-40, 40, B), rng.uniform(500, 1200, B)], 1).astype(np.float32)
    cube = np.full((B, 3), 250.0, dtype=np.float32)
    return c, cube


# ------------------------------------------------------------------------------------------------
# K5 MANO
# ------------------------------------------------------------------------------------------------
def test_mano_forward_backward_vs_reference_golden(golden, mano):
    P = T(golden["mano_params"]).requires_grad_(True)
    v, j, Rs = mano.forward(P[:, 48:58], P[:, 3:48], P[:, :3], get_skin=True)
    assert np.abs(N(v) - golden["mano_fwd_verts"]).max() < 1e-5       # metres
    assert np.abs(N(j) - golden["mano_fwd_joints"]).max() < 1e-5
    assert np.abs(N(Rs) - golden["mano_fwd_Rs"]).max() < 1e-5
    v2, j2 = mano.get_mano_vertices(P[:, :3], P[:, 3:48], P[:, 48:58], P[:, 58:62], global_scale=1 / 125)
    assert np.abs(N(v2) - golden["mano_gmv_verts"]).max() < 1e-4      # cube-normalised units (north star 1e-4)
    assert np.abs(N(j2) - golden["mano_gmv_joints"]).max() < 1e-4
    loss = (v2 * T(golden["mano_gw_verts"])).sum() + (j2 * T(golden["mano_gw_joints"])).sum()
    g, = torch.autograd.grad(loss, P)
    ref = golden["mano_grad_params"]
    assert np.abs(N(g) - ref).max() <= 1e-4 * np.abs(ref).max()       # gradients: 1e-4 relative to the largest entry


def test_mano_quaternion_root_vs_golden(golden, mano):
    P = T(golden["mano_quat_params"])
    v, j = mano.get_mano_vertices(P[:, :4], P[:, 4:49], P[:, 49:59], P[:, 59:63])
    assert np.abs(N(v) - golden["mano_quat_verts"]).max() < 1e-2      # millimetres here: 1e-2 mm == 1e-4 of the 250 mm cube/2
    assert np.abs(N(j) - golden["mano_quat_joints"]).max() < 1e-2


def test_mano_vs_oracle_random_batch(oracle_hand, mano):
    from oracle import hand_ref as H
    P = _params(33, 11, pose_scale=1.0)
    P[:, 58] = np.random.default_rng(3).uniform(0.7, 1.3, 33)
    P[:, 59:62] = np.random.default_rng(4).normal(size=(33, 3)) * 0.2
    Pc = torch.tensor(P, requires_grad=True)
    vo, jo = H.mano_vertices(oracle_hand, Pc[:, :3], Pc[:, 3:48], Pc[:, 48:58], Pc[:, 58:62], 1 / 125)
    gw = torch.tensor(np.random.default_rng(5).normal(size=(33, 779, 3)).astype(np.float32))
    go, = torch.autograd.grad((vo * gw).sum() + jo.sum(), Pc)
    Pg = T(P).requires_grad_(True)
    vg, jg = mano.get_mano_vertices(Pg[:, :3], Pg[:, 3:48], Pg[:, 48:58], Pg[:, 58:62], 1 / 125)
    gg, = torch.autograd.grad((vg * gw.cuda()).sum() + jg.sum(), Pg)
    assert np.abs(N(vg) - vo.detach().numpy()).max() < 1e-4
    assert np.abs(N(jg) - jo.detach().numpy()).max() < 1e-4
    assert np.abs(N(gg) - go.numpy()).max() <= 1e-4 * np.abs(go.numpy()).max()
    # reduced PCA (ncomp < 45), as hands_comp[:theta.size(-1)] allows
    vo6, _ = H.mano_vertices(oracle_hand, Pc[:, :3], Pc[:, 3:9], Pc[:, 48:58], Pc[:, 58:62], 1 / 125)
    vg6, _ = mano.get_mano_vertices(Pg[:, :3], Pg[:, 3:9], Pg[:, 48:58], Pg[:, 58:62], 1 / 125)
    assert np.abs(N(vg6) - vo6.detach().numpy()).max() < 1e-4


# ------------------------------------------------------------------------------------------------
# crop set-up, K1 full raster, fused crop renderer
# ------------------------------------------------------------------------------------------------
def test_crop_setup_bit_exact_vs_golden(golden, render):
    from dsf_amd import ops
    c2, M, bounds, minv = ops.crop_setup(T(golden["crop_center3d"]), T(golden["crop_cube"]), render.cam, 128, True)
    assert np.array_equal(N(c2), golden["crop_center2d"])
    assert np.array_equal(N(bounds), golden["crop_bounds"])
    assert np.array_equal(N(M), golden["crop_M"])
    assert np.abs(N(minv) - golden["crop_Minv"]).max() < 1e-3 * np.abs(golden["crop_Minv"]).max()
    # host-visible helper methods agree with the kernel
    xs, xe, ys, ye, zs, ze = render.comToBounds(c2, T(golden["crop_cube"]))
    assert np.array_equal(N(torch.stack([xs, xe, ys, ye], 1)), golden["crop_bounds"])
    assert np.array_equal(N(render.Offset2Trans(xs, xe, ys, ye)), golden["crop_M"])


def _world_verts(mano, B, seed):
    P = T(_params(B, seed))
    c, cube = _centers(B, seed + 1)
    v, _ = mano.get_mano_vertices(P[:, :3], P[:, 3:48], P[:, 48:58], P[:, 58:62], 1 / 125)
    return (v * T(cube).unsqueeze(1) / 2 + T(c).unsqueeze(1)).contiguous(), c, cube


def test_full_raster_bit_exact_vs_oracle(mano, render):
    from oracle import p3d
    verts, _, _ = _world_verts(mano, 3, 21)
    faces = N(mano.faces_i32)
    fr = render.rasterizer(verts)
    pv = p3d.project_verts(N(verts).reshape(-1, 3)).reshape(3, 779, 3)
    fv = pv[:, faces].reshape(-1, 3, 3)
    first = np.arange(3) * faces.shape[0]
    cnt = np.full(3, faces.shape[0])
    p2f, zbuf, bary, dists = p3d.rasterize_meshes(fv, first, cnt, 640)
    assert (p2f >= 0).sum() > 5000
    assert np.array_equal(N(fr.pix_to_face)[..., 0], p2f)                       # bit-exact face indices
    assert np.array_equal(N(fr.zbuf)[..., 0], zbuf)                             # same arithmetic -> same bits
    assert np.array_equal(N(fr.bary_coords)[..., 0, :], bary)
    assert np.abs(N(fr.dists)[..., 0] - dists).max() < 1e-7


def test_full_raster_known_answers(render):
    """Hand-checkable cases on a 16x16 raster (pytorch3d conventions)."""
    from dsf_amd import ops
    S = 16
    def run(fv):
        fv = T(np.asarray(fv, dtype=np.float32).reshape(-1, 3, 3))
        first = torch.zeros(1, dtype=torch.int64, device="cuda")
        cnt = torch.full((1,), fv.shape[0], dtype=torch.int64, device="cuda")
        p2f, z, b, d = ops.RasterizeMeshesFunction.apply(fv, first, cnt, S)
        return N(p2f)[0, :, :, 0], N(z)[0, :, :, 0]
    # axis-aligned right triangle covering NDC x<=0,y<=0 corner region; +x is LEFT, +y is UP in the image
    tri = [[-1.0, -1.0, 5.0], [1.0, -1.0, 5.0], [-1.0, 1.0, 5.0]]
    p2f, z = run([tri])
    ndc = lambda i: -1 + (2 * i + 1) / S
    exp = np.zeros((S, S), dtype=bool)
    for yo in range(S):
        for xo in range(S):
            x, y = ndc(S - 1 - xo), ndc(S - 1 - yo)
            exp[yo, xo] = (x + y) < 0                       # strictly inside the hypotenuse, edges x=-1,y=-1 never hit
    assert np.array_equal(p2f >= 0, exp)
    assert np.all(z[exp] == 5.0) and np.all(z[~exp] == -1.0)
    # two overlapping triangles: nearer z wins regardless of order; exact tie -> lowest face index
    far = [[-1, -1, 9.0], [1, -1, 9.0], [-1, 1, 9.0]]
    near = [[-1, -1, 3.0], [1, -1, 3.0], [-1, 1, 3.0]]
    assert set(np.unique(run([far, near])[0])) == {-1, 1}
    assert set(np.unique(run([near, far])[0])) == {-1, 0}
    assert set(np.unique(run([tri, tri])[0])) == {-1, 0}
    # back-facing triangle is still drawn (no culling); triangle behind the camera is skipped
    back = [tri[0], tri[2], tri[1]]
    assert (run([back])[0] >= 0).sum() == exp.sum()
    behind = [[-1, -1, -5.0], [1, -1, -5.0], [-1, 1, -5.0]]
    assert (run([behind])[0] >= 0).sum() == 0
    # pixel exactly on an edge is uncovered (strict > 0): edge through pixel centres x = ndc(k)
    xk = ndc(5)
    half = [[xk, -1.0, 2.0], [xk, 1.0, 2.0], [-1.0, 0.0, 2.0]]
    p2f, _ = run([half])
    assert not (p2f[:, S - 1 - 5] >= 0).any()


def _oracle_crop(verts, faces, Minv, rowmap, center_z, cube_z):
    """full oracle chain: raster 640 -> bg 0 -> resize rows -> nearest warp -> normalize."""
    from oracle import p3d
    from oracle import image_ref as I
    B = verts.shape[0]
    pv = p3d.project_verts(verts.reshape(-1, 3)).reshape(B, -1, 3)
    fv = pv[:, faces].reshape(-1, 3, 3)
    nf = faces.shape[0]
    p2f, zbuf, _, _ = p3d.rasterize_meshes(fv, np.arange(B) * nf, np.full(B, nf), 640, want_bary=False)
    depth = np.where(zbuf <= 0, 0, zbuf).astype(np.float32)
    loc = np.where(p2f >= 0, p2f - (np.arange(B) * nf)[:, None, None], -1)
    depth480 = depth[:, rowmap, :]
    loc480 = loc[:, rowmap, :]
    src = I.warp_source_index(Minv)
    flat_d = depth480.reshape(B, -1)
    flat_f = loc480.reshape(B, -1)
    ok = src >= 0
    crop = np.where(ok, np.take_along_axis(flat_d, np.maximum(src, 0).reshape(B, -1), 1).reshape(src.shape), 0).astype(np.float32)
    cf = np.where(ok, np.take_along_axis(flat_f, np.maximum(src, 0).reshape(B, -1), 1).reshape(src.shape), -1)
    cf = np.where(crop > 0, cf, -1)
    return I.normalize_depth(crop[:, None], center_z, cube_z), cf, fv, p2f


def test_crop_render_bit_exact_vs_oracle_chain(golden, mano, render):
    from dsf_amd import ops
    B = 6
    verts, c, cube = _world_verts(mano, B, 31)
    c[1] = (-230.0, 170.0, 520.0)                 # crop partly outside the frame -> zero padding
    cube[2] = (200.0, 300.0, 250.0)
    verts = verts.clone()
    P = T(_params(B, 31))
    v, _ = mano.get_mano_vertices(P[:, :3], P[:, 3:48], P[:, 48:58], P[:, 58:62], 1 / 125)
    verts = (v * T(cube).unsqueeze(1) / 2 + T(c).unsqueeze(1)).contiguous()
    c2, M, _, _ = ops.crop_setup(T(c), T(cube), render.cam, 128)
    Minv = torch.inverse(M.cpu())                  # the reference's own call; explicit input to both sides
    img, p2f = ops.RenderCropFunction.apply(verts, mano.faces_i32, Minv.cuda(), render.resize_rowmap, c2[:, 2].contiguous(),
                                            T(cube)[:, 2].contiguous(), render.cam, 640, 128)
    rowmap = golden["resize_rowmap"].astype(np.int64)
    assert np.array_equal(N(render.resize_rowmap), rowmap)
    exp_img, exp_f, _, _ = _oracle_crop(N(verts), N(mano.faces_i32), Minv.numpy(), rowmap, N(c2)[:, 2], cube[:, 2])
    assert (exp_f >= 0).mean() > 0.05
    assert np.array_equal(N(p2f), exp_f)           # bit-exact face index per crop pixel
    assert np.array_equal(N(img), exp_img)         # same pixels selected, same depth bits


def test_crop_render_collapsed_and_oversized_meshes_bit_exact(golden, mano, render):
    """Early-training predictions: a hand a few pixels wide (every face in one or two tiles -> thousands of
    candidates per tile, handled in rounds) and a hand larger than the crop."""
    from dsf_amd import ops
    B = 6
    _, c, cube = _world_verts(mano, B, 61)
    P = T(_params(B, 61))
    v, _ = mano.get_mano_vertices(P[:, :3], P[:, 3:48], P[:, 48:58], P[:, 58:62], 1 / 125)
    scale = T(np.array([0.02, 0.05, 0.004, 0.3, 2.5, 4.0], dtype=np.float32)).view(B, 1, 1)
    v = (v - v.mean(1, keepdim=True)) * scale
    verts = (v * T(cube).unsqueeze(1) / 2 + T(c).unsqueeze(1)).contiguous()
    c2, M, _, _ = ops.crop_setup(T(c), T(cube), render.cam, 128)
    Minv = torch.inverse(M.cpu())
    img, p2f = ops.RenderCropFunction.apply(verts, mano.faces_i32, Minv.cuda(), render.resize_rowmap, c2[:, 2].contiguous(),
                                            T(cube)[:, 2].contiguous(), render.cam, 640, 128)
    rowmap = golden["resize_rowmap"].astype(np.int64)
    exp_img, exp_f, _, _ = _oracle_crop(N(verts), N(mano.faces_i32), Minv.numpy(), rowmap, N(c2)[:, 2], cube[:, 2])
    assert (exp_f[:3] >= 0).sum() > 0 and (exp_f[4:] >= 0).mean() > 0.3
    assert np.array_equal(N(p2f), exp_f)
    assert np.array_equal(N(img), exp_img)


def test_crop_render_matches_golden_warp_maps(golden, mano, render):
    """Feed the reference's own torch.inverse(M) (golden) and check that the pixels the kernel reads
    are the ones the reference's warpPerspective read (golden warp_srcidx)."""
    from dsf_amd import ops
    B = 8
    Minv = T(golden["crop_Minv"])
    c, cube = golden["crop_center3d"], golden["crop_cube"]
    P = T(_params(B, 41))
    v, _ = mano.get_mano_vertices(P[:, :3], P[:, 3:48], P[:, 48:58], P[:, 58:62], 1 / 125)
    verts = (v * T(cube).unsqueeze(1) / 2 + T(c).unsqueeze(1)).contiguous()
    img, p2f = ops.RenderCropFunction.apply(verts, mano.faces_i32, Minv, render.resize_rowmap, None, None, render.cam, 640, 128)
    fr = render.rasterizer(verts)
    z = N(fr.zbuf)[..., 0]
    z = np.where(z <= 0, 0, z)[:, golden["resize_rowmap"].astype(np.int64), :].reshape(B, -1)
    src = golden["warp_srcidx"]
    exp = np.where(src >= 0, np.take_along_axis(z, np.maximum(src, 0).reshape(B, -1), 1).reshape(src.shape), 0)
    assert np.array_equal(N(img)[:, 0], exp.astype(np.float32))


def test_render_api_and_backward_vs_oracle(oracle_hand, mano, render):
    """Render.render end to end: outputs vs oracle pieces, d(img)/d(params) vs the oracle's
    raster backward + autograd chain."""
    from oracle import hand_ref as H, image_ref as I, p3d
    B = 4
    P = _params(B, 51)
    c, cube = _centers(B, 52)
    Pg = T(P).requires_grad_(True)
    img, juvd, jxyz, mxyz = render.render(Pg, T(c), T(cube))
    gw = np.random.default_rng(53).normal(size=(B, 1, 128, 128)).astype(np.float32)
    gg, = torch.autograd.grad((img * T(gw)).sum(), Pg)
    # oracle
    Pc = torch.tensor(P, requires_grad=True)
    vo, jo = H.mano_vertices(oracle_hand, Pc[:, :3], Pc[:, 3:48], Pc[:, 48:58], Pc[:, 58:62], 1 / 125)
    vw = vo * torch.tensor(cube).unsqueeze(1) / 2 + torch.tensor(c).unsqueeze(1)
    c2 = I.project_points(c)
    xs, xe, ys, ye, _, _ = I.crop_bounds(c2, cube)
    M = I.crop_matrix(xs, xe, ys, ye)
    assert np.abs(N(juvd) - I.joint_trans((jo * torch.tensor(cube).unsqueeze(1) / 2 + torch.tensor(c).unsqueeze(1)).detach().numpy(),
                                          M, c2, cube)).max() < 1e-4
    assert np.abs(N(mxyz) - vo.detach().numpy()).max() < 1e-4
    assert np.abs(N(jxyz) - jo.detach().numpy()).max() < 1e-4
    # image: fraction of differing pixels must be tiny (the GPU's torch.inverse may break exact .5 ties differently)
    Minv_gpu = N(torch.linalg.inv_ex(T(M))[0])
    faces = N(mano.faces_i32)
    exp_img, exp_f, fv, p2f_full = _oracle_crop(vw.detach().numpy(), faces, Minv_gpu, N(render.resize_rowmap).astype(np.int64),
                                                c2[:, 2], cube[:, 2])
    # (verts come from two MANO implementations here -> depths agree to 1e-4, not bitwise; only
    #  silhouette pixels may flip)
    assert (np.abs(N(img) - exp_img) > 1e-4).mean() < 2e-3
    # backward: oracle raster backward on the 640 grid, scattered from the crop gradient
    src = I.warp_source_index(Minv_gpu)
    half = cube[:, 2] / 2
    live = (exp_f >= 0) & (np.abs(exp_img[:, 0]) < 1.0)
    gz640 = np.zeros((B, 640, 640), dtype=np.float32)
    rowmap = N(render.resize_rowmap).astype(np.int64)
    for b in range(B):
        ii, jj = np.nonzero(live[b])
        s = src[b, ii, jj]
        np.add.at(gz640[b], (rowmap[s // 640], s % 640), gw[b, 0, ii, jj] / half[b])
    gfv = p3d.rasterize_backward_zbuf(fv, p2f_full, gz640)
    # chain to params through torch autograd of the oracle projection
    hw, hh = 320.0, 240.0
    X, Y, Z = vw.unbind(-1)
    xn = (-X * (CAM[0] / hw)) / Z
    yn = (-Y * (CAM[1] / hh)) / Z
    pv = torch.stack([xn, yn, Z], -1)
    fvt = pv[:, torch.tensor(faces).long()].reshape(-1, 3, 3)
    go, = torch.autograd.grad((fvt * torch.tensor(gfv)).sum(), Pc)
    assert np.abs(N(gg) - go.numpy()).max() <= 2e-3 * np.abs(go.numpy()).max()


def test_full_raster_backward_vs_oracle(mano, render):
    from oracle import p3d
    verts, _, _ = _world_verts(mano, 2, 61)
    verts.requires_grad_(True)
    fr = render.rasterizer(verts)
    gw = torch.randn_like(fr.zbuf) * (fr.pix_to_face >= 0)
    g, = torch.autograd.grad((fr.zbuf * gw).sum(), verts)
    faces = N(mano.faces_i32)
    pv = p3d.project_verts(N(verts).reshape(-1, 3)).reshape(2, 779, 3)
    fv = pv[:, faces].reshape(-1, 3, 3)
    gfv = p3d.rasterize_backward_zbuf(fv, N(fr.pix_to_face)[..., 0], N(gw)[..., 0])
    vt = torch.tensor(N(verts), requires_grad=True)
    X, Y, Z = vt.unbind(-1)
    pvt = torch.stack([(-X * (CAM[0] / 320.0)) / Z, (-Y * (CAM[1] / 240.0)) / Z, Z], -1)
    go, = torch.autograd.grad((pvt[:, torch.tensor(faces).long()].reshape(-1, 3, 3) * torch.tensor(gfv)).sum(), vt)
    assert np.abs(N(g) - go.numpy()).max() <= 1e-3 * np.abs(go.numpy()).max()


# ------------------------------------------------------------------------------------------------
# K3/K4 point-face distance
# ------------------------------------------------------------------------------------------------
def _pcl_near(verts, P, seed):
    rng = np.random.default_rng(seed)
    B, V, _ = verts.shape
    idx = rng.integers(0, V, (B, P))
    return (np.take_along_axis(verts, idx[..., None].repeat(3, -1), 1) + rng.normal(size=(B, P, 3)) * 0.03).astype(np.float32)


def test_point_face_dist_packed_vs_oracle(mano):
    from dsf_amd.metric.meshLoss import point_face_distance
    from oracle import p3d
    P = T(_params(3, 71))
    v, _ = mano.get_mano_vertices(P[:, :3], P[:, 3:48], P[:, 48:58], P[:, 58:62], 1 / 125)
    verts = N(v)
    faces = N(mano.faces_i32).astype(np.int64)
    # ragged batch: different point counts, one EMPTY cloud, a triangle subset for mesh 1
    counts = [700, 0, 1300]
    pts = np.concatenate([_pcl_near(verts[i:i + 1], n, 72 + i)[0] for i, n in enumerate(counts)])
    tris = np.concatenate([verts[0][faces], verts[1][faces[:300]], verts[2][faces]]).astype(np.float32)
    pf = np.array([0, 700, 700])
    tf = np.array([0, 1554, 1854])
    d_o, i_o = p3d.point_face_dist_forward(pts, pf, tris, tf)
    pt = T(pts).requires_grad_(True)
    tt = T(tris).requires_grad_(True)
    d = point_face_distance(pt, T(pf), tt, T(tf), max(counts))
    assert np.array_equal(N(d), d_o)                                   # identical arithmetic -> identical bits
    gw = np.random.default_rng(75).normal(size=d_o.shape).astype(np.float32)
    gp, gt = torch.autograd.grad((d * T(gw)).sum(), (pt, tt))
    gp_o, gt_o = p3d.point_face_dist_backward(pts, tris, i_o, gw)
    assert np.abs(N(gp) - gp_o).max() < 1e-5
    assert np.abs(N(gt) - gt_o).max() < 1e-4 * max(1.0, np.abs(gt_o).max())


def test_icp_losses_vs_oracle(mano):
    from dsf_amd.metric.meshLoss import ICPLoss, JointICPLoss, FingerICPLoss
    from dsf_amd import ops
    from oracle import p3d
    B, Pn = 3, 2048
    P = T(_params(B, 81))
    v, j = mano.get_mano_vertices(P[:, :3], P[:, 3:48], P[:, 48:58], P[:, 58:62], 1 / 125)
    v = v.detach().requires_grad_(True)
    pcl = T(_pcl_near(N(v), Pn, 82))
    faces = N(mano.faces_i32).astype(np.int64)
    # ICPLoss: value, argmin indices (bit-exact) and gradient
    dis, idx = ops.MeshPointDistance.apply(v, pcl, mano.faces_i32, mano.whole_first, None, 1)
    tris = N(v)[:, faces].reshape(-1, 3, 3)
    d_o, i_o = p3d.point_face_dist_forward(N(pcl).reshape(-1, 3), np.arange(B) * Pn, tris, np.arange(B) * 1554)
    assert np.array_equal(N(idx).reshape(-1), i_o - np.repeat(np.arange(B) * 1554, Pn))   # bit-exact argmin
    assert np.array_equal(N(dis).reshape(-1), d_o)
    loss = ICPLoss(v, pcl, mano.faces)
    assert np.allclose(N(loss), d_o.reshape(B, Pn).mean(1), rtol=1e-5, atol=1e-9)
    g, = torch.autograd.grad(loss.sum(), v)
    _, gt_o = p3d.point_face_dist_backward(N(pcl).reshape(-1, 3), tris, i_o, np.full(B * Pn, 1.0 / Pn, dtype=np.float32))
    go = np.zeros((B, 779, 3), dtype=np.float64)
    for b in range(B):
        np.add.at(go[b], faces.reshape(-1), gt_o.reshape(B, 1554 * 3, 3)[b])
    assert np.abs(N(g) - go).max() < 1e-4 * max(1.0, np.abs(go).max())
    # JointICPLoss against the reference's definition evaluated with the oracle (15x replicated form)
    seg = mano.seg_pcl(j, j, v, pcl)
    jl = JointICPLoss(v, pcl, mano.joint_faces, seg)
    exp = np.zeros((B, 15), dtype=np.float64)
    segn = N(seg)
    for k, fk in enumerate(mano.joint_faces):
        fk = N(fk).astype(np.int64)
        tr = N(v)[:, fk].reshape(-1, 3, 3)
        dk, _ = p3d.point_face_dist_forward(N(pcl).reshape(-1, 3), np.arange(B) * Pn, tr, np.arange(B) * fk.shape[0])
        dk = np.where(segn == k + 1, dk.reshape(B, Pn), 0)
        cnt = (dk > 0).sum(1)
        exp[:, k] = np.where(cnt == 0, 0, dk.sum(1) / (cnt + 1e-8))
    assert np.allclose(N(jl), exp, rtol=1e-5, atol=1e-9)
    fseg = torch.clamp((seg + 2) // 3, max=5)
    fl = FingerICPLoss(v, pcl, mano.finger_faces, fseg)
    assert fl.shape == (B, 5) and torch.isfinite(fl).all()
    torch.autograd.grad(jl.sum() + fl.sum(), v)


# ------------------------------------------------------------------------------------------------
# K6/K7 spheres, collision, segmentation
# ------------------------------------------------------------------------------------------------
def test_spheres_collision_seg_vs_golden(golden, mano):
    j = T(golden["mano_gmv_joints"]).requires_grad_(True)
    v = T(golden["mano_gmv_verts"]).requires_grad_(True)
    c, r = mano.get_sphere_radius(j, v)
    assert np.abs(N(c) - golden["sph_c"]).max() < 1e-5
    assert np.abs(N(r) - golden["sph_r"]).max() < 1e-5
    val = mano.calculate_coll(j, v)
    assert abs(float(val.detach()) - float(golden["coll_val"])) < 1e-6
    gj, gv = torch.autograd.grad(val, (j, v))
    assert np.abs(N(gj) - golden["coll_grad_j"]).max() < 1e-5
    assert np.abs(N(gv) - golden["coll_grad_v"]).max() < 1e-5
    j2 = T(golden["coll2_j"]).requires_grad_(True)
    v2 = T(golden["coll2_v"]).requires_grad_(True)
    val2 = mano.calculate_coll(j2, v2)
    assert abs(float(val2.detach()) - float(golden["coll2_val"])) < 1e-6
    gj2, gv2 = torch.autograd.grad(val2, (j2, v2))
    assert np.abs(N(gj2) - golden["coll2_grad_j"]).max() < 1e-5
    assert np.abs(N(gv2) - golden["coll2_grad_v"]).max() < 1e-5
    seg = mano.seg_pcl(T(golden["seg_joints_pix"]), T(golden["mano_gmv_joints"]), T(golden["mano_gmv_verts"]),
                       T(golden["seg_pcl_in"]))
    assert np.array_equal(N(seg).astype(np.int32), golden["seg_out"])          # integer labels: bit-exact


def test_seg_pcl_large_vs_oracle(oracle_hand, mano):
    from oracle import hand_ref as H
    B, Pn = 4, 16384
    P = T(_params(B, 91))
    v, j = mano.get_mano_vertices(P[:, :3], P[:, 3:48], P[:, 48:58], P[:, 58:62], 1 / 125)
    pcl = T((np.random.default_rng(92).normal(size=(B, Pn, 3)) * 0.4).astype(np.float32))
    jp = j + 0.02 * torch.randn_like(j)
    seg = mano.seg_pcl(jp, j, v, pcl)
    exp = H.segment_points(oracle_hand, jp.cpu(), j.cpu(), v.cpu(), pcl.cpu())
    assert (N(seg) != exp.numpy()).sum() <= 2           # sphere centres come from the GPU MANO (1-ulp inputs)


# ------------------------------------------------------------------------------------------------
# K8/K9/K10 image-side ops
# ------------------------------------------------------------------------------------------------
def test_loader_utils_vs_golden(golden):
    from dsf_amd.data.render_loader import loader
    from dsf_amd import ops
    L = loader()
    c3, cube, M, Minv = (T(golden[k]) for k in ("crop_center3d", "crop_cube", "crop_M", "crop_Minv"))
    uvd = T(golden["ld_uvd"]).requires_grad_(True)
    cam = L.cam
    xyz = ops.UvdToXyz.apply(uvd, c3, Minv, cube, cam, 128, True)
    assert np.abs(N(xyz) - golden["ld_uvd2xyznl"]).max() < 1e-5
    assert np.abs(N(ops.UvdToXyz.apply(uvd, c3, Minv, cube, cam, 128, False)) - golden["ld_uvd2xyz"]).max() < 1e-3   # mm
    back = ops.XyzToUvd.apply(T(golden["ld_uvd2xyznl"]), c3, M, cube, cam, 128, False)
    assert np.abs(N(back) - golden["ld_xyznl2uvd"]).max() < 1e-5
    # gradient of the round trip is identity (within fp32)
    rt = L.xyz_nl2uvdnl_tensor(L.uvd_nl2xyznl_tensor(uvd, c3, M, cube), c3, M, cube)
    gw = torch.randn_like(rt)
    g, = torch.autograd.grad((rt * gw).sum(), uvd)
    assert np.abs(N(g) - N(gw)).max() < 2e-3
    # JointTrans
    from dsf_amd.render_model.mano_layer import Render
    jt = ops.XyzToUvd.apply(T(golden["jt_in"]), T(golden["crop_center2d"]), M, cube, cam, 128, True)
    assert np.abs(N(jt) - golden["jt_out"]).max() < 1e-5
    # crop_hand / point image
    full_in = T(golden["ld_crop_hand_full_in"])
    out, xyz_nl, keep = ops.CropHand.apply(full_in, T(golden["ld_crop_joints"][:2]), c3[:2], Minv[:2], cube[:2], cam, 25.0, 20.0, 20.0)
    assert np.array_equal(N(out), golden["ld_crop_hand_full_out"])            # bit-exact: every keep / drop decision and every kept value
    exp_n = golden["ld_xyzimg_n"][:2]
    got = N(xyz_nl).reshape(2, 128, 128, 3).transpose(0, 3, 1, 2)[:, :, ::4, ::4]
    assert np.abs(got - exp_n).max() < 1e-5


def test_img2pcl_vs_golden_and_oracle(golden):
    from dsf_amd import ops
    from dsf_amd.data.render_loader import loader
    from oracle import image_ref as I
    L = loader()
    img = np.ones((3, 1, 128, 128), dtype=np.float32)
    img[:, :, :16] = golden["i2p_img"]
    c3, cube, Minv = golden["crop_center3d"][:3], golden["crop_cube"][:3], golden["crop_Minv"][:3]
    pcl, counts = ops.img2pcl(T(img), T(c3), T(Minv), T(cube), L.cam, 2048)
    assert N(counts).tolist() == [2048, 1024, 0]
    out = golden["i2p_out"]
    srt = lambda a: a[np.lexsort(a.T[::-1])]
    assert np.abs(srt(N(pcl)[0]) - srt(out[0])).max() < 1e-5           # same set (the reference permutes randomly)
    assert np.abs(N(pcl)[1] - out[1]).max() < 1e-5                       # exact multiple: tiled copies, deterministic
    assert not N(pcl)[2].any()                                           # empty -> zeros
    # random branches with an injected draw: oracle = smallest keys among the valid pixels, scan order
    img2 = golden["i2p_rand_img"]
    keys = np.random.default_rng(7).integers(0, 2 ** 31 - 1, (2, 128 * 128)).astype(np.int32)
    keys[0, :4000] = keys[0, 17]                                         # heavy ties at one key value
    pcl2, cnt2 = ops.img2pcl(T(img2), T(c3[:2]), T(Minv[:2]), T(cube[:2]), L.cam, 2048, T(keys))
    cand = I.image_to_points_candidates(img2, c3[:2], Minv[:2], cube[:2])
    for b in range(2):
        valid = np.nonzero(img2[b, 0].reshape(-1) <= 0.99)[0]
        n = valid.shape[0]
        assert int(cnt2[b]) == n
        mult, rem = 2048 // n, 2048 - (2048 // n) * n
        order = np.lexsort((np.arange(n), keys[b, valid].astype(np.uint32)))[:rem]
        pick = np.sort(order)
        exp = np.concatenate([cand[b]] * mult + [cand[b][pick]])
        assert np.abs(N(pcl2)[b] - exp).max() < 1e-5


def test_gfm_vs_golden(golden):
    from dsf_amd.util.generateFeature import GFM
    G = GFM()
    j = T(golden["gfm_joints"]).requires_grad_(True)
    dep = T(golden["gfm_depth"])
    feat = G.joint2offset(j, dep, 0.8, 64)
    assert np.abs(N(feat)[:, :, ::4, ::4] - golden["gfm_feat_sub"]).max() < 1e-5
    assert np.abs(N(feat).astype(np.float64).sum((2, 3)) - golden["gfm_feat_sum"]).max() < 1e-2
    gw = np.random.default_rng(77).normal(size=tuple(feat.shape)).astype(np.float32)
    gj, = torch.autograd.grad((feat * T(gw)).sum(), j)
    ref = golden["gfm_grad_joints"]
    assert np.abs(N(gj) - ref).max() <= 1e-4 * np.abs(ref).max()
    offs = T((N(feat) + np.random.default_rng(78).normal(size=tuple(feat.shape)) * 0.05).astype(np.float32)).requires_grad_(True)
    dec = G.offset2joint_softmax(offs, dep, 0.8)
    assert np.abs(N(dec) - golden["gfm_dec_joints"]).max() < 1e-5
    gwj = np.random.default_rng(79).normal(size=(2, 21, 3)).astype(np.float32)
    go, = torch.autograd.grad((dec * T(gwj)).sum(), offs)
    ref = golden["gfm_dec_grad_sub"]
    assert np.abs(N(go)[:, :, ::4, ::4] - ref).max() <= 1e-4 * np.abs(ref).max()
    assert np.abs(np.abs(N(go)).astype(np.float64).sum((2, 3)) - golden["gfm_dec_grad_abs_sum"]).max() < 1e-3


def test_losses_vs_golden(golden):
    from dsf_amd.metric.losses import SmoothL1Loss
    from dsf_amd.render_model.render_loss import depth_loss
    a = T(golden["sl1_a"]).requires_grad_(True)
    val = SmoothL1Loss()(a, T(golden["sl1_b"]))
    assert abs(float(val.detach()) - float(golden["sl1_val"])) < 1e-8
    g, = torch.autograd.grad(val, a)
    assert np.abs(N(g) - golden["sl1_grad"]).max() < 1e-9
    assert abs(float(depth_loss()(T(golden["dl_a"]), T(golden["dl_b"]))) - float(golden["dl_val"])) < 1e-6


def test_no_cpu_fallback(mano):
    """The product path must fail loudly off-GPU."""
    from dsf_amd.render_model.mano_layer import MANO_SMPL
    cpu = MANO_SMPL("synthetic", "nyu")
    with pytest.raises(RuntimeError):
        cpu.forward(torch.zeros(1, 10), torch.zeros(1, 45), torch.zeros(1, 3), get_skin=True)


# ------------------------------------------------------------------------------------------------
# whole step (BASELINE config 2 at the reference's CPU-runnable size, B=2): HIP path vs CPU oracle
# ------------------------------------------------------------------------------------------------
def test_whole_step_loss_and_grads_vs_oracle_step(mano_dict, render):
    from oracle import step_ref
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config
    torch.manual_seed(3)
    from oracle import nets
    net_cpu = nets.build(MANO_OCR_stage, "ResNet_stage_18", 21, True)          # torch.nn twin (CPU side of the comparison)
    # make the MANO heads produce non-degenerate hands
    with torch.no_grad():
        for head in (net_cpu.mano_regress[2], net_cpu.mano_regress_s2[2]):
            head.bias[58] = 1.0
            head.bias[3:48] = 0.2 * torch.randn(45)
            head.bias[:3] = torch.tensor([0.3, -0.2, 0.1])
    net_gpu = MANO_OCR_stage("ResNet_stage_18", 21, True).cuda()
    net_gpu.load_state_dict(net_cpu.state_dict())
    B = 2
    p, c, cube = synthetic_batch(B, "cpu", seed=5)
    orender = step_ref.OracleRender(mano_dict)
    tgt_c = step_ref.make_targets(orender, p, c, cube)
    loss_c = step_ref.step_loss(net_cpu, orender, tgt_c)
    loss_c.backward()
    tgt_g = {k: v.cuda() for k, v in tgt_c.items()}
    step = RenderSupervisedStep(net_gpu, render, Config)
    loss_g, terms = step.loss(tgt_g)
    loss_g.backward()
    assert abs(float(loss_g.detach()) - float(loss_c.detach())) <= 2e-3 * abs(float(loss_c.detach()))
    # gradients: batch-norm at B=2 amplifies fp32 summation-order noise through ~40 layers, so compare
    # the whole gradient vector (relative L2 error, cosine) and bound the worst single tensor loosely
    num = den = dot = ng = 0.0
    worst = worst_l2 = 0.0
    worst_cos = 1.0
    noise = []
    for (n, pc), (_, pg) in zip(net_cpu.named_parameters(), net_gpu.named_parameters()):
        if pc.grad is None:
            continue
        ref, got = pc.grad.double(), pg.grad.cpu().double()
        num += float(((got - ref) ** 2).sum()); den += float((ref ** 2).sum())
        dot += float((got * ref).sum()); ng += float((got ** 2).sum())
        worst = max(worst, (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6))
        l2, cs = float((got - ref).norm() / ref.norm().clamp_min(1e-30)), float((got * ref).sum() / (got.norm() * ref.norm()).clamp_min(1e-30))
        if l2 > 3e-2 or cs < 0.9995:
            noise.append((n, l2, cs, float(ref.norm())))
        else:
            worst_l2, worst_cos = max(worst_l2, l2), min(worst_cos, cs)
    assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5
    assert dot / (den * ng) ** 0.5 > 0.9995
    # every single tensor: direction and size of its gradient (measured: cosine >= 0.99996, relative L2 <= 0.85 %), and its
    # worst element relative to its largest (measured 6.9 %: one element of a long row through 40 BatchNorm layers at B = 2).
    # The only tensor outside those bars is the bias of the fusion convolution: it feeds a BatchNorm, its true gradient is
    # exactly zero and both fp32 paths return rounding noise (norm 1.5e-7 against 4.9e2 for the whole gradient).
    assert worst_cos > 0.9999 and worst_l2 < 2e-2, (worst_cos, worst_l2)
    assert all(n == "fusion.0.bias" and nr < 1e-6 * den ** 0.5 for n, _, _, nr in noise), noise
    assert worst < 0.15, worst


def test_torch_library_ops_equal_the_function_path_and_trace(mano_dict, render):
    """torch.ops.dsf.* (dsf_amd/torch_ops.py: dispatcher registration of the pytorch3d._C boundary) run the same
    launchers as dsf_amd.ops: bitwise equal outputs and gradients; torch.library.opcheck (schema, fake kernel, autograd
    registration) passes; a function through them traces under make_fx with fake tensors."""
    import dsf_amd.torch_ops  # noqa: F401
    from dsf_amd import ops
    from dsf_amd.train_step import synthetic_batch
    p, c, cube = synthetic_batch(2, "cuda", seed=4)
    mano = render.mano_layer
    with torch.no_grad():
        v, _ = mano.get_mano_vertices(p[:, :3], p[:, 3:48], p[:, 48:58], p[:, 58:62], 1 / 125)
        verts = (v * cube.unsqueeze(1) / 2 + c.unsqueeze(1)).contiguous()
    fv = ops.project_face_verts(verts, mano.faces_i32, render.cam).requires_grad_(True)
    Fn = mano.faces_i32.shape[0]
    first = torch.arange(2, device="cuda") * Fn
    nf = torch.full((2,), Fn, device="cuda", dtype=torch.int64)
    a = ops.RasterizeMeshesFunction.apply(fv, first, nf, 160)
    b = torch.ops.dsf.rasterize_meshes(fv, first, nf, 160)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    gz = torch.randn_like(a[1])
    ga, = torch.autograd.grad((a[1] * gz).sum(), fv)
    gb, = torch.autograd.grad((b[1] * gz).sum(), fv)
    assert torch.allclose(ga, gb, rtol=0, atol=1e-6 * float(ga.abs().max())) and float(ga.abs().sum()) > 0   # (float atomics per face)
    # point-face distance: 2 clouds of 300 points against the 2 meshes' triangles
    tris = verts[:, mano.faces_i32.long()].reshape(-1, 3, 3).contiguous().requires_grad_(True)
    pts = (verts[:, ::2][:, :300] + 3.0 * torch.randn(2, 300, 3, device="cuda")).reshape(-1, 3).contiguous().requires_grad_(True)
    pfirst = torch.tensor([0, 300], device="cuda")
    da = ops.PointFaceDistance.apply(pts, pfirst, tris, first, 300)
    db, ib = torch.ops.dsf.point_face_dist_forward(pts, pfirst, tris, first, 300)
    assert torch.equal(da, db) and ib.dtype == torch.int64 and int(ib.min()) >= 0
    w = torch.randn_like(da)
    g1 = torch.autograd.grad((da * w).sum(), [pts, tris])
    g2 = torch.autograd.grad((db * w).sum(), [pts, tris])
    assert torch.equal(g1[0], g2[0]) and torch.allclose(g1[1], g2[1], rtol=0, atol=1e-6 * float(g1[1].abs().max()))   # (float atomics on shared triangles)
    torch.library.opcheck(torch.ops.dsf.point_face_dist_forward.default, (pts.detach(), pfirst, tris.detach(), first, 300),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    torch.library.opcheck(torch.ops.dsf.rasterize_meshes.default, (fv.detach(), first, nf, 64),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    # traces with fake tensors (what torch.compile / export do first)
    from torch.fx.experimental.proxy_tensor import make_fx

    def icp(points, pf, triangles, tf):
        d, _ = torch.ops.dsf.point_face_dist_forward(points, pf, triangles, tf, 300)
        return d.mean()
    gm = make_fx(icp, tracing_mode="fake")(pts.detach(), pfirst, tris.detach(), first)
    assert any("point_face_dist_forward" in str(n.target) for n in gm.graph.nodes)
    assert torch.equal(gm(pts.detach(), pfirst, tris.detach(), first), db.detach().mean())
