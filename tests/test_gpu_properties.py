"""Size-independent properties at BASELINE.json's full sizes (B = 64, 2048-point clouds, 16384-pixel
crops), where the CPU oracle would take minutes: consistency between independent kernels,
determinism, invariances and round trips."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CAM = (588.03, 587.07, 320.0, 240.0)
B = 64


@pytest.fixture(scope="module")
def render():
    from dsf_amd.render_model.mano_layer import Render
    return Render("synthetic", "nyu", CAM, (640, 480)).cuda()


@pytest.fixture(scope="module")
def batch(render):
    from dsf_amd.train_step import synthetic_batch
    p, c, cube = synthetic_batch(B, "cuda", seed=77)
    with torch.no_grad():
        v, j = render.mano_layer.get_mano_vertices(p[:, :3], p[:, 3:48], p[:, 48:58], p[:, 58:62], 1 / 125)
    return p, c, cube, v, j


def test_mano_rigid_invariances(render, batch):
    p, c, cube, v, j = batch
    mano = render.mano_layer
    # a different root rotation moves the hand rigidly: all pairwise vertex distances are preserved
    p2 = p.clone()
    p2[:, :3] = p[:, :3].flip(0)
    v2, j2 = mano.get_mano_vertices(p2[:, :3], p2[:, 3:48], p2[:, 48:58], p2[:, 58:62], 1 / 125)
    idx = torch.randperm(779, device="cuda")[:96]
    d1 = (v[:, idx, None] - v[:, None, idx]).norm(dim=-1)          # (cdist's matmul form is too lossy for this)
    d2 = (v2[:, idx, None] - v2[:, None, idx]).norm(dim=-1)
    assert (d1 - d2).abs().max() < 1e-4
    # cam translation / scale act affinely on the output
    p3 = p.clone()
    p3[:, 58] = 1.25
    p3[:, 59:62] = torch.tensor([0.1, -0.2, 0.3], device="cuda")
    v3, _ = mano.get_mano_vertices(p3[:, :3], p3[:, 3:48], p3[:, 48:58], p3[:, 58:62], 1 / 125)
    assert (v3 - (v * 1.25 + p3[:, None, 59:62])).abs().max() < 1e-5
    # wrist cap vertex is the mean of its ring
    ring = [121, 214, 215, 279, 239, 234, 92, 38, 122, 118, 117, 119, 120, 108, 79, 78]
    assert (v[:, 778] - v[:, ring].mean(1)).abs().max() < 1e-6


def test_crop_kernel_equals_full_raster_plus_gather_and_is_deterministic(render, batch):
    """Two independent kernels (64x16-tile full raster vs fused crop mode) must agree bit for bit on every
    crop pixel; repeated launches must be bit-identical (no atomics on the forward path)."""
    from dsf_amd import ops
    p, c, cube, v, j = batch
    mano = render.mano_layer
    verts = (v * cube.unsqueeze(1) / 2 + c.unsqueeze(1)).contiguous()
    c2, M, _, _ = ops.crop_setup(c, cube, render.cam, 128)
    minv = torch.linalg.inv_ex(M)[0].contiguous()
    img, p2f = ops.RenderCropFunction.apply(verts, mano.faces_i32, minv, render.resize_rowmap, None, None, render.cam, 640, 128)
    img2, p2f2 = ops.RenderCropFunction.apply(verts, mano.faces_i32, minv, render.resize_rowmap, None, None, render.cam, 640, 128)
    assert torch.equal(img, img2) and torch.equal(p2f, p2f2)
    assert (p2f >= 0).float().mean() > 0.03
    fr = render.rasterizer(verts)
    z = fr.zbuf[..., 0]
    z = torch.where(z <= 0, torch.zeros_like(z), z)[:, render.resize_rowmap.long(), :]           # resize rows
    # source pixel of each crop pixel with the same float pipeline (oracle restatement in torch on the GPU is NOT
    # bit-safe, so recover it from the crop kernel's own face index: where the crop pixel is covered, the full
    # raster must hold the same face and depth at SOME pixel of that face -- check through the depth value set)
    for b_ in range(0, B, 8):
        covered = p2f[b_] >= 0
        zfull = fr.zbuf[b_, ..., 0]
        ffull = fr.pix_to_face[b_, ..., 0] - b_ * 1554
        vals = set(zip(ffull[ffull >= 0].tolist(), zfull[ffull >= 0].tolist()))
        crop_pairs = set(zip(p2f[b_][covered].tolist(), img[b_, 0][covered].tolist()))
        assert crop_pairs <= vals                       # every (face, depth) the crop kernel produced exists in the full raster


def test_icp_properties_full_size(render, batch):
    from dsf_amd.metric.meshLoss import ICPLoss, JointICPLoss
    from dsf_amd import ops
    p, c, cube, v, j = batch
    mano = render.mano_layer
    g = torch.Generator(device="cuda").manual_seed(3)
    idx = torch.randint(0, 778, (B, 2048), device="cuda", generator=g)
    on_verts = torch.gather(v, 1, idx[..., None].expand(-1, -1, 3))
    d0, _ = ops.MeshPointDistance.apply(v, on_verts, mano.faces_i32, mano.whole_first, None, 1)
    assert d0.max() < 1e-10 and d0.min() >= 0                                  # points on mesh vertices
    pcl = on_verts + 0.02 * torch.randn(on_verts.shape, device="cuda", generator=g)
    whole, widx = ops.MeshPointDistance.apply(v, pcl, mano.faces_i32, mano.whole_first, None, 1)
    assert (whole >= 0).all() and (widx >= 0).all() and (widx < 1554).all()
    # rigid translation of mesh + cloud leaves every distance unchanged (fp32 noise only)
    t = torch.tensor([0.3, -0.2, 0.1], device="cuda")
    whole_t, _ = ops.MeshPointDistance.apply(v + t, pcl + t, mano.faces_i32, mano.whole_first, None, 1)
    assert (whole - whole_t).abs().max() < 1e-6
    # a point's distance to its own part can never beat its distance to the whole mesh
    seg = mano.seg_pcl(j, j, v, pcl)
    part, pidx = ops.MeshPointDistance.apply(v, pcl, mano.joint_faces_i32, mano.joint_faces_first, seg, 15)
    has = seg > 0
    assert (part[has] >= whole[has] - 1e-9).all()
    assert (part[~has] == 0).all() and (pidx[~has] == -1).all()
    # the selected triangle belongs to the point's part
    first = mano.joint_faces_first.long()
    lo, hi = first[(seg - 1).clamp(min=0)], first[seg.clamp(min=1)]
    assert ((pidx.long() >= lo) & (pidx.long() < hi))[has].all()
    jl = JointICPLoss(v, pcl, mano.joint_faces, seg)
    assert jl.shape == (B, 15) and torch.isfinite(jl).all()
    assert torch.allclose(ICPLoss(v, pcl, mano.faces), whole.mean(-1))


def test_seg_and_collision_full_size(render, batch):
    p, c, cube, v, j = batch
    mano = render.mano_layer
    pcl = 0.5 * torch.randn(B, 16384, 3, device="cuda")
    seg = mano.seg_pcl(j, j, v, pcl)
    assert seg.dtype == torch.int64 and int(seg.min()) >= 0 and int(seg.max()) <= 15
    assert torch.equal(seg, mano.seg_pcl(j, j, v, pcl))
    # a point on the shell of finger sphere k (centre + radius along +x) far from the palm gets bone k's label
    cs, rs = mano.get_sphere_radius(j, v)
    k = 21 + 3 * 11 + 1                                                      # a middle sphere of bone 11
    pt = (cs[:, k] + torch.stack([rs[:, k], torch.zeros_like(rs[:, k]), torch.zeros_like(rs[:, k])], -1)).unsqueeze(1)
    lab = mano.seg_pcl(j, j, v, pt)
    dpalm = ((pt - cs[:, :21]).norm(dim=-1) - rs[:, :21]).abs().min(-1)[0]
    ok = dpalm > 1e-4
    assert (lab[ok, 0] == 12).all()
    coll = mano.calculate_coll(j, v)
    assert torch.isfinite(coll) and coll >= 0
    # spread the fingers apart by scaling joints away from the wrist: the hinge can only shrink with larger gaps
    assert mano.calculate_coll(j * 1.0, v) == coll


def test_img2pcl_and_gfm_round_trip_full_size(render, batch):
    from dsf_amd.data.render_loader import loader
    from dsf_amd.util.generateFeature import GFM
    from dsf_amd import ops
    p, c, cube, v, j = batch
    L = loader()
    img, juvd, jxyz, mesh = render.render(p, c, cube)
    _, M, _, _ = ops.crop_setup(c, cube, render.cam, 128)
    pcl = L.Img2pcl(img, 128, c, M, cube, 2048)
    assert pcl.shape == (B, 2048, 3) and torch.isfinite(pcl).all()
    # every sampled point is a foreground pixel of the image: its depth coordinate matches some pixel depth
    fg = img < 0.99
    for b_ in range(0, B, 16):
        depths = torch.unique(img[b_][fg[b_]])
        assert torch.isin(pcl[b_, :, 2], depths).all()
    # exact-multiple rule: a cloud with n valid pixels has floor(2048/n) whole copies at the front
    n = int(fg[0].sum())
    if 0 < n < 1024:
        assert torch.equal(pcl[0, :n], pcl[0, n:2 * n])
    # GFM: decoding the encoding of joints that lie on the rendered surface returns them
    G = GFM()
    maps = G.joint2offset(juvd, img, 0.8, 64)
    back = G.offset2joint_softmax(maps, img, 0.8)
    seen = (maps[:, 63:] > 0).flatten(2).any(-1)                     # joints with heat support
    assert seen.float().mean() > 0.5
    assert (back - juvd)[seen].abs().max() < 0.06


def test_crop_kernel_is_independent_of_its_launch_shape(render, batch, monkeypatch):
    """Round 5: the crop rasteriser bins faces per tile, BIN_TILES tiles of a workgroup at a time, tiles spread over the
    workgroups by Morton code, heavy tiles shared by four waves.  Every launch shape -- 4, 8, 16, 32, 64, 256 tiles per
    workgroup (one to 32 batches), the Morton and the linear tile order, a batch whose samples are rendered four at a time --
    must give the bits of every other: the image and the face index of each pixel depend on its sample only."""
    from dsf_amd import ops
    p, c, cube, v, j = batch
    mano = render.mano_layer
    verts = (v * cube.unsqueeze(1) / 2 + c.unsqueeze(1)).contiguous()
    c2, M, _, _ = ops.crop_setup(c, cube, render.cam, 128)
    minv = torch.linalg.inv_ex(M)[0].contiguous()
    cz, cbz = c2[:, 2].contiguous(), cube[:, 2].contiguous()

    def run(sl=slice(None)):
        return ops.RenderCropFunction.apply(verts[sl].contiguous(), mano.faces_i32, minv[sl].contiguous(), render.resize_rowmap,
                                            cz[sl].contiguous(), cbz[sl].contiguous(), render.cam, 640, 128)
    monkeypatch.delenv("DSF_CROP_WG_TARGET", raising=False)
    img, p2f = run()
    assert (p2f >= 0).float().mean() > 0.03
    for target in ("64", "128", "256", "512", "1024", "4096", "16384"):       # 1 ... 64 workgroups per sample at B = 64
        monkeypatch.setenv("DSF_CROP_WG_TARGET", target)
        a, f = run()
        assert torch.equal(a, img) and torch.equal(f, p2f), target
    monkeypatch.delenv("DSF_CROP_WG_TARGET")
    for b0 in range(0, B, 4):                                                  # 4 samples per launch: 64 workgroups per sample
        a, f = run(slice(b0, b0 + 4))
        assert torch.equal(a, img[b0:b0 + 4]) and torch.equal(f, p2f[b0:b0 + 4]), b0
    a, f = run(slice(5, 8))                                                    # 3 samples: a workgroup count that is no power of two
    assert torch.equal(a, img[5:8]) and torch.equal(f, p2f[5:8])


def test_crop_sizes_other_than_128_agree_with_the_full_raster(render, batch):
    """64- and 256-pixel crops take the linear tile order (the Morton order is for 16 x 16 tiles): every (face, depth) pair
    the crop kernel produces must exist in the full raster of the same mesh, and a repeat must be bitwise equal."""
    from dsf_amd import ops
    p, c, cube, v, j = batch
    mano = render.mano_layer
    sl = slice(0, 6)
    verts = (v * cube.unsqueeze(1) / 2 + c.unsqueeze(1))[sl].contiguous()
    fr = render.rasterizer(verts)
    for crop in (64, 256):
        c2, M, _, _ = ops.crop_setup(c[sl].contiguous(), cube[sl].contiguous(), render.cam, crop)
        minv = torch.linalg.inv_ex(M)[0].contiguous()
        img, p2f = ops.RenderCropFunction.apply(verts, mano.faces_i32, minv, render.resize_rowmap, None, None, render.cam, 640, crop)
        img2, p2f2 = ops.RenderCropFunction.apply(verts, mano.faces_i32, minv, render.resize_rowmap, None, None, render.cam, 640, crop)
        assert img.shape == (6, 1, crop, crop) and torch.equal(img, img2) and torch.equal(p2f, p2f2)
        assert (p2f >= 0).float().mean() > 0.03
        for b_ in range(6):
            covered = p2f[b_] >= 0
            zfull = fr.zbuf[b_, ..., 0]
            ffull = fr.pix_to_face[b_, ..., 0] - b_ * 1554
            vals = set(zip(ffull[ffull >= 0].tolist(), zfull[ffull >= 0].tolist()))
            assert set(zip(p2f[b_][covered].tolist(), img[b_, 0][covered].tolist())) <= vals, (crop, b_)


def test_labelled_point_to_mesh_is_independent_of_how_groups_are_dealt(render, batch, monkeypatch):
    """Round 5: the workgroups of a (sample, part) deal the part's 64-point groups out between them.  One, three, eight or
    sixteen workgroups per part, labels concentrated in one part, spread evenly, or absent: same distances, same indices."""
    from dsf_amd import ops
    from dsf_amd.metric.meshLoss import _cached_parts
    p, c, cube, v, j = batch
    mano = render.mano_layer
    jx, mesh = render.get_mesh_xyz(p)
    g = torch.Generator(device="cuda").manual_seed(9)
    for P in (2048, 1000, 130):
        idx = torch.randint(0, 779, (B, P), device="cuda", generator=g)
        pcl = (torch.gather(mesh, 1, idx[..., None].expand(-1, -1, 3)) + 0.03 * torch.randn(B, P, 3, device="cuda", generator=g)).contiguous()
        seg = mano.seg_pcl(jx, jx, mesh, pcl)
        cat, first = _cached_parts(list(mano.joint_faces), mesh.device)
        labs = {"seg_pcl": seg, "one part": torch.full_like(seg, 13), "none": torch.zeros_like(seg),
                "uniform": torch.randint(0, 16, seg.shape, device="cuda", generator=g).to(seg.dtype)}
        for name, lab in labs.items():
            monkeypatch.setenv("DSF_PFD_SPLITS", "1")
            d1, i1 = ops.MeshPointDistance.apply(mesh, pcl, cat, first, lab, 15)
            for s_ in ("3", "8", "16"):
                monkeypatch.setenv("DSF_PFD_SPLITS", s_)
                d, i = ops.MeshPointDistance.apply(mesh, pcl, cat, first, lab, 15)
                assert torch.equal(d, d1) and torch.equal(i, i1), (P, name, s_)
            monkeypatch.delenv("DSF_PFD_SPLITS")
            d, i = ops.MeshPointDistance.apply(mesh, pcl, cat, first, lab, 15)
            assert torch.equal(d, d1) and torch.equal(i, i1), (P, name)
            if name == "none":
                assert (i1 == -1).all() and (d1 == 0).all()
            if name == "one part":
                lo, hi = int(first[12]), int(first[13])
                assert ((i1 >= lo) & (i1 < hi)).all()
