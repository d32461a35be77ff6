"""SURVEY 8(f) row 3 -- the self-intersection volume metric (reference eval_coll.py:348-373, 611-626, 640-674) on the HIP
kernels of csrc/volume.hip against the numpy restatement of trimesh's voxelise + contains (oracle/volume_ref.py):
integer voxel counts per part pair, BIT-EXACT; hand-checkable cube cases; the pitch-2 / pitch-1 protocol."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hand():
    from dsf_amd.render_model.mano_layer import MANO_SMPL
    from dsf_amd.eval_coll import PartModel
    mano = MANO_SMPL("synthetic", "nyu").cuda()
    return mano, PartModel.from_skinning(mano.faces.cpu().numpy().astype(np.int64), mano.weight.cpu().numpy())


def _posed(mano, B, seed, pose_scale):
    from dsf_amd.train_step import synthetic_batch
    p, _, _ = synthetic_batch(B, "cuda", seed=seed)
    p[:, 3:48] *= pose_scale
    with torch.no_grad():
        v, _ = mano.get_mano_vertices(p[:, :3], p[:, 3:48], p[:, 48:58], p[:, 58:62])        # mm
    return v


def test_part_model_is_watertight_and_follows_the_reference_pair_rule(hand):
    _, pm = hand
    assert pm.n_parts == 15 and len(pm.pairs) == 15 * 14 // 2 - 14 and (1, 2) not in pm.pairs and (0, 13) not in pm.pairs
    for f in pm.part_faces:
        use = {}
        for t in f:
            for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
                k = (min(a, b), max(a, b))
                use[k] = use.get(k, 0) + 1
        assert set(use.values()) == {2}                                  # closed: every edge shared by exactly two faces


@pytest.mark.parametrize("pitch", [2, 1])
def test_hand_part_counts_bit_exact_vs_oracle(hand, pitch):
    from oracle import volume_ref as V
    from dsf_amd.eval_coll import self_intersection
    mano, pm = hand
    verts = _posed(mano, 6, 3, 3.0)                                      # exaggerated poses: fingers run into the palm
    vol, pc = self_intersection(pm, verts, pitch, return_pairs=True)
    pc = pc.cpu().numpy()
    any_hit = 0
    for b in range(verts.shape[0]):
        vo, pairs = V.self_intersection(pm.get_part_mesh(verts[b].cpu().numpy()), pitch, per_pair=True)
        exp = np.array([pairs[tuple(pr)] for pr in pm.pairs])
        assert np.array_equal(pc[b], exp), (b, np.nonzero(pc[b] != exp)[0][:5])
        assert float(vol[b]) == vo
        any_hit += int(exp.sum() > 0)
    assert any_hit >= 4
    rest = torch.zeros(2, 62, device="cuda")
    rest[:, 58] = 1
    v0, _ = mano.get_mano_vertices(rest[:, :3], rest[:, 3:48], rest[:, 48:58], rest[:, 58:62])
    assert float(self_intersection(pm, v0, pitch).sum()) == 0.0          # the rest pose does not self-intersect


def test_cube_known_answers_and_errors():
    from oracle import volume_ref as V
    from dsf_amd.eval_coll import PartModel, self_intersection
    A, inner, far, B = V.cube((0.5, 0.5, 0.5), (20.5, 20.5, 20.5)), V.cube((6, 6, 6), (14, 14, 14)), \
        V.cube((40, 40, 40), (48, 48, 48)), V.cube((14, 14, 14), (26, 26, 26))
    for second, expect in ((inner, 98), (B, 4 ** 3 - 3 ** 3)):
        pool = np.concatenate([far[0], A[0], second[0]]).astype(np.float32)
        pm = PartModel(pool.shape[0], [], [far[1], A[1] + 8, second[1] + 16], parent_id=[0, 0, 0])
        assert pm.pairs == [(1, 2)]
        vol, pc = self_intersection(pm, torch.tensor(pool).cuda().unsqueeze(0), 2, return_pairs=True)
        assert int(pc[0, 0]) == expect and float(vol[0]) == expect * 8.0
    with pytest.raises(RuntimeError):
        self_intersection(pm, torch.tensor(pool * 10).cuda().unsqueeze(0), 2, grid=32)   # a 200 mm part needs > 100 cells per axis
    with pytest.raises(ValueError):
        self_intersection(pm, torch.tensor(pool * 1000).cuda().unsqueeze(0), 2, grid=64)  # > 10 subdivision rounds: trimesh raises too
    with pytest.raises(RuntimeError):
        self_intersection(pm, torch.tensor(pool).unsqueeze(0), 2)                         # CPU tensor: no fallback


def test_two_pass_protocol(hand):
    """eval_coll.py:640-674: pitch 2 for all meshes, pitch 1 only for the colliding ones."""
    from dsf_amd.eval_coll import intersection_volumes, self_intersection
    mano, pm = hand
    verts = torch.cat([_posed(mano, 5, 11, 3.0), _posed(mano, 3, 12, 0.0)])
    v2, v1 = intersection_volumes(pm, verts, chunk=4)
    assert v2.shape == (8,) and (v1[v2 == 0] == 0).all() and (v1[v2 > 0] > 0).any()
    again = self_intersection(pm, verts, 1).cpu().numpy()
    assert np.array_equal(v1[v2 > 0], again[v2 > 0])
