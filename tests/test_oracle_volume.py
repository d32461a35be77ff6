"""Hand-checkable anchors of the numpy restatement of the reference's self-intersection volume (oracle/volume_ref.py;
reference eval_coll.py:611-626).  trimesh is absent here, so these known answers are what pins the oracle."""
import numpy as np

from oracle import volume_ref as V


def test_surface_voxels_of_a_lattice_aligned_cube():
    """A cube with corners on the lattice: its surface cells are exactly the lattice points of its surface."""
    v, f = V.cube((0, 0, 0), (8, 8, 8))
    cells = V.voxel_cells(v, f, 2.0)
    k = 8 // 2 + 1
    assert cells.shape[0] == k ** 3 - (k - 2) ** 3                     # 5^3 - 3^3 = 98 surface points
    assert cells.min() == 0 and cells.max() == 4
    on_surface = ((cells == 0) | (cells == 4)).any(1)
    assert on_surface.all()
    # pitch 1: 9^3 - 7^3
    assert V.voxel_cells(v, f, 1.0).shape[0] == 9 ** 3 - 7 ** 3


def test_contains_is_the_geometric_inside_test():
    v, f = V.cube((0.25, 0.25, 0.25), (10.25, 10.25, 10.25))
    g = np.arange(-2, 14, dtype=np.float64)
    pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    exp = ((pts > 0.25) & (pts < 10.25)).all(1)
    assert np.array_equal(V.contains(v, f, pts), exp)
    # a point whose ray passes exactly through an edge / a vertex of the far face is still counted once
    v2, f2 = V.cube((0, 0, 0), (4, 4, 4))
    assert V.contains(v2, f2, np.array([[1.0, 2.0, 2.0], [1.0, 0.0, 2.0], [5.0, 2.0, 2.0], [-1.0, 2.0, 2.0]])).tolist() == \
        [True, V.contains(v2, f2, np.array([[1.0, 0.0, 2.0]]))[0], False, False]


def test_cube_pairs_known_volumes():
    pitch = 2.0
    A = V.cube((0.5, 0.5, 0.5), (20.5, 20.5, 20.5))
    inner = V.cube((6, 6, 6), (14, 14, 14))                            # nested: all 98 surface cells of `inner` are inside A
    outer_far = V.cube((40, 40, 40), (48, 48, 48))                     # disjoint
    parents = [0, 0, 0]                                                # part 0 is everyone's parent: those pairs are skipped
    vol, pairs = V.self_intersection([outer_far, A, inner], pitch, parent_id=parents, per_pair=True)
    assert pairs == {(1, 2): 98} and vol == 98 * 8.0
    # overlapping corner: B = [14, 26]^3, surface cells with all coordinates in (0.5, 20.5) -> lattice points of B's surface
    # (even coordinates 14..26) with every coordinate <= 20: on B's surface means some coordinate == 14 (26 is outside A)
    B = V.cube((14, 14, 14), (26, 26, 26))
    vol, pairs = V.self_intersection([outer_far, A, B], pitch, parent_id=parents, per_pair=True)
    inside_vals = [14, 16, 18, 20]
    exp = sum(1 for x in inside_vals for y in inside_vals for z in inside_vals if 14 in (x, y, z))
    assert pairs[(1, 2)] == exp == 4 ** 3 - 3 ** 3 and vol == exp * 8.0
    # the reference's pair rule: t >= s only, parents / children skipped
    assert V.valid_pairs(15)[0] == (0, 2) and (1, 2) not in V.valid_pairs(15) and (1, 4) in V.valid_pairs(15)
    assert len(V.valid_pairs(15)) == 15 * 14 // 2 - 14


def test_subdivision_depth_is_uniform_per_face_and_matches_the_lattice_form():
    """The closed form the HIP kernel uses -- barycentric lattice of depth n = min{n : longest edge / 2^n <= pitch / 2} --
    gives exactly the vertex set of the recursive midpoint subdivision (float32-valued inputs: all arithmetic exact)."""
    rng = np.random.default_rng(5)
    tri = rng.uniform(-20, 20, (6, 3, 3)).astype(np.float32).astype(np.float64)
    for pitch in (1.0, 2.0):
        for t in tri:
            got = {tuple(r) for r in V.voxel_cells(t, np.array([[0, 1, 2]]), pitch)}
            longest = max(np.linalg.norm(t[i] - t[(i + 1) % 3]) for i in range(3))
            n = 0
            while longest / (1 << n) > pitch / 2:
                n += 1
            N = 1 << n
            pts = [(i * t[0] + j * t[1] + (N - i - j) * t[2]) / N for i in range(N + 1) for j in range(N + 1 - i)]
            exp = {tuple(r) for r in np.round(np.array(pts) / pitch).astype(np.int64)}
            assert got == exp
