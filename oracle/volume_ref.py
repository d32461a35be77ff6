"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by dsf_amd/).

CPU restatement (numpy float64) of the reference's self-intersection volume metric,
``self_intersection`` (/root/reference/eval_coll.py:611-626; ``intersect_vox`` in util/intersect.py:102-107 is the same
construction for two meshes).  The arithmetic lives in the third-party package **trimesh**, which the reference imports
without pinning a version (it is not in requirements.txt) and which is absent from this image, so its two published
routines are restated here:

  * ``Trimesh.voxelized(pitch)`` -> ``voxel.creation.voxelize_subdivide(mesh, pitch, max_iter=10, edge_factor=2.0)``:
    ``remesh.subdivide_to_size`` splits every face with an edge longer than pitch / 2 at its edge midpoints (4 children)
    until none is left (at most 10 rounds, else ValueError), the vertices of ALL rounds are rounded to the lattice
    (``np.round(v / pitch)``), unique cells are kept, and ``VoxelGrid.points`` are the cell indices times the pitch;
  * ``Trimesh.contains(points)``: ray parity (trimesh casts along a fixed direction, both ways, and re-casts the points
    whose two counts disagree) -- for a closed mesh, the geometric inside test.  Restated as parity along +x with an exact
    float64 edge-function predicate in the (y, z) projection.

PARITY UNPINNED against trimesh itself (absent here); anchored by hand-checkable cases (tests/test_oracle_volume.py:
lattice counts of cube surfaces, nested / overlapping / disjoint cubes).
"""
import numpy as np


def _subdivide(v, f):
    """one midpoint subdivision: every face -> 4 (trimesh.remesh.subdivide); returns (vertices, faces)"""
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    mab, mbc, mca = (a + b) / 2.0, (b + c) / 2.0, (c + a) / 2.0
    n = f.shape[0]
    nv = np.concatenate([a, b, c, mab, mbc, mca])
    ia, ib, ic, iab, ibc, ica = (np.arange(n) + k * n for k in range(6))
    nf = np.concatenate([np.stack([ia, iab, ica], 1), np.stack([iab, ib, ibc], 1), np.stack([ica, ibc, ic], 1),
                         np.stack([iab, ibc, ica], 1)])
    return nv, nf


def subdivide_to_size(vertices, faces, max_edge, max_iter=10):
    """trimesh.remesh.subdivide_to_size: the vertices of every round (the voxeliser uses all of them)."""
    cur_v, cur_f = np.asarray(vertices, dtype=np.float64), np.asarray(faces, dtype=np.int64)
    done = []
    for _ in range(max_iter + 1):
        tri = cur_v[cur_f]
        e = np.sqrt(((tri[:, [1, 2, 0]] - tri) ** 2).sum(-1))
        too_long = (e > max_edge).any(1)
        done.append(cur_v)
        if not too_long.any():
            break
        cur_v, cur_f = _subdivide(cur_v, cur_f[too_long])
    else:
        raise ValueError("max_iter exceeded!")
    return np.concatenate(done)


def voxel_cells(vertices, faces, pitch):
    """unique lattice cells (int64 (n,3)) of trimesh's voxelize_subdivide(mesh, pitch, edge_factor=2).  Only the vertices
    the faces reference count: a Trimesh holds just its own vertices (the reference builds each part from
    ``water_mesh[part_vertex_id_list[i]]``, eval_coll.py:370, and trimesh drops unreferenced ones)."""
    faces = np.asarray(faces, dtype=np.int64)
    used, inv = np.unique(faces, return_inverse=True)
    vertices, faces = np.asarray(vertices, dtype=np.float64)[used], inv.reshape(faces.shape)
    v = subdivide_to_size(vertices, faces, pitch / 2.0)
    return np.unique(np.round(v / pitch).astype(np.int64), axis=0)


def contains(vertices, faces, points):
    """ray parity along +x; exact sign tests in the (y, z) projection, half-open rule on shared edges"""
    v = np.asarray(vertices, dtype=np.float64)
    tri = v[np.asarray(faces, dtype=np.int64)]
    pts = np.asarray(points, dtype=np.float64)
    inside = np.zeros(pts.shape[0], dtype=bool)
    if tri.shape[0] == 0 or pts.shape[0] == 0:
        return inside
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    area = (b[:, 1] - a[:, 1]) * (c[:, 2] - a[:, 2]) - (b[:, 2] - a[:, 2]) * (c[:, 1] - a[:, 1])
    flip = area < 0
    b2, c2 = np.where(flip[:, None], c, b), np.where(flip[:, None], b, c)
    area = np.abs(area)
    ok = area > 0
    a, b2, c2, area = a[ok], b2[ok], c2[ok], area[ok]

    def edge(u, w, q):
        dy, dz = (w[:, 1] - u[:, 1])[None], (w[:, 2] - u[:, 2])[None]
        e = dy * (q[:, 2:3] - u[None, :, 2]) - dz * (q[:, 1:2] - u[None, :, 1])
        return e, (e > 0) | ((e == 0) & ((dz > 0) | ((dz == 0) & (dy < 0))))
    for s in range(0, pts.shape[0], 4096):
        q = pts[s:s + 4096]
        e0, o0 = edge(b2, c2, q)
        e1, o1 = edge(c2, a, q)
        e2, o2 = edge(a, b2, q)
        x = (e0 * a[None, :, 0] + e1 * b2[None, :, 0] + e2 * c2[None, :, 0]) / area[None]
        hit = o0 & o1 & o2 & (x > q[:, 0:1])
        inside[s:s + 4096] = (hit.sum(1) & 1).astype(bool)
    return inside


PARENT_ID = [0, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13]          # eval_coll.py:615


def valid_pairs(n_parts=15, parent_id=PARENT_ID):
    """(s, t) pairs the reference visits (eval_coll.py:617-621): t > s, not parent / child"""
    return [(s, t) for s in range(n_parts) for t in range(s, n_parts)
            if not (s == t or parent_id[s] == t or parent_id[t] == s)]


def self_intersection(part_meshes, pitch=2, parent_id=PARENT_ID, per_pair=False):
    """eval_coll.py:611-626.  part_meshes: list of (vertices (n,3), faces (m,3)).  -> volume (count * pitch^3)"""
    cells = [voxel_cells(v, f, pitch) for v, f in part_meshes]
    total, pairs = 0, {}
    for s, t in valid_pairs(len(part_meshes), parent_id):
        inside = contains(part_meshes[s][0], part_meshes[s][1], cells[t].astype(np.float64) * pitch)
        pairs[(s, t)] = int(inside.sum())
        total += pairs[(s, t)]
    vol = total * float(np.power(pitch, 3))
    return (vol, pairs) if per_pair else vol


def cube(lo, hi):
    """closed axis-aligned box as 12 triangles"""
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    v = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4],
                  [1, 5, 7], [1, 7, 3]])
    return v, f
