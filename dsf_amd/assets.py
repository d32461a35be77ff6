"""Synthetic MANO-shaped hand asset.

The real ``MANO_RIGHT.pkl`` is license-gated and absent, so the hot path is
exercised on a procedural hand that has *exactly* the tensor shapes, dtypes
and hard-coded vertex ids the reference relies on
(/root/reference/render_model/mano_layer.py:98-153):

  f (1538,3) uint32, v_template (778,3), shapedirs (778,3,10),
  J_regressor scipy-sparse (16,778), hands_components (45,45),
  hands_mean (45,), posedirs (778,3,135), kintree_table (2,16),
  weights (778,16)

and honours the ids baked into the reference: wrist loop
(mano_layer.py:103-105,636) and finger-tip vertices 333/444/672/555/744
(mano_layer.py:125-129).

Topology: a two-sheet (palm side / back side) triangulation of a planar
hand silhouette on a 5.5 mm grid, glued along the outline except at the
wrist edge, i.e. a triangulated disk with a 16-vertex boundary loop; with
one extra cell-centre vertex Euler's formula gives exactly V=778, F=1538.

``load_mano_dict(path)`` also reads a real MANO pickle when the user has
one (chumpy objects are unpickled through a shim, chumpy is not needed).
"""
import os
import pickle

import numpy as np

WRIST_LOOP = [121, 214, 215, 279, 239, 234, 92, 38, 122, 118, 117, 119, 120, 108, 79, 78]
TIP_IDS = [333, 444, 672, 555, 744]          # index, middle, pinky, ring, thumb
PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]
CELL = 0.0055                                 # metres

# finger name -> (first column, first row, length in cells); MANO joint order
# is index(1-3) middle(4-6) pinky(7-9) ring(10-12) thumb(13-15).
_FINGERS = {
    "index": (0, 15, 11),
    "middle": (4, 15, 13),
    "pinky": (12, 15, 9),
    "ring": (8, 15, 12),
    "thumb": (-4, 5, 12),
}
_ORDER = ["index", "middle", "pinky", "ring", "thumb"]


def _mask_cells():
    cells = set()
    for r in range(3, 15):                    # palm block
        for c in range(0, 15):
            cells.add((c, r))
    for r, (c0, c1) in enumerate([(3, 11), (2, 12), (1, 13)]):   # wrist taper
        for c in range(c0, c1):
            cells.add((c, r))
    for name in _ORDER:
        c0, r0, ln = _FINGERS[name]
        for r in range(r0, r0 + ln):
            for c in range(c0, c0 + 3):
                cells.add((c, r))
    for r in range(5, 8):                     # thumb bridge
        cells.add((-1, r))
    assert len(cells) == 384, len(cells)
    return cells


def _build_topology():
    cells = _mask_cells()
    gverts = set()
    for (c, r) in cells:
        gverts.update([(c, r), (c + 1, r), (c, r + 1), (c + 1, r + 1)])
    # outline vertices: touched by <4 mask cells
    def ncell(v):
        c, r = v
        return sum(((c + dc, r + dr) in cells) for dc in (-1, 0) for dr in (-1, 0))
    outline = {v for v in gverts if ncell(v) < 4}
    wrist_edge = [(c, 0) for c in range(3, 12)]            # 9 grid verts on the open edge
    wrist_inner = set(wrist_edge[1:-1])                     # duplicated (not glued)
    glued = outline - wrist_inner

    ids = {}
    pos2 = []                                               # (c, r, side) side: 0 glued, +1 top, -1 bottom
    def add(v, side):
        ids[(v, side)] = len(pos2)
        pos2.append((v[0], v[1], side))
    for v in sorted(gverts):
        if v in glued:
            add(v, 0)
        else:
            add(v, +1)
            add(v, -1)
    extra_cell = (7, 8)
    extra_id = len(pos2)
    pos2.append((extra_cell[0] + 0.5, extra_cell[1] + 0.5, +1))
    assert len(pos2) == 778, len(pos2)

    def vid(v, side):
        return ids[(v, 0)] if v in glued else ids[(v, side)]

    faces = []
    for (c, r) in sorted(cells):
        a, b, d, e = (c, r), (c + 1, r), (c + 1, r + 1), (c, r + 1)
        for side in (+1, -1):
            A, B, D, E = (vid(x, side) for x in (a, b, d, e))
            if side == +1 and (c, r) == extra_cell:
                tri = [(A, B, extra_id), (B, D, extra_id), (D, E, extra_id), (E, A, extra_id)]
            elif (a in glued and d in glued) or ((c + r) % 2 and not (b in glued and e in glued)):
                tri = [(A, B, E), (B, D, E)]
            else:
                tri = [(A, B, D), (A, D, E)]
            if side == -1:
                tri = [(t[0], t[2], t[1]) for t in tri]
            faces.extend(tri)
    assert len(faces) == 1538, len(faces)

    # wrist loop in cyclic order: top sheet left->right, bottom sheet right->left
    loop = [vid(v, +1) for v in wrist_edge] + [vid(v, -1) for v in wrist_edge[-2:0:-1]]
    assert len(loop) == 16 and len(set(loop)) == 16
    tips = []
    for name in _ORDER:
        c0, r0, ln = _FINGERS[name]
        tips.append(vid((c0 + 1, r0 + ln), 0))
    return np.array(pos2, dtype=np.float64), np.array(faces, dtype=np.int64), loop, tips, wrist_inner, outline


def _joint_layout():
    """16 joint rest positions in grid units (c, r) plus 5 tip positions."""
    J = np.zeros((16, 2))
    tips = np.zeros((5, 2))
    J[0] = (7.5, 1.5)
    for k, name in enumerate(_ORDER):
        c0, r0, ln = _FINGERS[name]
        cc = c0 + 1.5
        base = r0 if name != "thumb" else r0 + 1.0
        span = ln if name != "thumb" else ln - 1.0
        J[1 + 3 * k] = (cc, base)
        J[2 + 3 * k] = (cc, base + 0.42 * span)
        J[3 + 3 * k] = (cc, base + 0.72 * span)
        tips[k] = (cc - 0.5, r0 + ln)
    return J, tips


def _seg_dist(p, a, b):
    ab = b - a
    t = np.clip(((p - a) @ ab) / max(ab @ ab, 1e-12), 0.0, 1.0)
    return np.linalg.norm(p - (a + t[:, None] * ab), axis=1)


def build_synthetic_mano(seed=0):
    """Returns a dict with the MANO pickle keys (numpy / scipy objects)."""
    import scipy.sparse as sp

    rng = np.random.default_rng(seed)
    pos2, faces, loop, tips, wrist_inner, outline = _build_topology()
    V = pos2.shape[0]
    gc, gr, side = pos2[:, 0], pos2[:, 1], pos2[:, 2]

    # thickness from distance to the glued outline (wrist edge stays open/thick)
    glued_pts = np.array([v for v in outline if v not in wrist_inner], dtype=np.float64)
    d = np.sqrt(((pos2[:, None, :2] - glued_pts[None]) ** 2).sum(-1)).min(1)
    D = 2.0
    prof = np.sqrt(np.clip(1.0 - (1.0 - np.minimum(d, D) / D) ** 2, 0.0, 1.0))
    half_thick = np.where(gr >= 15, 1.45, 2.2)            # fingers thinner than palm (cells)
    half_thick = np.where(gc < -0.5, 1.6, half_thick)     # thumb
    z = side * prof * half_thick

    # metres; x = hand length axis, y = width axis, z = palm normal
    verts = np.stack([(gr - 8.0) * CELL, (gc - 7.5) * CELL, z * CELL], axis=1)

    # ---- permute so the hard-coded ids land on the right vertices ----
    fixed = {}
    for slot, v in zip(WRIST_LOOP, loop):
        fixed[v] = slot
    for slot, v in zip(TIP_IDS, tips):
        assert v not in fixed
        fixed[v] = slot
    used = set(fixed.values())
    free_slots = [s for s in range(V) if s not in used]
    new_of_old = np.empty(V, dtype=np.int64)
    it = iter(free_slots)
    for old in range(V):
        new_of_old[old] = fixed[old] if old in fixed else next(it)
    v_template = np.empty_like(verts)
    v_template[new_of_old] = verts
    grid2 = np.empty((V, 2))
    grid2[new_of_old] = pos2[:, :2]
    f = new_of_old[faces]

    # ---- joints, skinning weights, joint regressor ----
    Jg, tipg = _joint_layout()
    child_pt = np.zeros((16, 2))
    child_pt[0] = (7.5, 9.0)
    for k in range(5):
        child_pt[1 + 3 * k] = Jg[2 + 3 * k]
        child_pt[2 + 3 * k] = Jg[3 + 3 * k]
        child_pt[3 + 3 * k] = tipg[k]
    region = np.zeros(V, dtype=np.int64)                  # 0 palm, k+1 finger k
    for k, name in enumerate(_ORDER):
        c0, r0, ln = _FINGERS[name]
        inside = (grid2[:, 0] >= c0 - 1e-9) & (grid2[:, 0] <= c0 + 3 + 1e-9) & (grid2[:, 1] >= r0 + 0.5)
        region[inside] = k + 1
    W = np.zeros((V, 16))
    sigma = 1.3
    for i in range(16):
        di = _seg_dist(grid2, Jg[i], child_pt[i])
        fing = 0 if i == 0 else (i - 1) // 3 + 1
        if i == 0:
            allowed = np.ones(V, dtype=bool)
            di = np.where(region == 0, np.minimum(di, 1.0), di + 1.5)
        else:
            allowed = (region == fing) | ((region == 0) & (i % 3 == 1))
        W[:, i] = np.where(allowed, np.exp(-(di / sigma) ** 2), 0.0)
    W[W < 0.02 * W.max(1, keepdims=True)] = 0.0
    W /= W.sum(1, keepdims=True)

    Jreg = np.zeros((16, V))
    for i in range(16):
        rad = 3.0 if i == 0 else 1.9
        dj = np.linalg.norm(grid2 - Jg[i], axis=1)
        sel = dj < rad
        assert sel.sum() >= 10, (i, sel.sum())
        w = np.where(sel, 1.0 + 0.5 * np.cos(np.pi * np.minimum(dj / rad, 1.0)), 0.0)
        Jreg[i] = w / w.sum()

    # ---- shape blendshapes ----
    cen = v_template.mean(0)
    rel = v_template - cen
    along = np.clip((grid2[:, 1] - 15.0) / 13.0, 0.0, 1.0)
    sd = np.zeros((V, 3, 10))
    sd[:, :, 0] = 0.020 * rel
    sd[:, 0, 1] = 0.030 * rel[:, 0]
    sd[:, 1, 2] = 0.030 * rel[:, 1]
    sd[:, 2, 3] = 0.060 * rel[:, 2]
    sd[:, 0, 4] = 0.004 * along
    for k in range(5, 10):
        fr = rng.uniform(20.0, 60.0, size=(3, 3))
        ph = rng.uniform(0, 2 * np.pi, size=(3, 3))
        for a in range(3):
            sd[:, a, k] = 0.0008 * np.prod(np.sin(v_template * fr[a] + ph[a]), axis=1)

    # ---- pose correctives: localised around the driving joint ----
    J3 = Jreg @ v_template
    pd = np.zeros((V, 3, 135))
    for j in range(135):
        jt = j // 9 + 1
        dd = np.linalg.norm(v_template - J3[jt], axis=1)
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        pd[:, :, j] = 0.0015 * np.exp(-(dd / 0.012) ** 2)[:, None] * direction[None]

    q, _ = np.linalg.qr(rng.normal(size=(45, 45)))
    comps = q * np.linspace(1.0, 0.25, 45)[:, None]
    mean = np.zeros((15, 3))
    mean[:, 1] = 0.12
    mean[12:, 1] = 0.05
    mean += 0.02 * rng.normal(size=(15, 3))

    kintree = np.stack([np.array(PARENTS, dtype=np.int64) % (2 ** 32), np.arange(16)]).astype(np.uint32)
    return {
        "f": f.astype(np.uint32),
        "v_template": v_template.astype(np.float64),
        "shapedirs": sd,
        "J_regressor": sp.csc_matrix(Jreg),
        "hands_components": comps,
        "hands_mean": mean.reshape(45),
        "posedirs": pd,
        "kintree_table": kintree,
        "weights": W,
    }


class _ChShim:
    """Stand-in for chumpy.Ch while unpickling a real MANO file."""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"x": state})

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self.__dict__.get("x"))
        return a.astype(dtype) if dtype is not None else a


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] == "chumpy":
            return _ChShim
        return super().find_class(module, name)


def load_mano_dict(path):
    """``path``: a MANO pickle, or the literal token ``synthetic`` /
    ``synthetic:<seed>`` (also when it is the tail of ``<dir>/MANO_RIGHT.pkl``
    and that file does not exist)."""
    token = os.path.basename(os.path.dirname(path)) if path.endswith("MANO_RIGHT.pkl") else path
    if os.path.isfile(path):
        with open(path, "rb") as fh:
            return _Unpickler(fh, encoding="latin1").load()
    if str(token).startswith("synthetic") or str(path).startswith("synthetic"):
        tok = token if str(token).startswith("synthetic") else path
        seed = int(tok.split(":")[1]) if ":" in tok else 0
        return build_synthetic_mano(seed)
    raise FileNotFoundError(
        "MANO model not found at %r (pass 'synthetic' to use the procedural hand)" % (path,))


def dump_reference_pickle(path, seed=0):
    """Writes the synthetic asset in the exact pickle layout the reference's
    loader expects (used only by tests/golden/make_golden.py)."""
    with open(path, "wb") as fh:
        pickle.dump(build_synthetic_mano(seed), fh, protocol=2)
