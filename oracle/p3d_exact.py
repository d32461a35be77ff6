"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by dsf_amd/).

An INDEPENDENT evaluation of pytorch3d 0.4.0's rasterisation rule used to cross-check the
float32 restatement in oracle/p3d_ref.c: the same published rule -- strict barycentric inside test through the edge
functions and ``area + 1e-8``, screen-space-linear z, nearest z wins, lowest face index on exact ties -- evaluated in
float64 with exact rational arithmetic (``fractions.Fraction``) for every sign decision that float64 cannot settle.
The inputs are the float32 vertex values taken as exact numbers, so the result is what the rule gives with NO rounding.

Where the float32 restatement and this evaluation disagree on a pixel's face, ``explain()`` classifies the pixel:
  * ``edge``: the pixel centre is within float32 rounding distance of an edge of one of the two faces (the sign of an
    edge function is decided by rounding);
  * ``ztie``: both faces cover the pixel and their interpolated depths agree to float32 rounding;
  * ``unexplained``: anything else -- a restatement error.  tests/test_oracle_p3d.py requires zero of these.
This does not replace the wheel (parity against pytorch3d itself stays unpinned here); it
shows that the restatement has no error above floating-point rounding against the rule it restates, and it is the
yardstick against which a CUDA build's FMA contraction (which moves the same borderline pixels) can be judged.
"""
from fractions import Fraction

import numpy as np

K_EPS = float(np.float32(1e-8))
U = 2.0 ** -24            # float32 unit roundoff


def pix_to_ndc(i, S):
    """PixToNdc as pytorch3d evaluates it -- in float32 (the pixel centre the rule tests IS that rounded value) --
    returned as float64."""
    i = np.asarray(i)
    return (np.float32(-1.0) + (2 * i + 1).astype(np.float32) / np.float32(S)).astype(np.float64)


def _edge64(px, py, ax, ay, bx, by):
    t1 = (px - ax) * (by - ay)
    t2 = (py - ay) * (bx - ax)
    return t1 - t2, np.abs(t1) + np.abs(t2)


def _edge_exact(px, py, ax, ay, bx, by):
    Fr = Fraction
    return (Fr(px) - Fr(ax)) * (Fr(by) - Fr(ay)) - (Fr(py) - Fr(ay)) * (Fr(bx) - Fr(ax))


def rasterize_exact(face_verts, S):
    """One mesh: face_verts (F,3,3) float32 (x_ndc, y_ndc, z) -> pix_to_face (S,S) int64, zbuf (S,S) float64,
    margin (S,S) float64 = the smallest |edge function| / (rounding scale) over the winning face's three edges
    (< ~8 means float32 could have decided that pixel either way)."""
    fv = np.asarray(face_verts, dtype=np.float32).astype(np.float64)
    p2f = np.full((S, S), -1, dtype=np.int64)
    zbuf = np.full((S, S), np.inf)
    margin = np.full((S, S), np.inf)
    centres = pix_to_ndc(np.arange(S), S)
    for f in range(fv.shape[0]):
        (x0, y0, z0), (x1, y1, z1), (x2, y2, z2) = fv[f]
        if max(z0, z1, z2) < 0:
            continue
        fa, fa_scale = _edge64(x0, y0, x1, y1, x2, y2)
        if abs(fa) <= K_EPS + 4 * U * fa_scale:
            fa = float(_edge_exact(x0, y0, x1, y1, x2, y2))
            if abs(fa) <= K_EPS:
                continue
        xs = np.nonzero((centres >= min(x0, x1, x2)) & (centres <= max(x0, x1, x2)))[0]
        ys = np.nonzero((centres >= min(y0, y1, y2)) & (centres <= max(y0, y1, y2)))[0]
        if xs.size == 0 or ys.size == 0:
            continue
        px, py = np.meshgrid(centres[xs], centres[ys])
        area, _ = _edge64(x2, y2, x0, y0, x1, y1)
        area = area + K_EPS                                   # exact enough: |area| >> 1e-24 whenever the face survives
        es, scales = [], []
        for (ax, ay, bx, by) in ((x1, y1, x2, y2), (x2, y2, x0, y0), (x0, y0, x1, y1)):
            e, sc = _edge64(px, py, ax, ay, bx, by)
            # float64 cannot settle the sign: redo those pixels in exact rational arithmetic
            doubt = np.abs(e) <= 8 * 2.0 ** -52 * sc
            for (iy, ix) in zip(*np.nonzero(doubt)):
                e[iy, ix] = float(_edge_exact(px[iy, ix], py[iy, ix], ax, ay, bx, by))
            es.append(e)
            scales.append(sc)
        w = [e / area for e in es]
        inside = (w[0] > 0) & (w[1] > 0) & (w[2] > 0)
        pz = w[0] * z0 + w[1] * z1 + w[2] * z2
        ok = inside & (pz >= 0)
        if not ok.any():
            continue
        marg = np.minimum.reduce([np.abs(e) / (U * np.maximum(sc, 1e-300)) for e, sc in zip(es, scales)])
        yo = (S - 1 - ys)[:, None] * np.ones_like(xs)[None, :]
        xo = np.ones_like(ys)[:, None] * (S - 1 - xs)[None, :]
        better = ok & (pz < zbuf[yo, xo])                       # strict: lowest face index keeps exact ties
        p2f[yo[better], xo[better]] = f
        zbuf[yo[better], xo[better]] = pz[better]
        margin[yo[better], xo[better]] = marg[better]
    zbuf[p2f < 0] = -1.0
    return p2f, zbuf, margin


def face_at_pixel(face_verts, f, yo, xo, S):
    """(covered?, depth, edge margin in float32 rounding units) of face f at output pixel (yo, xo), float64 + exact signs."""
    fv = np.asarray(face_verts, dtype=np.float32).astype(np.float64)
    (x0, y0, z0), (x1, y1, z1), (x2, y2, z2) = fv[f]
    px, py = float(pix_to_ndc(S - 1 - xo, S)), float(pix_to_ndc(S - 1 - yo, S))
    area = float(_edge_exact(x2, y2, x0, y0, x1, y1)) + K_EPS
    es, ms = [], []
    for (ax, ay, bx, by) in ((x1, y1, x2, y2), (x2, y2, x0, y0), (x0, y0, x1, y1)):
        e = float(_edge_exact(px, py, ax, ay, bx, by))
        sc = abs((px - ax) * (by - ay)) + abs((py - ay) * (bx - ax))
        es.append(e)
        ms.append(abs(e) / (U * max(sc, 1e-300)))
    w = [e / area for e in es]
    pz = w[0] * z0 + w[1] * z1 + w[2] * z2
    return (w[0] > 0 and w[1] > 0 and w[2] > 0 and pz >= 0), pz, min(ms)


def explain(face_verts, p2f_f32, zbuf_f32, S, edge_units=16.0, z_rel=16 * U):
    """Compares a float32 rasterisation (p3d_ref.c or a HIP kernel) of ONE mesh with the exact evaluation.
    -> dict(pixels, disagree, edge, ztie, unexplained, max_z_err) ; z error is measured on agreeing covered pixels."""
    p2f_x, z_x, _ = rasterize_exact(face_verts, S)
    p2f_f32 = np.asarray(p2f_f32)
    diff = np.nonzero(p2f_x != p2f_f32)
    out = {"pixels": S * S, "covered": int((p2f_x >= 0).sum()), "disagree": int(diff[0].size), "edge": 0, "ztie": 0,
           "unexplained": 0}
    for yo, xo in zip(*diff):
        fa, fb = int(p2f_x[yo, xo]), int(p2f_f32[yo, xo])
        near_edge = False
        depths = []
        for f in (fa, fb):
            if f < 0:
                continue
            cov, pz, m = face_at_pixel(face_verts, f, yo, xo, S)
            near_edge |= m <= edge_units
            depths.append(pz)
        if near_edge:
            out["edge"] += 1
        elif len(depths) == 2 and abs(depths[0] - depths[1]) <= z_rel * max(abs(depths[0]), abs(depths[1])) * 4:
            out["ztie"] += 1
        else:
            out["unexplained"] += 1
    same = (p2f_x == p2f_f32) & (p2f_x >= 0)
    zz = np.asarray(zbuf_f32, dtype=np.float64)
    out["max_z_rel_err"] = float((np.abs(zz[same] - z_x[same]) / np.maximum(np.abs(z_x[same]), 1e-30)).max()) if same.any() else 0.0
    return out
