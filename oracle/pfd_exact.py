"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by dsf_amd/).

An INDEPENDENT evaluation of pytorch3d 0.4.0's point-to-triangle rule (SURVEY.md Appendix A.4:
``PointTriangle3DistanceForward``) used to cross-check the float32 restatement in oracle/p3d_ref.c, as oracle/p3d_exact.py
does for the rasteriser: the same published rule -- normal / (|normal| + 1e-8), projection onto the plane, Gram-matrix
barycentrics with ``denom + 1e-8``, "inside and non-degenerate -> t^2, else the smallest of the three clamped segment
distances", argmin with the lowest index on exact ties -- written from the formulas, vectorised in numpy float64 (2^29 times
the resolution of the arithmetic under test).  Inputs are the float32 values taken as exact numbers.

``explain()`` compares a float32 result (p3d_ref.c or a HIP kernel) with it.  The squared distance is a continuous
function of the point even where the rule switches branches (inside <-> nearest edge), so a float32 evaluation may pick
another branch, or another triangle among near-equidistant ones, but its DISTANCE must agree to float32 rounding:
  * ``dist``: |d32 - d64(chosen triangle)| beyond the rounding bar -> the per-pair arithmetic is wrong;
  * ``argmin``: d64(chosen) exceeds the float64 minimum by more than the bar -> the selection is wrong;
  * indices that differ while both distances are inside the bar are ``ties`` (shared edges and vertices: ubiquitous).
This does not replace the wheel (parity against pytorch3d itself stays unpinned).
"""
import numpy as np

K_EPS = float(np.float32(1e-8))


def _seg(p, a, b):
    """squared distance point(s) p (P,1,3) to segments a->b (1,T,3), the rule's clamped projection"""
    ba = b - a
    l2 = (ba * ba).sum(-1)
    t = ((p - a) * ba).sum(-1) / np.where(l2 <= K_EPS, 1.0, l2)
    t = np.clip(t, 0.0, 1.0)
    d = p - (a + t[..., None] * ba)
    out = (d * d).sum(-1)
    deg = (p - b)
    return np.where(l2 <= K_EPS, (deg * deg).sum(-1), out)


def pair_dist2(points, tris):
    """points (P,3), tris (T,3,3) float32 -> (P,T) float64 squared distances by the published rule."""
    p = np.asarray(points, np.float32).astype(np.float64)[:, None, :]
    v = np.asarray(tris, np.float32).astype(np.float64)[None]
    v0, v1, v2 = v[:, :, 0], v[:, :, 1], v[:, :, 2]
    n = np.cross(v2 - v0, v1 - v0)
    nn = np.sqrt((n * n).sum(-1))
    n = n / (nn + K_EPS)[..., None]
    t = ((v0 - p) * n).sum(-1)
    p0 = p + t[..., None] * n
    q0, q1, q2 = v1 - v0, v2 - v0, p0 - v0
    d00, d01, d11 = (q0 * q0).sum(-1), (q0 * q1).sum(-1), (q1 * q1).sum(-1)
    d20, d21 = (q2 * q0).sum(-1), (q2 * q1).sum(-1)
    denom = d00 * d11 - d01 * d01 + K_EPS
    w1 = (d11 * d20 - d01 * d21) / denom
    w2 = (d00 * d21 - d01 * d20) / denom
    w0 = 1.0 - w1 - w2
    inside = (w0 >= 0) & (w0 <= 1) & (w1 >= 0) & (w1 <= 1) & (w2 >= 0) & (w2 <= 1) & (nn > K_EPS)
    edges = np.minimum(np.minimum(_seg(p, v0, v1), _seg(p, v0, v2)), _seg(p, v1, v2))
    return np.where(inside, t * t, edges)


def explain(points, tris, dists32, idxs32, rel=2e-5, ulps=16.0):
    """One mesh: -> dict(points, ties, dist, argmin, max_rel) ; ``dist`` / ``argmin`` count errors above rounding.
    Rounding bar: ``rel`` of the distance plus the square of ``ulps`` float32 roundings of the largest coordinate (a point ON
    the surface has distance 0 in exact arithmetic and (coordinate rounding)^2 in float32)."""
    L = max(float(np.abs(np.asarray(points)).max()), float(np.abs(np.asarray(tris)).max()), 1e-30)
    absolute = (ulps * 2.0 ** -24 * L) ** 2
    d = pair_dist2(points, tris)
    P = d.shape[0]
    best = d.min(1)
    idx64 = d.argmin(1)
    idxs32 = np.asarray(idxs32).astype(np.int64)
    chosen = d[np.arange(P), idxs32]
    d32 = np.asarray(dists32, np.float64)
    bar = rel * np.maximum(best, chosen) + absolute
    out = {"points": P, "ties": int(((idxs32 != idx64) & (chosen - best <= bar)).sum()),
           "dist": int((np.abs(d32 - chosen) > bar).sum()), "argmin": int((chosen - best > bar).sum()),
           "max_rel": float((np.abs(d32 - chosen) / np.maximum(chosen, absolute)).max())}
    return out
