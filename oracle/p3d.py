"""ORACLE -- TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/p3d_ref.c."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_p3d.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "p3d_ref.c")):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def project_verts(verts, cam=(588.03, 587.07, 320.0, 240.0), image_size=(640, 480)):
    v = np.ascontiguousarray(verts, dtype=np.float32)
    out = np.empty_like(v)
    f = ctypes.c_float
    lib().orc_project_verts(_p(v), ctypes.c_int64(v.size // 3), f(cam[0]), f(cam[1]), f(cam[2]), f(cam[3]),
                            f(image_size[0]), f(image_size[1]), _p(out))
    return out


def rasterize_meshes(face_verts, mesh_first, mesh_nfaces, S, want_bary=True):
    fv = np.ascontiguousarray(face_verts, dtype=np.float32)
    mf = np.ascontiguousarray(mesh_first, dtype=np.int64)
    mn = np.ascontiguousarray(mesh_nfaces, dtype=np.int64)
    N = mf.shape[0]
    p2f = np.empty((N, S, S), dtype=np.int64)
    zbuf = np.empty((N, S, S), dtype=np.float32)
    bary = np.empty((N, S, S, 3), dtype=np.float32) if want_bary else None
    dists = np.empty((N, S, S), dtype=np.float32) if want_bary else None
    lib().orc_rasterize_meshes(_p(fv), _p(mf), _p(mn), ctypes.c_int(N), ctypes.c_int(S), _p(p2f), _p(zbuf),
                               _p(bary) if want_bary else None, _p(dists) if want_bary else None)
    return p2f, zbuf, bary, dists


def rasterize_backward_zbuf(face_verts, pix_to_face, grad_zbuf):
    fv = np.ascontiguousarray(face_verts, dtype=np.float32)
    p2f = np.ascontiguousarray(pix_to_face, dtype=np.int64)
    gz = np.ascontiguousarray(grad_zbuf, dtype=np.float32)
    N, S = p2f.shape[0], p2f.shape[1]
    out = np.zeros_like(fv)
    lib().orc_rasterize_backward_zbuf(_p(fv), _p(p2f), _p(gz), ctypes.c_int(N), ctypes.c_int(S), _p(out))
    return out


def point_face_dist_forward(points, points_first, tris, tris_first):
    pts = np.ascontiguousarray(points, dtype=np.float32)
    tr = np.ascontiguousarray(tris, dtype=np.float32)
    pf = np.ascontiguousarray(points_first, dtype=np.int64)
    tf = np.ascontiguousarray(tris_first, dtype=np.int64)
    P, T = pts.shape[0], tr.shape[0]
    d = np.empty(P, dtype=np.float32)
    i = np.empty(P, dtype=np.int64)
    lib().orc_point_face_dist_forward(_p(pts), _p(pf), _p(tr), _p(tf), ctypes.c_int(pf.shape[0]),
                                      ctypes.c_int64(P), ctypes.c_int64(T), _p(d), _p(i))
    return d, i


def point_face_dist_backward(points, tris, idxs, grad_dists):
    pts = np.ascontiguousarray(points, dtype=np.float32)
    tr = np.ascontiguousarray(tris, dtype=np.float32)
    ix = np.ascontiguousarray(idxs, dtype=np.int64)
    g = np.ascontiguousarray(grad_dists, dtype=np.float32)
    gp = np.zeros_like(pts)
    gt = np.zeros_like(tr)
    lib().orc_point_face_dist_backward(_p(pts), _p(tr), _p(ix), _p(g), ctypes.c_int64(pts.shape[0]), _p(gp), _p(gt))
    return gp, gt
