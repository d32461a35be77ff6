"""``torch.library`` registration of the native operator boundary: the four ``pytorch3d._C`` entry points the
reference's dependency exposes (``rasterize_meshes``, ``rasterize_meshes_backward``, ``point_face_dist_forward``,
``point_face_dist_backward``; call sites mano_layer.py:1083, metric/meshLoss.py:52-66) as ``torch.ops.dsf.*`` with

* the HIP launchers of ``include/dsf_hip.h`` as the CUDA(= ROCm) kernels -- there is no CPU kernel: the ops raise on CPU tensors,
* fake (meta) kernels, so that ``FakeTensorMode`` / ``torch.compile`` / ``torch.export`` can trace through them,
* autograd formulas (``register_autograd``) that route to the ``*_backward`` ops.

The Python layer of this package keeps calling ``dsf_amd.ops`` (``autograd.Function`` + ctypes, same launchers); this module is
the dispatcher-visible face of the same kernels for code that was written against ``pytorch3d._C`` or wants to compile.
Import it to register: ``import dsf_amd.torch_ops``.
"""
import torch
from torch.library import custom_op

from . import _lib as L
from ._lib import I, I64, F, ptr, check, stream_ptr, f32


def _gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dsf ops run on the GPU only (got a %s tensor); there is no CPU kernel" % t.device)


@custom_op("dsf::rasterize_meshes", mutates_args=(), device_types="cuda")
def rasterize_meshes(face_verts: torch.Tensor, mesh_to_face_first_idx: torch.Tensor, num_faces_per_mesh: torch.Tensor,
                     image_size: int, blur_radius: float = 0.0, faces_per_pixel: int = 1, perspective_correct: bool = False,
                     clip_barycentric_coords: bool = False,
                     cull_backfaces: bool = False) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (pix_to_face int64 (N,S,S,1), zbuf, barycentric (N,S,S,1,3), dists): ``pytorch3d._C.rasterize_meshes`` minus the two
    bin-size hints (naive path semantics)."""
    _gpu(face_verts, mesh_to_face_first_idx, num_faces_per_mesh)
    fv = f32(face_verts)
    N, S = mesh_to_face_first_idx.shape[0], int(image_size)
    new = lambda shape, dt=torch.float32: torch.empty(shape, device=fv.device, dtype=dt)
    p2f, zbuf, bary, dists = new((N, S, S, 1), torch.int64), new((N, S, S, 1)), new((N, S, S, 1, 3)), new((N, S, S, 1))
    ws = new((max(N, 1), 4))
    check(L.lib().dsf_rasterize_meshes(ptr(fv), ptr(mesh_to_face_first_idx.contiguous()), ptr(num_faces_per_mesh.contiguous()),
                                       I(N), I64(fv.shape[0]), I(S), F(blur_radius), I(faces_per_pixel),
                                       I(int(perspective_correct)), I(int(clip_barycentric_coords)), I(int(cull_backfaces)),
                                       ptr(p2f), ptr(zbuf), ptr(bary), ptr(dists), ptr(ws), stream_ptr()), "dsf_rasterize_meshes")
    return p2f, zbuf, bary, dists


@rasterize_meshes.register_fake
def _(face_verts, mesh_to_face_first_idx, num_faces_per_mesh, image_size, blur_radius=0.0, faces_per_pixel=1,
      perspective_correct=False, clip_barycentric_coords=False, cull_backfaces=False):
    N, S = mesh_to_face_first_idx.shape[0], image_size
    new = lambda shape, dt=torch.float32: face_verts.new_empty(shape, dtype=dt)
    return new((N, S, S, 1), torch.int64), new((N, S, S, 1)), new((N, S, S, 1, 3)), new((N, S, S, 1))


@custom_op("dsf::rasterize_meshes_backward", mutates_args=(), device_types="cuda")
def rasterize_meshes_backward(face_verts: torch.Tensor, pix_to_face: torch.Tensor, grad_zbuf: torch.Tensor) -> torch.Tensor:
    """-> grad_face_verts; only zbuf is differentiated by the reference (mano_layer.py:1023)."""
    _gpu(face_verts, pix_to_face, grad_zbuf)
    fv = f32(face_verts)
    g = torch.empty_like(fv)
    check(L.lib().dsf_rasterize_meshes_backward(ptr(fv), ptr(pix_to_face.contiguous()), ptr(f32(grad_zbuf)), ptr(None), ptr(None),
                                                I(pix_to_face.shape[0]), I64(fv.shape[0]), I(pix_to_face.shape[1]), ptr(g),
                                                stream_ptr()), "dsf_rasterize_meshes_backward")
    return g


@rasterize_meshes_backward.register_fake
def _(face_verts, pix_to_face, grad_zbuf):
    return torch.empty_like(face_verts)


def _raster_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], output[0])


def _raster_backward(ctx, g_p2f, g_zbuf, g_bary, g_dists):
    face_verts, p2f = ctx.saved_tensors
    return (torch.ops.dsf.rasterize_meshes_backward(face_verts, p2f, g_zbuf),) + (None,) * 8


rasterize_meshes.register_autograd(_raster_backward, setup_context=_raster_setup)


@custom_op("dsf::point_face_dist_forward", mutates_args=(), device_types="cuda")
def point_face_dist_forward(points: torch.Tensor, points_first_idx: torch.Tensor, tris: torch.Tensor,
                            tris_first_idx: torch.Tensor, max_points: int) -> tuple[torch.Tensor, torch.Tensor]:
    """-> (squared distances (P,), closest-triangle indices (P,) int64): ``pytorch3d._C.point_face_dist_forward``."""
    _gpu(points, points_first_idx, tris, tris_first_idx)
    pts, tr = f32(points), f32(tris)
    P, T, N = pts.shape[0], tr.shape[0], points_first_idx.shape[0]
    dists = torch.empty(P, device=pts.device, dtype=torch.float32)
    idxs = torch.empty(P, device=pts.device, dtype=torch.int64)
    check(L.lib().dsf_point_face_dist_forward(ptr(pts), ptr(points_first_idx.contiguous()), ptr(tr), ptr(tris_first_idx.contiguous()),
                                              I(N), I64(P), I64(T), I64(int(max_points)), ptr(dists), ptr(idxs), stream_ptr()),
          "dsf_point_face_dist_forward")
    return dists, idxs


@point_face_dist_forward.register_fake
def _(points, points_first_idx, tris, tris_first_idx, max_points):
    return points.new_empty((points.shape[0],)), points.new_empty((points.shape[0],), dtype=torch.int64)


@custom_op("dsf::point_face_dist_backward", mutates_args=(), device_types="cuda")
def point_face_dist_backward(points: torch.Tensor, tris: torch.Tensor, idxs: torch.Tensor,
                             grad_dists: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """-> (grad_points, grad_tris): ``pytorch3d._C.point_face_dist_backward``."""
    _gpu(points, tris, idxs, grad_dists)
    pts, tr = f32(points), f32(tris)
    gp, gt = torch.empty_like(pts), torch.empty_like(tr)
    check(L.lib().dsf_point_face_dist_backward(ptr(pts), ptr(tr), ptr(idxs.contiguous()), ptr(f32(grad_dists)), I64(pts.shape[0]),
                                               I64(tr.shape[0]), ptr(gp), ptr(gt), stream_ptr()), "dsf_point_face_dist_backward")
    return gp, gt


@point_face_dist_backward.register_fake
def _(points, tris, idxs, grad_dists):
    return torch.empty_like(points), torch.empty_like(tris)


def _pfd_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[2], output[1])


def _pfd_backward(ctx, g_dists, _g_idx):
    points, tris, idxs = ctx.saved_tensors
    gp, gt = torch.ops.dsf.point_face_dist_backward(points, tris, idxs, g_dists)
    return gp, None, gt, None, None


point_face_dist_forward.register_autograd(_pfd_backward, setup_context=_pfd_setup)
