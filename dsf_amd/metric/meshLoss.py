"""Drop-in counterpart of the reference's ``metric/meshLoss.py`` (point-cloud -> mesh losses).

The reference packs B (or 15*B) single-mesh ``Meshes``/``Pointclouds`` objects in Python lists
and calls pytorch3d's CUDA op (/root/reference/metric/meshLoss.py:21-70, 347-395).  Here the
batched losses go through one fused HIP launch (``dsf_mesh_point_dist_forward``): each point is
tested against the triangles of its own part only, which is exactly the subset the reference
keeps after its 15x replicated launch.  ``point_face_distance`` is the packed pytorch3d-style op.
"""
import torch

from .. import ops


def point_face_distance(points, points_first_idx, tris, tris_first_idx, max_points):
    """pytorch3d ``_PointFaceDistance.apply`` contract (reference :21-70): (P,) squared distances."""
    return ops.PointFaceDistance.apply(points, points_first_idx, tris, tris_first_idx, max_points)


def _as_parts(faces_list, device):
    """list of (F_j,3) face tensors (float or int) -> (int32 cat, int32 offsets)."""
    lens = [int(f.shape[0]) for f in faces_list]
    cat = torch.cat([f.reshape(-1, 3) for f in faces_list]).to(device=device, dtype=torch.int32).contiguous()
    first = torch.tensor([0] + list(torch.tensor(lens).cumsum(0).tolist()), dtype=torch.int32, device=device)
    return cat, first


_PART_CACHE = {}


def _cached_parts(faces_list, device):
    """the int32 table of a list of face tensors, cached per list.  The entry HOLDS the face tensors it was built from (so
    their addresses cannot be handed to other tensors while it lives) and the key carries their version counters (an in-place
    edit of a face list makes a new entry): a table is never served for other faces than those it was built from."""
    key = (tuple((f.data_ptr(), int(f.shape[0]), f._version) for f in faces_list), str(device))
    hit = _PART_CACHE.get(key)
    if hit is None:
        if len(_PART_CACHE) > 64:
            _PART_CACHE.clear()
        hit = _PART_CACHE[key] = _as_parts(faces_list, device) + (tuple(faces_list),)
    return hit[0], hit[1]


def _masked_part_mean(dis, pcl_seg, n_parts):
    """mean over the points of part j with dis > 0, 0 when there are none (reference :389-394)."""
    if dis.is_cuda and dis.dtype == torch.float32 and pcl_seg.dtype == torch.int64 and n_parts <= 16 and dis.dim() == 2:
        return ops.PartMean.apply(dis, pcl_seg, n_parts)     # one launch each way (csrc/step_ops.hip)
    labels = torch.arange(1, n_parts + 1, device=dis.device).view(1, n_parts, 1)
    sel = pcl_seg.unsqueeze(1).eq(labels)
    per = torch.where(sel, dis.unsqueeze(1), torch.zeros_like(dis).unsqueeze(1))
    valid = per.gt(0).sum(-1)
    loss = per.sum(-1) / (valid + 1e-8)
    return torch.where(valid.eq(0), torch.zeros_like(loss), loss)


def ICPLoss(mesh, pcl, faces):
    """mesh (B,V,3), pcl (B,P,3), faces (F,3) -> (B,) mean squared point-to-mesh distance (reference :347-353)."""
    cat, first = _cached_parts([faces], mesh.device)
    dis, _ = ops.MeshPointDistance.apply(mesh, pcl, cat, first, None, 1)
    return dis.mean(-1)


def JointICPLoss(mesh, pcl, faces, pcl_seg):
    """faces: list of 15 part face lists (MANO_SMPL.joint_faces); pcl_seg (B,P) labels 0..15 -> (B,15)
    (reference :377-395)."""
    cat, first = _cached_parts(list(faces), mesh.device)
    dis, _ = ops.MeshPointDistance.apply(mesh, pcl, cat, first, pcl_seg, len(faces))
    return _masked_part_mean(dis, pcl_seg, len(faces))


def FingerICPLoss(mesh, pcl, faces, pcl_seg):
    """5 finger parts (MANO_SMPL.finger_faces), labels 0..5 -> (B,5) (reference :356-374)."""
    cat, first = _cached_parts(list(faces), mesh.device)
    dis, _ = ops.MeshPointDistance.apply(mesh, pcl, cat, first, pcl_seg, len(faces))
    return _masked_part_mean(dis, pcl_seg, len(faces))
