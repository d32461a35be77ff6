"""Self-intersection volume of the hand parts on the GPU: counterpart of the reference's offline metric
``eval_coll.py`` (``MANO_SMPL.get_part_mesh`` :348-373, ``self_intersection`` :611-626, the pitch-2 then pitch-1 protocol of
:640-674), which voxelises 15 watertight part meshes with trimesh and counts the voxels of one part inside another, one
mesh at a time on the CPU.  Here a whole batch of meshes goes through ``dsf_part_intersection_volume`` (csrc/volume.hip).

Part meshes.  The reference reads them from ``MANO_PART.pkl`` (vertex id lists ``v-i`` into the 779 + 14 "water mesh" and
face lists ``f-i``; absent, MANO-licensed) plus 14 hard-coded boundary loops whose means are the cap vertices (:350-366).
``PartModel.from_dict`` takes exactly that data; ``PartModel.from_skinning`` derives an equivalent watertight decomposition
for ANY MANO-shaped model (the synthetic asset included): every face goes to the part of its vertices' dominant joint, each
boundary loop of a part is closed by a fan around the mean of its vertices -- the reference's construction.
"""
import ctypes

import numpy as np
import torch

from . import _lib as L
from ._lib import I, check, stream_ptr

PARENT_ID = [0, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13]            # eval_coll.py:615
JOINT_TO_PART = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 0, 13, 14]    # 16 skinning joints -> 15 parts (thumb root with the palm)


def valid_pairs(n_parts=15, parent_id=PARENT_ID):
    """the (s, t) pairs ``self_intersection`` visits (:617-621): t > s, neither the other's parent"""
    return [(s, t) for s in range(n_parts) for t in range(s, n_parts)
            if not (s == t or parent_id[s] == t or parent_id[t] == s)]


def _boundary_loops(faces):
    """edges used by exactly one face of the set (undirected: the reference's wrist-cap faces are not consistently
    oriented with the MANO faces, which parity tests do not mind), chained into closed loops (vertex id lists)"""
    use = {}
    for f in faces:
        for a, b in ((f[0], f[1]), (f[1], f[2]), (f[2], f[0])):
            k = (min(a, b), max(a, b))
            use[k] = use.get(k, 0) + 1
    adj = {}
    for (a, b), n in use.items():
        if n == 1:
            adj.setdefault(a, []).append(b)
            adj.setdefault(b, []).append(a)
    loops = []
    while adj:
        start = next(iter(adj))
        loop, prev, cur = [start], None, start
        while True:
            nb = adj[cur]
            b = nb.pop()                                   # any unused boundary edge at this vertex
            adj[b].remove(cur)
            if not adj[cur]:
                del adj[cur]
            if b in adj and not adj[b]:
                del adj[b]
            if b == start:
                break
            loop.append(b)
            prev, cur = cur, b
        loops.append(loop)
    return loops


class PartModel:
    """Watertight part meshes over a shared vertex pool = the mesh vertices + one cap vertex per boundary loop."""

    def __init__(self, n_mesh_verts, cap_loops, part_faces, parent_id=PARENT_ID):
        self.n_mesh_verts = int(n_mesh_verts)
        self.cap_loops = [np.asarray(l, dtype=np.int64) for l in cap_loops]
        self.part_faces = [np.asarray(f, dtype=np.int64).reshape(-1, 3) for f in part_faces]
        self.parent_id = list(parent_id)
        self.n_parts = len(self.part_faces)
        self.pairs = valid_pairs(self.n_parts, self.parent_id)
        self._dev = {}

    @classmethod
    def from_dict(cls, model_part, edge_vertex_id, n_mesh_verts=779, part_num=15):
        """``MANO_PART.pkl`` as the reference reads it (eval_coll.py:139-147, 350-372): ``v-i`` = ids into the water mesh,
        ``f-i`` = faces over the part's own vertex list."""
        faces = [np.asarray(model_part['v-%d' % i])[np.asarray(model_part['f-%d' % i])] for i in range(part_num)]
        return cls(n_mesh_verts, edge_vertex_id, faces)

    @classmethod
    def from_skinning(cls, faces, weights, joint_to_part=JOINT_TO_PART, n_mesh_verts=779):
        faces = np.asarray(faces, dtype=np.int64)
        w = np.asarray(weights)
        label = np.asarray(joint_to_part)[np.argmax(w, axis=-1)]
        label = np.concatenate([label, np.zeros(n_mesh_verts - label.shape[0], dtype=label.dtype)])   # the wrist-cap vertex: palm
        n_parts = int(max(joint_to_part)) + 1
        fl = np.sort(label[faces], axis=1)
        # majority label of the three vertices (ties: the smallest)
        face_part = np.where(fl[:, 1] == fl[:, 2], fl[:, 1], fl[:, 0])
        loops, part_faces = [], []
        for i in range(n_parts):
            mine = faces[face_part == i]
            caps = []
            for loop in _boundary_loops(mine.tolist()):
                cap = n_mesh_verts + len(loops)
                loops.append(loop)
                caps += [[loop[(k + 1) % len(loop)], loop[k], cap] for k in range(len(loop))]
            part_faces.append(np.concatenate([mine, np.asarray(caps, dtype=np.int64).reshape(-1, 3)]))
        return cls(n_mesh_verts, loops, part_faces)

    # ---- the reference's ``get_part_mesh`` ---------------------------------------------------------------------
    def water_mesh(self, mesh):
        """(B, V, 3) mesh -> (B, V + n_caps, 3): the cap vertices are the means of their loops (eval_coll.py:364-367)"""
        caps = [mesh[:, torch.as_tensor(l, device=mesh.device)].mean(dim=1, keepdim=True) for l in self.cap_loops]
        return torch.cat([mesh] + caps, dim=1).contiguous()

    def get_part_mesh(self, mesh):
        """one mesh (V,3) numpy -> list of (vertices, faces) per part, as the reference returns trimesh objects"""
        m = np.asarray(mesh, dtype=np.float64)
        pool = np.concatenate([m] + [m[l].mean(axis=0, keepdims=True) for l in self.cap_loops], axis=0)
        return [(pool, f) for f in self.part_faces]

    def _device_tables(self, device):
        t = self._dev.get(device)
        if t is None:
            cat = np.concatenate(self.part_faces).astype(np.int32)
            first = np.concatenate([[0], np.cumsum([f.shape[0] for f in self.part_faces])]).astype(np.int32)
            t = (torch.from_numpy(cat).to(device), torch.from_numpy(first).to(device),
                 torch.tensor(self.pairs, dtype=torch.int32, device=device).reshape(-1, 2), int(max(f.shape[0] for f in self.part_faces)))
            self._dev[device] = t
        return t


def self_intersection(part_model, mesh, pitch=2, grid=None, return_pairs=False):
    """``self_intersection`` (eval_coll.py:611-626) for a batch: mesh (B, V, 3) f32 on the GPU (mm) -> volumes (B,) float64
    (count * pitch^3) [, per-pair counts (B, n_pairs) int32].  Reading the result is the only synchronisation."""
    if not mesh.is_cuda:
        raise RuntimeError("dsf_amd ops run on the GPU only (got a %s tensor); there is no CPU path" % mesh.device)
    pool = part_model.water_mesh(mesh.float())
    B, V, _ = pool.shape
    faces, first, pairs, max_f = part_model._device_tables(mesh.device)
    if grid is None:
        ext = float((mesh.amax(dim=1) - mesh.amin(dim=1)).max()) if B else 0.0          # one host read: sizing only
        grid = max(32, int(np.ceil((ext / pitch + 4) / 32.0)) * 32)
    nbytes = int(L.lib().dsf_part_volume_workspace_bytes(I(B), I(part_model.n_parts), I(grid)))
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=mesh.device)
    count = torch.zeros(max(B, 1), dtype=torch.int64, device=mesh.device)
    pc = torch.zeros((B, pairs.shape[0]), dtype=torch.int32, device=mesh.device) if return_pairs else None
    status = torch.zeros(1, dtype=torch.int32, device=mesh.device)
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    check(L.lib().dsf_part_intersection_volume(vp(pool), vp(faces), vp(first), vp(pairs), I(B), I(V), I(part_model.n_parts),
                                               I(faces.shape[0]), I(pairs.shape[0]), I(max_f), ctypes.c_double(float(pitch)),
                                               I(grid), vp(ws), vp(count), vp(pc), vp(status), stream_ptr()),
          "dsf_part_intersection_volume")
    st = int(status.item())
    if st & 2:
        raise ValueError("max_iter exceeded!")                         # trimesh's error for faces needing > 10 subdivisions
    if st & 1:
        raise RuntimeError("self_intersection: a part does not fit a %d^3 voxel grid at pitch %g (pass a larger `grid`)" % (grid, pitch))
    vol = count[:B].double() * float(pitch) ** 3
    return (vol, pc) if return_pairs else vol


def intersection_volumes(part_model, meshes, chunk=256):
    """The reference's two-pass protocol (eval_coll.py:640-674): pitch 2 for every mesh, pitch 1 for those that collide.
    meshes (N, V, 3) on the GPU -> (coll_vox_pitch2 (N,), coll_vox_pitch1 (N,)) float64 on the host."""
    v2 = torch.cat([self_intersection(part_model, meshes[i:i + chunk], 2) for i in range(0, meshes.shape[0], chunk)])
    hit = torch.nonzero(v2 > 0).flatten()
    v1 = torch.zeros_like(v2)
    for i in range(0, hit.numel(), chunk):
        idx = hit[i:i + chunk]
        v1[idx] = self_intersection(part_model, meshes[idx], 1)
    return v2.cpu().numpy(), v1.cpu().numpy()
