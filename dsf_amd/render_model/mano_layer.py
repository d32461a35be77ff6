"""Drop-in counterpart of the reference's ``render_model/mano_layer.py`` for MI355X.

Same classes, method names, argument order and return tuples
(``MANO_SMPL`` /root/reference/render_model/mano_layer.py:82-770, ``Render``
:925-1340), but every arithmetic path runs in the hand-written HIP kernels of
``libdsf_hip.so`` (include/dsf_hip.h): the ~40 torch kernels + Python loop of
``MANO_SMPL.forward`` are one fused launch, and pytorch3d's 640x640 rasteriser
followed by two ``grid_sample`` calls is the fused crop renderer.  There is no
CPU fallback: calling these on CPU tensors raises.
"""

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import ops
from ..assets import load_mano_dict

MANO2HANDS = [0, 13, 1, 4, 10, 7, 14, 15, 20, 2, 3, 16, 5, 6, 17, 11, 12, 19, 8, 9, 18]
MANO2MSRA = [0, 1, 2, 3, 16, 4, 5, 6, 17, 10, 11, 12, 19, 7, 8, 9, 18, 13, 14, 15, 20]
MANO2ICVL = [0, 13, 14, 15, 1, 2, 3, 4, 5, 6, 10, 11, 12, 7, 8, 9]
MANO2NYU = [18, 8, 19, 11, 17, 5, 16, 2, 20, 15, 14, 0]
HANDS2MANO = [0, 2, 9, 10, 3, 12, 13, 5, 18, 19, 4, 15, 16, 1, 6, 7, 11, 14, 20, 17, 8]

_WRIST_RING = [121, 214, 215, 279, 239, 234, 92, 38, 122, 118, 117, 119, 120, 108, 79, 78]
_TIPS = [333, 444, 672, 555, 744]


def quat2mat(quat):
    """(B,4) (w,x,y,z) -> (B,3,3); normalises first (reference :773-794)."""
    q = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    m = torch.stack([w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
                     2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
                     2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z], dim=1)
    return m.view(-1, 3, 3)


def batch_rodrigues(theta):
    """(N,3) axis-angle -> (N,3,3) through the quaternion (reference :797-805)."""
    angle = torch.norm(theta + 1e-8, p=2, dim=1, keepdim=True)
    half = angle * 0.5
    return quat2mat(torch.cat([torch.cos(half), torch.sin(half) * (theta / angle)], dim=1))


def _rows62(model_paras):
    """``model_paras[:, :62]`` (reference :1078, :1103), without the slice when the rows ARE 62 wide: autograd's backward of a slice is a
    zero fill + a copy, twice per render"""
    return model_paras if model_paras.size(-1) == 62 else model_paras[:, :62]


def _rotmat(rot):
    return batch_rodrigues(rot) if rot.size(-1) == 3 else quat2mat(rot)


def RotationPoints(verts, joints, center3d, rot):
    """Rotate about ``center3d`` (reference :874-884)."""
    # (B, V, 3) x (B, 3, 3)^T as ONE batched product per tensor: the reference's broadcast form, matmul((B,1,3,3), (B,V,3,1)), is a
    # bmm over B*V 3x3 matrices -- 149,568 of them at config 4's 192 meshes, 1.07 ms per call in hipBLASLt (round-2 profile)
    if verts.is_cuda and not (torch.is_grad_enabled() and (verts.requires_grad or joints.requires_grad or rot.requires_grad)) \
            and verts.dim() == 3 and joints.dim() == 3 and rot.dim() == 2:
        return ops.view_rotate(verts, joints, center3d, rot)  # one launch (csrc/step_ops.hip): the synthetic branch renders without gradients
    Rt = _rotmat(rot).transpose(1, 2)
    c = center3d.unsqueeze(1)
    return torch.bmm(verts - c, Rt) + c, torch.bmm(joints - c, Rt) + c


def RotationNormalPoints(points, rot):
    return torch.bmm(points, _rotmat(rot).transpose(1, 2))


def _collision_mask():
    """66x66 sphere-pair mask (reference :240-269): 21 palm spheres never collide with each
    other; a finger bone ignores its own, its neighbours' and its knuckle's spheres; the
    thumb root ignores the palm."""
    n_palm, per = 21, 3
    m = torch.ones(66, 66)
    m[:n_palm, :n_palm] = 0
    for bone in range(15):
        finger = bone // 3 + 1
        lo = n_palm + per * bone
        rows = slice(lo, lo + per)
        if bone % 3 == 0:
            m[rows, 4 * finger] = 0
            m[4 * finger, rows] = 0
            m[rows, lo:lo + 2 * per] = 0
        else:
            m[rows, lo - per:min(lo + 2 * per + 1, n_palm + 3 * per * finger)] = 0
    t0 = n_palm + 12 * per
    m[t0:t0 + per + 1, :n_palm] = 0
    m[:n_palm, t0:t0 + per + 1] = 0
    return m


class MANO_SMPL(nn.Module):
    def __init__(self, mano_pkl_path, dataset, scale=1000):
        super().__init__()
        if 'msra' in dataset:
            self.transfer = MANO2MSRA
        elif 'icvl' in dataset:
            self.transfer = MANO2ICVL
        elif 'hands' in dataset:
            self.transfer = MANO2HANDS
        elif 'nyu' in dataset:
            self.transfer = MANO2NYU
        else:
            self.transfer = range(21)
        self.dataset = dataset
        self.scale = scale
        model = load_mano_dict(mano_pkl_path)

        t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).float().contiguous()
        faces = np.array(model['f'], dtype=np.int64)
        cap = np.array([[_WRIST_RING[i], _WRIST_RING[(i + 1) % 16], 778] for i in range(16)], dtype=np.int64)
        faces = np.concatenate([faces, cap], 0)                       # 1538 + 16 wrist-cap faces
        v_template = t32(model['v_template'])
        self.size = [v_template.shape[0], 3]
        sd = np.array(model['shapedirs'], dtype=np.float64)
        self.num_betas = sd.shape[-1]
        shapedirs = t32(sd.reshape(-1, self.num_betas).T)
        jreg = np.array(model['J_regressor'].T.toarray(), dtype=np.float64)
        tips = np.zeros((778, 5))
        tips[_TIPS, np.arange(5)] = 1
        J_regressor = t32(np.concatenate([jreg, tips], 1))            # (778, 21)
        pd = np.array(model['posedirs'], dtype=np.float64)
        posedirs = t32(pd.reshape(-1, pd.shape[-1]).T)
        self.parents = np.array(model['kintree_table'])[0].astype(np.int32)
        assert all(int(self.parents[i]) < i for i in range(1, 16)), "kinematic tree must be topologically ordered"
        w_np = np.array(model['weights'], dtype=np.float64)
        weight = t32(w_np)

        # index sets the reference derives in O(F*16) Python loops (:156-185)
        vertex_seg = np.argmax(w_np, axis=-1)
        self.vertex_seg = torch.from_numpy(vertex_seg).float()
        self.vertex_joint_index_list = [torch.from_numpy(np.nonzero(vertex_seg == i)[0]) for i in range(16)]
        strong = [np.nonzero(w_np[:, i] > 0.1)[0] for i in range(16)]
        joint_faces = [faces[np.isin(faces, strong[i]).any(1)] for i in range(1, 16)]
        self.vertex_finger_index_list = [torch.from_numpy(np.concatenate([strong[3 * k + 1], strong[3 * k + 2],
                                                                          strong[3 * k + 3]])) for k in range(5)]
        finger_faces = [faces[np.isin(faces, self.vertex_finger_index_list[k].numpy()).any(1)] for k in range(5)]
        self.finger_seg = torch.from_numpy(np.array([0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5])[vertex_seg])

        self.register_buffer('v_template', v_template)
        self.register_buffer('shapedirs', shapedirs)
        self.register_buffer('J_regressor', J_regressor)
        self.register_buffer('hands_comp', t32(model['hands_components']))
        self.register_buffer('hands_mean', t32(model['hands_mean']))
        self.register_buffer('posedirs', posedirs)
        self.register_buffer('e3', torch.eye(3))
        self.register_buffer('weight', weight)
        self.register_buffer('base_rot_mat_x', torch.tensor([[[1.0, 0, 0], [0, -1, 0], [0, 0, -1]]]))
        # face tensors keep the reference's float dtype at the API; int32 twins feed the kernels
        self.register_buffer('faces', torch.from_numpy(faces).float())
        self._n_joint_faces = [f.shape[0] for f in joint_faces]
        self._n_finger_faces = [f.shape[0] for f in finger_faces]
        self.register_buffer('_joint_faces_f', torch.from_numpy(np.concatenate(joint_faces)).float())
        self.register_buffer('_finger_faces_f', torch.from_numpy(np.concatenate(finger_faces)).float())
        self.register_buffer('faces_i32', torch.from_numpy(faces).int())
        self.register_buffer('joint_faces_i32', torch.from_numpy(np.concatenate(joint_faces)).int())
        self.register_buffer('joint_faces_first', torch.tensor(np.concatenate([[0], np.cumsum(self._n_joint_faces)]),
                                                                dtype=torch.int32))
        self.register_buffer('finger_faces_i32', torch.from_numpy(np.concatenate(finger_faces)).int())
        self.register_buffer('finger_faces_first', torch.tensor(np.concatenate([[0], np.cumsum(self._n_finger_faces)]),
                                                                 dtype=torch.int32))
        self.register_buffer('whole_first', torch.tensor([0, faces.shape[0]], dtype=torch.int32))

        # host precompute for the fused kernel: rest joints are linear in beta
        j16 = J_regressor[:, :16]
        self.register_buffer('j_template', (j16.double().t() @ v_template.double()).float().contiguous())
        jsd = torch.einsum('vj,kvc->kjc', j16.double(), shapedirs.double().view(self.num_betas, 778, 3))
        self.register_buffer('j_shapedirs', jsd.float().reshape(self.num_betas, 48).contiguous())
        jt = J_regressor.t().contiguous()
        rowptr, col, val = [0], [], []
        for j in range(21):
            nz = torch.nonzero(jt[j]).flatten()
            col.append(nz)
            val.append(jt[j][nz])
            rowptr.append(rowptr[-1] + nz.numel())
        self.register_buffer('jreg_rowptr', torch.tensor(rowptr, dtype=torch.int32))
        self.register_buffer('jreg_col', torch.cat(col).int())
        self.register_buffer('jreg_val', torch.cat(val).float())
        self.register_buffer('parents_i32', torch.from_numpy(self.parents.astype(np.int32)))
        self.register_buffer('wrist_ring_i32', torch.tensor(_WRIST_RING, dtype=torch.int32))
        self.register_buffer('jreg_mask_u8', (jt > 0).to(torch.uint8).contiguous())

        self.cur_device = None
        self.rotate_base = False
        self.child = [2, 3, 16, 5, 6, 17, 8, 9, 18, 11, 12, 19, 14, 15, 20]
        self.per_adj_shpere = 2
        self.interval_value = torch.linspace(0, 1, self.per_adj_shpere + 2)[:-1].reshape(1, 1, -1)
        self.interval = self.per_adj_shpere + 1
        self.plam_per_adj_shpere = 4
        self.plam_interval_value = torch.linspace(0, 1, self.plam_per_adj_shpere + 2)[1:-1].reshape(1, 1, -1)
        self.plam_interval = self.plam_per_adj_shpere + 1
        self.register_buffer('mask', _collision_mask())
        self._native_key = None

    # ---- views with the reference's attribute names -------------------------------------------
    @property
    def is_cuda(self):
        return self.v_template.is_cuda

    @property
    def joint_faces(self):
        return list(torch.split(self._joint_faces_f, self._n_joint_faces))

    @property
    def finger_faces(self):
        return list(torch.split(self._finger_faces_f, self._n_finger_faces))

    def _native(self):
        """C structs holding device pointers of the model buffers (rebuilt after .to()/.cuda())."""
        key = (self.v_template.data_ptr(), self.posedirs.data_ptr())
        if self._native_key != key:
            if not self.v_template.is_cuda:
                raise RuntimeError("MANO_SMPL buffers are on %s; dsf_amd runs on the GPU only -- call .cuda()"
                                   % self.v_template.device)
            def p(t):
                assert t.is_contiguous(), "model buffers handed to the kernels must be contiguous"
                return t.data_ptr()
            self.c_struct = L.dsf_mano_model(p(self.v_template), p(self.shapedirs), p(self.posedirs),
                                             p(self.J_regressor), p(self.j_template), p(self.j_shapedirs),
                                             p(self.hands_comp), p(self.hands_mean), p(self.weight),
                                             p(self.parents_i32), p(self.wrist_ring_i32), p(self.jreg_rowptr),
                                             p(self.jreg_col), p(self.jreg_val))
            sm = L.dsf_sphere_model()
            sm.jreg_mask = p(self.jreg_mask_u8)
            sm.coll_mask = p(self.mask)
            for i, v in enumerate(self.interval_value.flatten().tolist()):
                sm.t_finger[i] = v
            for i, v in enumerate(self.plam_interval_value.flatten().tolist()):
                sm.t_palm[i] = v
            self.sphere_struct = sm
            self._native_key = key
        return self

    # ---- MANO layer ---------------------------------------------------------------------------
    def _as_tensor(self, x):
        if not isinstance(x, torch.Tensor):
            x = torch.tensor(x, dtype=torch.float)
        return x.to(self.v_template.device)

    def forward(self, beta, theta, quat_or_euler, get_skin=False):
        beta, theta, quat_or_euler = self._as_tensor(beta), self._as_tensor(theta), self._as_tensor(quat_or_euler)
        verts, joints, Rs = ops.ManoFunction.apply(self._native(), beta, theta, quat_or_euler, None, 1.0, 1.0)
        return (verts, joints, Rs) if get_skin else joints

    def get_mano_vertices_packed(self, model_paras, global_scale=None):
        """get_mano_vertices on packed rows [rot 3|4, theta 45, beta 10, cam 4] (the network's output / Render._split
        layout) without slicing them apart: same values, one kernel each way.

        The trainer evaluates the layer twice on the same parameters within a step (``Render.render(mano)`` and
        ``get_mesh_xyz(mano)``, train_render.py:459-466 / 719): the last two results are kept, keyed by the memory,
        layout and version of the parameter rows (plus the global write epoch that raw-pointer writers such as FusedAdamW
        bump, which torch's version counter does not see), so the second evaluation reuses the first one's output (and
        its autograd node: the backward kernel then runs once on the summed gradients).  Every step class clears the
        memo before it builds a new graph."""
        k2 = 1.0 if global_scale is None else float(global_scale)
        key = (model_paras.data_ptr(), tuple(model_paras.shape), tuple(model_paras.stride()), model_paras._version,
               model_paras.dtype, model_paras.requires_grad, torch.is_grad_enabled(), k2, L.WRITE_EPOCH[0])
        cache = self.__dict__.setdefault("_packed_cache", [])
        for ent in cache:
            if ent[0] == key:
                return ent[2], ent[3]
        verts, joints = ops.ManoPackedFunction.apply(self._native(), model_paras, 1000.0, k2)
        cache.insert(0, (key, model_paras, verts, joints))         # the entry holds the rows, so their memory cannot be reused
        del cache[2:]
        return verts, joints

    def clear_cache(self):
        self.__dict__["_packed_cache"] = []

    def get_mano_vertices(self, quat_or_euler, pose, shape, cam, global_scale=None):
        """-> verts (B,779,3), joints (B,21,3) in mm * global_scale * cam scale + cam trans."""
        quat_or_euler, pose, shape = self._as_tensor(quat_or_euler), self._as_tensor(pose), self._as_tensor(shape)
        cam = self._as_tensor(cam)
        k2 = 1.0 if global_scale is None else float(global_scale)
        verts, joints, _ = ops.ManoFunction.apply(self._native(), shape, pose, quat_or_euler, cam, 1000.0, k2)
        return verts, joints

    def quat2mat(self, quat):
        return quat2mat(quat)

    def batch_rodrigues(self, theta):
        return batch_rodrigues(theta)

    # ---- sphere model -------------------------------------------------------------------------
    def get_sphere_radius(self, joints, mesh):
        """-> centres (B,66,3), radii (B,66).  Forward only (gradients flow through calculate_coll)."""
        return ops.sphere_set(self._native().sphere_struct, joints.detach(), mesh.detach())

    def get_sphere(self, joints):
        dummy = joints.new_zeros(joints.size(0), 779, 3)
        return self.get_sphere_radius(joints, dummy)[0]

    def get_radius(self, joints, mesh):
        return self.get_sphere_radius(joints, mesh)[1]

    def calculate_coll(self, joints, meshs):
        rows = ops.CollisionRows.apply(self._native().sphere_struct, joints, meshs)
        return rows.mean()

    def seg_pcl(self, joints, joints_mano, mesh, pcl):
        """Part label 0..15 per point: centres from ``joints`` (pixel branch), radii from
        ``joints_mano`` (reference :404-426)."""
        sm = self._native().sphere_struct
        c, r = ops.sphere_mixed(sm, joints.detach(), joints_mano.detach(), mesh.detach())     # one launch, once per (joints, joints, mesh)
        return ops.seg_pcl(c, r, pcl.detach())


class _Fragments:
    def __init__(self, pix_to_face, zbuf, bary_coords, dists):
        self.pix_to_face, self.zbuf, self.bary_coords, self.dists = pix_to_face, zbuf, bary_coords, dists


class _MeshRasterizer:
    """Callable with the role of pytorch3d's MeshRasterizer for the reference's settings
    (:939-952): world verts (B,V,3) -> fragments of the max(image_size)^2 raster."""

    def __init__(self, render):
        self.render = render

    def __call__(self, verts):
        r = self.render
        B = verts.size(0)
        nf = r.mano_layer.faces_i32.size(0)
        fv = _ProjectFaceVerts.apply(verts, r.mano_layer.faces_i32, r.cam)
        first = torch.arange(B, device=verts.device, dtype=torch.int64) * nf
        count = torch.full((B,), nf, device=verts.device, dtype=torch.int64)
        return _Fragments(*ops.RasterizeMeshesFunction.apply(fv, first, count, max(r.img_size)))


class _ProjectFaceVerts(torch.autograd.Function):
    """camera transform + Meshes packing; backward in torch (full-mode raster is off the hot path)."""

    @staticmethod
    def forward(ctx, verts, faces_i32, cam):
        ctx.save_for_backward(verts, faces_i32)
        ctx.cam = cam
        return ops.project_face_verts(verts, faces_i32, cam)

    @staticmethod
    def backward(ctx, g):
        verts, faces = ctx.saved_tensors
        cam = ctx.cam
        B, V, _ = verts.shape
        hw, hh = cam.img_w / 2.0, cam.img_h / 2.0
        fxn, fyn = cam.fx / hw, cam.fy / hh
        pxn, pyn = -(cam.px - hw) / hw, -(cam.py - hh) / hh
        g = g.view(B, -1, 3)                                          # (B, F*3, 3) grads of (xn, yn, zv)
        idx = faces.long().view(-1)
        acc = torch.zeros_like(verts).index_add_(1, idx, g)
        X, Y, Z = verts.unbind(-1)
        xn, yn = (-X * fxn + Z * pxn) / Z, (-Y * fyn + Z * pyn) / Z
        gx, gy, gz = acc.unbind(-1)
        return torch.stack([-gx * fxn / Z, -gy * fyn / Z, gz + gx * (pxn - xn) / Z + gy * (pyn - yn) / Z], -1), None, None


class Render(nn.Module):
    def __init__(self, mano_path, dataset, cam_para, image_size, crop_size=(128, 128), inverse='torch'):
        super().__init__()
        self.mano_layer = MANO_SMPL(mano_path + '/MANO_RIGHT.pkl', dataset)
        self.paras = cam_para
        self.img_size = image_size
        self.crop_size = crop_size
        self.cam = L.camera(cam_para, image_size)
        self.inverse = inverse          # 'torch': torch.inverse(M) as the reference; 'closed': affine closed form
        self.rasterizer = _MeshRasterizer(self)
        S = max(image_size)
        # 640 -> 480 row table of Render.resize, derived from torch's own ops
        idx = torch.arange(S, dtype=torch.float32).view(1, 1, S, 1).expand(1, 1, S, S).contiguous()
        grid = F.affine_grid(torch.tensor([[[1.0, 0, 0], [0, 1.0, 0]]]), (1, 1, image_size[1], image_size[0]),
                             align_corners=False)
        rows = F.grid_sample(idx, grid, mode='nearest', align_corners=False)[0, 0, :, 0]
        self.register_buffer('resize_rowmap', rows.round().int().contiguous())
        c = crop_size[0]
        g = 2 * (torch.arange(c).float() + 0.5) / c - 1.0
        self.register_buffer('xy_mesh', torch.stack(torch.meshgrid(g, g, indexing='xy'), -1).reshape(1, -1, 2))
        ii = torch.arange(c).float()
        xx, yy = torch.meshgrid(ii, ii, indexing='xy')
        self.register_buffer('crop_mesh', torch.stack([xx, yy, torch.ones_like(xx)], -1).reshape(1, -1, 3))
        if dataset == 'nyu':
            self.depth_range = [500, 1200]
        if dataset == 'msra' or dataset == 'icvl':
            self.depth_range = [150, 600]

    # ---- parameter split ----------------------------------------------------------------------
    @staticmethod
    def _split(model_paras, augmentShape=None):
        qd = 4 if model_paras.size(-1) == 63 else 3
        beta = model_paras[:, qd + 45:qd + 55]
        if augmentShape is not None:
            beta = beta + augmentShape
        return model_paras[:, :qd], model_paras[:, qd:qd + 45], beta, model_paras[:, qd + 55:]

    # ---- fused leaf: world verts -> normalised crop -----------------------------------------------
    def _inverse(self, M, minv_closed):
        if self.inverse == 'closed' and minv_closed is not None:
            return minv_closed
        return ops.inverse3x3(M)                 # same LAPACK-style routine as torch.inverse, no host sync; once per M tensor

    def _depth_crop(self, hand_verts, center3d, cube_size, M=None, normalise=True):
        center2d, M_auto, _, minv_c = ops.crop_setup(center3d, cube_size, self.cam, self.crop_size[0],
                                                     want_closed_inverse=(M is None and self.inverse == 'closed'))
        if M is None:
            M = M_auto
            minv = self._inverse(M, minv_c)
        else:
            minv = ops.inverse3x3(M)
        cz = ops.column(center2d, 2) if normalise else None
        cbz = ops.column(cube_size, 2) if normalise else None
        img, p2f = ops.RenderCropFunction.apply(hand_verts, self.mano_layer.faces_i32, minv, self.resize_rowmap, cz, cbz,
                                                self.cam, max(self.img_size), self.crop_size[0])
        return img, center2d, M, minv

    def forward(self, model_paras, center3d, cube_size, augmentView=None, augmentShape=None, augmentCenter=None,
                augmentSize=None, mask=True):
        """Synthetic-branch render (reference :983-1039) -> 8-tuple."""
        batch_size = model_paras.size(0)
        device = model_paras.device
        quat, theta, beta, cam = self._split(model_paras, augmentShape)
        hand_verts, hand_joints = self.mano_layer.get_mano_vertices(quat, theta, beta, cam)
        if center3d is None:
            depth = torch.rand([batch_size, 1]) * (self.depth_range[1] - self.depth_range[0]) + self.depth_range[0]
            center3d = torch.cat((torch.zeros([batch_size, 2]), depth), dim=-1).to(device)
        fused = hand_verts.is_cuda and not (torch.is_grad_enabled() and (hand_verts.requires_grad or center3d.requires_grad or
                                                                         cube_size.requires_grad))
        if fused:
            # the placement (p - mean(joints) + centre), the view rotation and -- below -- the cube normalisation of both point
            # tensors as one launch each (csrc/step_ops.hip; this branch renders without gradients): 20 elementwise launches less
            hand_verts, hand_joints = ops.view_rotate(hand_verts, hand_joints, center3d, augmentView, recentre=True)
        else:
            synth_center = hand_joints.mean(dim=1, keepdim=True)
            hand_verts = hand_verts - synth_center
            hand_joints = hand_joints - synth_center
            hand_verts = hand_verts + center3d.unsqueeze(1)
            hand_joints = hand_joints + center3d.unsqueeze(1)
            if augmentView is not None:
                hand_verts, hand_joints = RotationPoints(hand_verts, hand_joints, center3d, augmentView)
        if augmentCenter is not None:
            center3d = center3d + augmentCenter
        if augmentSize is not None:
            cube_size = cube_size * augmentSize
        img, center2d, M, _ = self._depth_crop(hand_verts, center3d, cube_size)
        joint_uvd = self.JointTrans(hand_joints, M, center2d, cube_size)
        verts_uvd = self.JointTrans(hand_verts, M, center2d, cube_size)
        if fused:
            verts_xyz, joint_xyz = ops.cube_normalise(hand_verts, hand_joints, center3d, cube_size)
        else:
            joint_xyz = (hand_joints - center3d.unsqueeze(1)) / cube_size.unsqueeze(1) * 2
            verts_xyz = (hand_verts - center3d.unsqueeze(1)) / cube_size.unsqueeze(1) * 2
        if mask:
            img = self.mask_img(img, joint_uvd, 0.15, 0.3)
        return img, joint_uvd, verts_uvd, joint_xyz, verts_xyz, center3d, cube_size, M

    def render(self, model_paras, center3d, cube_size, M=None):
        """MANO params (cube-normalised) -> (img (B,1,128,128), joint_uvd, joint_xyz, mesh_xyz) (reference :1071-1097)."""
        hand_verts, hand_joints = self.mano_layer.get_mano_vertices_packed(_rows62(model_paras), global_scale=1 / 125)
        if hand_verts.is_cuda and not center3d.requires_grad and not cube_size.requires_grad:
            # the four point transforms below in one launch each way (csrc/step_ops.hip; same operations per element)
            hand_verts, hand_joints, mesh_xyz, joint_xyz = ops.CubePoints.apply(hand_verts, hand_joints, center3d, cube_size)
            img, center2d, M, _ = self._depth_crop(hand_verts, center3d, cube_size)
            return img, self.JointTrans(hand_joints, M, center2d, cube_size), joint_xyz, mesh_xyz
        hand_verts = hand_verts * cube_size.unsqueeze(1) / 2 + center3d.unsqueeze(1)
        hand_joints = hand_joints * cube_size.unsqueeze(1) / 2 + center3d.unsqueeze(1)
        img, center2d, M, _ = self._depth_crop(hand_verts, center3d, cube_size)
        joint_uvd = self.JointTrans(hand_joints, M, center2d, cube_size)
        joint_xyz = (hand_joints - center3d.unsqueeze(1)) / cube_size.unsqueeze(1) * 2
        mesh_xyz = (hand_verts - center3d.unsqueeze(1)) / cube_size.unsqueeze(1) * 2
        return img, joint_uvd, joint_xyz, mesh_xyz

    def normal_render(self, model_paras, center3d, cube_size):
        hand_verts, hand_joints = self.mano_layer.get_mano_vertices_packed(_rows62(model_paras), global_scale=1 / 125)
        hand_verts = (hand_verts + 1) / 2 * cube_size.unsqueeze(1) + center3d.unsqueeze(1)
        hand_joints = (hand_joints + 1) / 2 * cube_size.unsqueeze(1) + center3d.unsqueeze(1)
        img, center2d, M, _ = self._depth_crop(hand_verts, center3d, cube_size)
        joint_uvd = self.JointTrans(hand_joints, M, center2d, cube_size)
        joint_xyz = (hand_joints - center3d.unsqueeze(1)) / cube_size.unsqueeze(1) * 2
        verts_xyz = (hand_verts - center3d.unsqueeze(1)) / cube_size.unsqueeze(1) * 2
        return img, joint_uvd, joint_xyz, verts_xyz

    def M_render(self, model_paras, center3d, cube_size, M=None, mask=True):
        quat, theta, beta, cam = self._split(model_paras)
        hand_verts, hand_joints = self.mano_layer.get_mano_vertices(quat, theta, beta, cam)
        img, center2d, M, _ = self._depth_crop(hand_verts, center3d, cube_size, M=M)
        if mask:
            img = self.mask_img(img, self.JointTrans(hand_joints, M, center2d, cube_size), 0.15, 0.3)
        return img

    def get_mesh_xyz(self, model_paras):
        hand_mesh, hand_joints = self.mano_layer.get_mano_vertices_packed(_rows62(model_paras), global_scale=1 / 125)
        return hand_joints, hand_mesh

    def mesh2img(self, hand_mesh, center3d, cube_size):
        return self._depth_crop(hand_mesh, center3d, cube_size)[0]

    def getDepth(self, hand_verts, hand_joints, center3d, cube_size, M, rot=None):
        if rot is not None:
            hand_verts, hand_joints = RotationPoints(hand_verts, hand_joints, center3d, rot)
        img, center2d, M, _ = self._depth_crop(hand_verts, center3d, cube_size, M=M)
        return img, self.JointTrans(hand_joints, M, center2d, cube_size)

    # ---- small host-visible helpers with the reference's names ------------------------------------
    def comToBounds(self, com, size):
        fx, fy, fu, fv = self.paras
        zstart = com[:, 2] - size[:, 2] / 2.
        zend = com[:, 2] + size[:, 2] / 2.
        lo = lambda c, s, f: torch.floor((c * com[:, 2] / f - s / 2.) / com[:, 2] * f + 0.5).int()
        hi = lambda c, s, f: torch.floor((c * com[:, 2] / f + s / 2.) / com[:, 2] * f + 0.5).int()
        return lo(com[:, 0], size[:, 0], fx), hi(com[:, 0], size[:, 0], fx), lo(com[:, 1], size[:, 1], fy), \
            hi(com[:, 1], size[:, 1], fy), zstart, zend

    def Offset2Trans(self, xstart, xend, ystart, yend):
        c0, c1 = self.crop_size
        wb, hb = xend - xstart, yend - ystart
        wide = wb.gt(hb)
        sz0 = torch.where(wide, torch.full_like(wb, c0), (wb * c0 / hb).int())
        sz1 = torch.where(wide, (hb * c0 / wb).int(), torch.full_like(wb, c1))
        s = torch.where(wide, c0 / wb, c1 / hb)
        ox = torch.floor(c0 / 2. - sz0 / 2.).int().float()
        oy = torch.floor(c1 / 2. - sz1 / 2.).int().float()
        M = torch.zeros(wb.size(0), 3, 3, device=wb.device)
        M[:, 0, 0] = s
        M[:, 1, 1] = s
        M[:, 2, 2] = 1
        M[:, 0, 2] = s * (-xstart).float() + ox
        M[:, 1, 2] = s * (-ystart).float() + oy
        return M

    def resize(self, img):
        """(B,1,S,S) raster -> (B,1,H,W) by the nearest row table."""
        return img.index_select(2, self.resize_rowmap.long())[..., :self.img_size[0]]

    def affine_grid(self, img, M):
        b, _, h_ori, w_ori = img.size()
        h, w = self.crop_size
        src = torch.matmul(torch.inverse(M).view(b, 1, 3, 3), self.crop_mesh.expand(b, -1, -1).unsqueeze(-1))
        src = src.squeeze(-1)[:, :, 0:2]
        return ((src / torch.tensor([w_ori, h_ori], device=img.device, dtype=src.dtype)) * 2 - 1).view(b, h, w, 2)

    def warpPerspective(self, img, M):
        return F.grid_sample(img, self.affine_grid(img, M), mode='nearest', align_corners=False)

    def normalize_img(self, imgD, com, cube):
        z = com[:, 2].view(-1, 1, 1, 1)
        half = cube[:, 2].view(-1, 1, 1, 1) / 2.
        out = torch.where((imgD == -1) | (imgD == 0), z + half, imgD)
        out = torch.minimum(torch.maximum(out, z - half), z + half)
        return (out - z) / half

    def JointTrans(self, joint, M, com, cube):
        return ops.XyzToUvd.apply(joint, com, M, cube, self.cam, self.crop_size[0], True)

    def points3DToImg(self, joint_xyz):
        fx, fy, fu, fv = self.paras
        u = joint_xyz[..., 0] * fx / (joint_xyz[..., 2] + 1e-8) + fu
        v = joint_xyz[..., 1] * fy / joint_xyz[..., 2] + fv
        return torch.stack([u, v, joint_xyz[..., 2]], -1)

    def pointsImgTo3D(self, point_uvd):
        fx, fy, fu, fv = self.paras
        x = (point_uvd[..., 0] - fu) * point_uvd[..., 2] / fx
        y = (point_uvd[..., 1] - fv) * point_uvd[..., 2] / fy
        return torch.stack([x, y, point_uvd[..., 2]], -1)

    def mask_img(self, img, img_joint, mask_offset, mask_para, min_mask_num=3, max_mask_num=10, draws=None):
        """Random occluding spheres in (u,v,d) space (reference :1326-1340).  ``draws`` =
        (joint_id list, uvd_offset (B,k,3), radius (B,k)) makes the random draw an explicit input."""
        b, j, _ = img_joint.size()
        if draws is None:
            k = int(np.random.choice(np.arange(min_mask_num, max_mask_num), 1)[0])
            joint_id = np.random.choice(np.arange(0, j), k, replace=False)
            offset = ((torch.rand(b, k, 3) - 0.5) * mask_offset * 2).to(img.device)
            radius = torch.rand(b, k, device=img.device) * mask_para
        else:
            joint_id, offset, radius = draws
        centre = img_joint[:, joint_id, :] + offset                                   # (B,k,3)
        pix = torch.cat((self.xy_mesh.expand(b, -1, -1), img.reshape(b, -1, 1)), dim=-1)  # (B,HW,3)
        d2 = ((pix.unsqueeze(1) - centre.unsqueeze(2)) ** 2).sum(-1)
        hit = (d2.sqrt() < radius.unsqueeze(-1)).any(1).view(b, 1, img.size(-2), img.size(-1))
        return torch.where(hit, torch.ones_like(img), img)
