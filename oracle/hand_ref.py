"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by dsf_amd/).

CPU fp32 restatement (torch-CPU tensor arithmetic, autograd gives the gradient
oracle) of the hand-model part of the hot path.  Pinned against the reference
itself: tests/test_oracle_golden.py compares every function here with
tests/golden/reference_golden.npz, which tests/golden/make_golden.py produced
by importing /root/reference/render_model/mano_layer.py in the build container.

Each function cites the reference lines it follows.
"""
import numpy as np
import torch

WRIST_RING = [121, 214, 215, 279, 239, 234, 92, 38, 122, 118, 117, 119, 120, 108, 79, 78]
TIP_VERTS = [333, 444, 672, 555, 744]
BONE_CHILD = [2, 3, 16, 5, 6, 17, 8, 9, 18, 11, 12, 19, 14, 15, 20]
N_PALM = 21       # 1 + 5*4 spheres
N_FINGER = 45     # 15 bones * 3 spheres


class HandModel:
    """Buffers of MANO_SMPL.__init__ (mano_layer.py:83-269) from a MANO dict."""

    def __init__(self, d):
        f32 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), dtype=torch.float32)
        base_faces = np.asarray(d["f"]).astype(np.int64)
        cap = np.array([[WRIST_RING[i], WRIST_RING[(i + 1) % 16], 778] for i in range(16)], dtype=np.int64)
        self.faces = torch.tensor(np.concatenate([base_faces, cap], 0))              # :102-106
        self.v_template = f32(d["v_template"])                                       # (778,3)
        sd = np.asarray(d["shapedirs"], dtype=np.float64)
        self.shapedirs = f32(sd.reshape(-1, sd.shape[-1]).T)                          # (10,2334) :116-120
        jr = np.asarray(d["J_regressor"].T.toarray(), dtype=np.float64)               # (778,16)
        tips = np.zeros((778, 5))
        for k, v in enumerate(TIP_VERTS):
            tips[v, k] = 1.0
        self.J_regressor = f32(np.concatenate([jr, tips], 1))                         # (778,21) :123-132
        self.hands_comp = f32(d["hands_components"])
        self.hands_mean = f32(d["hands_mean"])
        pd = np.asarray(d["posedirs"], dtype=np.float64)
        self.posedirs = f32(pd.reshape(-1, pd.shape[-1]).T)                           # (135,2334) :142-145
        self.parents = np.asarray(d["kintree_table"])[0].astype(np.int32)             # :147
        self.weight = f32(d["weights"])                                               # (778,16)
        # joint_faces[j-1]: faces touching a vertex with weight[:, j] > 0.1  (:161-171)
        fnp = self.faces.numpy()
        w = self.weight.numpy()
        self.joint_faces, self.finger_faces = [], []
        for j in range(1, 16):
            hit = np.isin(fnp, np.nonzero(w[:, j] > 0.1)[0]).any(1)
            self.joint_faces.append(torch.tensor(fnp[hit]))
        for k in range(5):                                                           # :174-185
            vs = np.concatenate([np.nonzero(w[:, 3 * k + j] > 0.1)[0] for j in (1, 2, 3)])
            hit = np.isin(fnp, vs).any(1)
            self.finger_faces.append(torch.tensor(fnp[hit]))
        self.t_finger = torch.linspace(0, 1, 4)[:-1]                                  # :231
        self.t_palm = torch.linspace(0, 1, 6)[1:-1]                                   # :236
        self.coll_mask = torch.tensor(collision_mask())


def collision_mask():
    """66x66 pair mask, mano_layer.py:240-269, restated per (row, col) rule."""
    m = np.zeros((66, 66), dtype=np.float32)
    m[:N_PALM, N_PALM:] = 1.0
    m[N_PALM:, :] = 1.0
    for bone in range(15):
        finger = bone // 3 + 1
        rows = N_PALM + 3 * bone + np.arange(3)
        own = N_PALM + 3 * bone
        if bone % 3 == 0:
            knuckle = 4 * finger                      # last interior sphere of this finger's palm bone
            m[rows, knuckle] = 0.0
            m[knuckle, rows] = 0.0
            m[rows[:, None], np.arange(own, own + 6)[None]] = 0.0
        else:
            end = N_PALM + 9 * finger
            m[rows[:, None], np.arange(own - 3, min(own + 7, end))[None]] = 0.0
    thumb = 36
    m[N_PALM + thumb:N_PALM + thumb + 4, :N_PALM] = 0.0
    m[:N_PALM, N_PALM + thumb:N_PALM + thumb + 4] = 0.0
    return m


def quat_to_mat(q):
    """mano_layer.py:697-718 / 773-794: normalise, then the 9 quadratic forms."""
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    ww, xx, yy, zz = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    rows = [ww + xx - yy - zz, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
            2 * wz + 2 * xy, ww - xx + yy - zz, 2 * yz - 2 * wx,
            2 * xz - 2 * wy, 2 * wx + 2 * yz, ww - xx - yy + zz]
    return torch.stack(rows, 1).view(-1, 3, 3)


def rodrigues(theta):
    """mano_layer.py:720-728 (1e-8 is added INSIDE the norm, the axis is theta/angle)."""
    ang = torch.norm(theta + 1e-8, p=2, dim=1, keepdim=True)
    axis = theta / ang
    half = ang * 0.5
    return quat_to_mat(torch.cat([torch.cos(half), torch.sin(half) * axis], 1))


def kinematic_chain(Rs, Js, parents):
    """mano_layer.py:730-770. Rs (B,16,3,3), Js (B,16,3) -> posed joints (B,16,3), A (B,16,4,4)."""
    B = Rs.shape[0]
    G = [None] * 16
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0]).view(1, 1, 4).expand(B, 1, 4)

    def rigid(R, t):
        return torch.cat([torch.cat([R, t.unsqueeze(-1)], 2), bottom], 1)

    G[0] = rigid(Rs[:, 0], Js[:, 0])
    for i in range(1, 16):
        p = int(parents[i])
        G[i] = G[p] @ rigid(Rs[:, i], Js[:, i] - Js[:, p])
    G = torch.stack(G, 1)
    posed = G[:, :, :3, 3]
    Jh = torch.cat([Js, torch.zeros(B, 16, 1)], 2).unsqueeze(-1)          # (B,16,4,1)
    corr = G @ Jh                                                          # (B,16,4,1)
    A = G - torch.cat([torch.zeros(B, 16, 4, 3), corr], 3)
    return posed, A


def mano_forward(hm, beta, theta, rot):
    """MANO_SMPL.forward(get_skin=True), mano_layer.py:573-641."""
    B = beta.shape[0]
    v_shaped = (beta @ hm.shapedirs).view(B, 778, 3) + hm.v_template                    # :586
    J = torch.einsum("bvc,vj->bjc", v_shaped, hm.J_regressor)                           # :587-591
    R0 = rodrigues(rot) if rot.shape[-1] == 3 else quat_to_mat(rot)                      # :593-609
    full_pose = theta @ hm.hands_comp[:theta.shape[-1]] + hm.hands_mean
    Rs = rodrigues(full_pose.view(-1, 3)).view(B, 15, 3, 3)
    pose_feat = (Rs - torch.eye(3)).view(B, 135)                                        # :611
    v_posed = v_shaped + (pose_feat @ hm.posedirs).view(B, 778, 3)                       # :613
    _, A = kinematic_chain(torch.cat([R0.view(B, 1, 3, 3), Rs], 1), J[:, :16], hm.parents)
    T = (hm.weight @ A.view(B, 16, 16)).view(B, 778, 4, 4)                               # :619-621
    vh = torch.cat([v_posed, torch.ones(B, 778, 1)], 2).unsqueeze(-1)
    verts = (T @ vh)[:, :, :3, 0]                                                        # :623-629
    joints = torch.einsum("bvc,vj->bjc", verts, hm.J_regressor)                          # :630-633
    wrist = verts[:, WRIST_RING].mean(1, keepdim=True)                                   # :636
    return torch.cat([verts, wrist], 1), joints, Rs


def mano_vertices(hm, rot, pose, shape, cam, global_scale=None):
    """get_mano_vertices, mano_layer.py:643-693 (non-icvl branch)."""
    verts, joints, _ = mano_forward(hm, shape, pose, rot)
    scale = cam[:, 0].view(-1, 1, 1)
    trans = cam[:, 1:].view(-1, 1, 3)
    verts, joints = verts * 1000, joints * 1000
    if global_scale is not None:
        verts, joints = verts * global_scale, joints * global_scale
    return verts * scale + trans, joints * scale + trans


def sphere_set(hm, joints, mesh):
    """get_sphere_radius, mano_layer.py:271-317 -> centres (B,66,3), radii (B,66)."""
    B = joints.shape[0]
    owns = (hm.J_regressor > 0).t().unsqueeze(0)                                         # (1,21,778)
    diff = joints.unsqueeze(2) - mesh[:, None, :778, :]
    dist = torch.sqrt((diff * diff).sum(-1) + 1e-8)
    dist = torch.where(owns, dist, torch.full_like(dist, 100.0))
    r = torch.topk(dist, 10, dim=-1, largest=False)[0].mean(-1)                          # (B,21)
    r = torch.cat([r[:, :16], r[:, [3, 6, 9, 12, 15]] / 1.5], 1)
    knuckles = [1, 4, 7, 10, 13]
    r_root = torch.clamp(r[:, 0:1] - 0.05, 0.01, 0.4)
    r_palm = (r[:, knuckles] - r_root).unsqueeze(-1) * hm.t_palm.view(1, 1, -1) + r_root.unsqueeze(-1)
    r_fing = (r[:, BONE_CHILD] - r[:, 1:16]).unsqueeze(-1) * hm.t_finger.view(1, 1, -1) + r[:, 1:16].unsqueeze(-1)
    radii = torch.cat([r_root, r_palm.reshape(B, -1), r_fing.reshape(B, -1)], 1)
    c_root = joints[:, 0:1]
    c_palm = (joints[:, knuckles] - c_root).unsqueeze(2) * hm.t_palm.view(1, 1, -1, 1) + c_root.unsqueeze(2)
    c_fing = (joints[:, BONE_CHILD] - joints[:, 1:16]).unsqueeze(2) * hm.t_finger.view(1, 1, -1, 1) \
        + joints[:, 1:16].unsqueeze(2)
    centres = torch.cat([c_root, c_palm.reshape(B, -1, 3), c_fing.reshape(B, -1, 3)], 1)
    return centres, radii


def collision_loss(hm, joints, mesh):
    """calculate_coll, mano_layer.py:373-386."""
    c, r = sphere_set(hm, joints.clone(), mesh)
    d = c.unsqueeze(2) - c.unsqueeze(1)
    d = torch.sqrt((d ** 2).sum(-1) + 1e-8)
    pen = torch.clamp(r.unsqueeze(2) + r.unsqueeze(1) - d, min=0.0) * hm.coll_mask
    # NB :383 sums dim -1 twice WITH keepdim, so the second sum is over a size-1 axis:
    # the 0.1 gate is per sphere ROW (b, i), not per sample.
    gate = (pen.sum(-1, keepdim=True) < 0.1).float()
    return (pen * gate).sum(-1).mean()


def segment_points(hm, joints_pix, joints_mano, mesh, pcl):
    """seg_pcl, mano_layer.py:404-426 -> int64 labels (B,P) in 0..15."""
    c, _ = sphere_set(hm, joints_pix.clone(), mesh)
    _, r = sphere_set(hm, joints_mano.clone(), mesh)

    def shell(cs, rs):
        d = torch.sqrt(((pcl.unsqueeze(2) - cs.unsqueeze(1)) ** 2).sum(-1) + 1e-8)
        return torch.abs(d - rs.unsqueeze(1)).min(-1)

    fd, fi = shell(c[:, N_PALM:], r[:, N_PALM:])
    pd_, _ = shell(c[:, :N_PALM], r[:, :N_PALM])
    label = (fi.float() / 3).long() + 1
    return torch.where(pd_ < fd, torch.zeros_like(label), label)
