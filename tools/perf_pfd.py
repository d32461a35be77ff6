"""Point-to-mesh kernels of config 3 in isolation (B = 64, 2048-point clouds, 1554 faces): microseconds per launch and
algorithmic pair-tests per second of ICPLoss (every point against every triangle) and JointICPLoss (15 parts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.metric.meshLoss import ICPLoss, JointICPLoss
from dsf_amd.train_step import synthetic_batch
B = int(os.environ.get("B", "64"))
render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
mano = render.mano_layer
p, c, cube = synthetic_batch(B, "cuda", seed=3)
with torch.no_grad():
    jx, mesh = render.get_mesh_xyz(p)
    g = torch.Generator(device="cuda").manual_seed(0)
    idx = torch.randint(0, 779, (B, 2048), device="cuda", generator=g)
    pcl = (torch.gather(mesh, 1, idx[..., None].expand(-1, -1, 3)) + 0.02 * torch.randn(B, 2048, 3, device="cuda", generator=g)).contiguous()
    seg = mano.seg_pcl(jx, jx, mesh, pcl)
def timed(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
pairs = B * 2048 * 1554


def scenario(name, mesh):
    with torch.no_grad():
        t_icp = timed(lambda: ICPLoss(mesh, pcl, mano.faces))
        t_part = timed(lambda: JointICPLoss(mesh, pcl, mano.joint_faces, seg))
        print("%s: mean ICPLoss value %.6g, mean JointICPLoss value %.6g (a -DPFD_VARIANT=3 build returns evaluated triangles per point instead)"
              % (name, float(ICPLoss(mesh, pcl, mano.faces).mean()), float(JointICPLoss(mesh, pcl, mano.joint_faces, seg).mean())))
    print("B %d: ICPLoss forward %.1f us = %.3f T pair-tests/s (algorithmic: every point x every triangle); JointICPLoss forward %.1f us" % (B, t_icp, pairs / t_icp / 1e6, t_part))
    m2 = mesh.clone().requires_grad_(True)

    def fb():
        m2.grad = None
        (ICPLoss(m2, pcl, mano.faces).mean() + JointICPLoss(m2, pcl, mano.joint_faces, seg).mean()).backward()
    print("ICP + part ICP forward + backward: %.1f us" % timed(fb))


# the cloud hugs the mesh (a trained network), the mesh sits beside the cloud (an early one), and the mesh is a blob at the
# cube centre (a freshly initialised MANO head predicts scale ~ 0: every triangle is a near-minimiser of every point and no
# exact cull can discard any -- the state of bench.py's configs 3 and 5, whose networks are random-initialised)
scenario("cloud on the mesh", mesh)
scenario("mesh shifted by half its size", mesh + 0.5 * (mesh.amax(1, keepdim=True) - mesh.amin(1, keepdim=True)))
scenario("mesh collapsed to 1 %", mesh.mean(1, keepdim=True) + 0.01 * (mesh - mesh.mean(1, keepdim=True)))

# the labelled launch (JointICPLoss's kernel call alone) under three label distributions: the workgroups of a (sample, part) deal
# the part's groups of 64 member points out between them (round 5; round 4 gave each index range of the cloud one workgroup, and
# the labels of a real depth crop -- most points on the palm -- made the launch the latency of one workgroup: 1.4 ms in config 5)
from dsf_amd import ops
from dsf_amd.metric.meshLoss import _cached_parts
cat, first = _cached_parts(list(mano.joint_faces), mesh.device)
with torch.no_grad():
    for name, lab in (("labels of seg_pcl (cloud sampled from the vertices)", seg), ("every point in part 13", torch.full_like(seg, 13)), ("no labelled point", torch.zeros_like(seg))):
        print("labelled launch, %s: %.0f us" % (name, timed(lambda: ops.MeshPointDistance.apply(mesh, pcl, cat, first, lab, 15))))
