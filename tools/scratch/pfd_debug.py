import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import torch
    from dsf_amd.render_model.mano_layer import Render
    from dsf_amd.metric.meshLoss import ICPLoss, JointICPLoss
    from dsf_amd.train_step import synthetic_batch
    B, P, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
    mano = render.mano_layer
    p, c, cube = synthetic_batch(B, "cuda", seed=3)
    with torch.no_grad():
        jx, mesh = render.get_mesh_xyz(p)
        g = torch.Generator(device="cuda").manual_seed(0)
        idx = torch.randint(0, 779, (B, P), device="cuda", generator=g)
        pcl = (torch.gather(mesh, 1, idx[..., None].expand(-1, -1, 3)) + 0.02 * torch.randn(B, P, 3, device="cuda", generator=g)).contiguous()
        torch.cuda.synchronize(); print("inputs ok", flush=True)
        if mode == "icp":
            d = ICPLoss(mesh, pcl, mano.faces)
        else:
            seg = mano.seg_pcl(jx, jx, mesh, pcl)
            torch.cuda.synchronize(); print("seg ok", flush=True)
            d = JointICPLoss(mesh, pcl, mano.joint_faces, seg)
        torch.cuda.synchronize()
        print("B %d P %d %s ok: %s" % (B, P, mode, d.flatten()[:3].tolist()), flush=True)
else:
    for B, P, mode in ((1, 256, "icp"), (1, 100, "icp"), (2, 2048, "icp"), (1, 5000, "icp"), (1, 256, "part"), (2, 2048, "part")):
        r = subprocess.run([sys.executable, __file__, str(B), str(P), mode], capture_output=True, text=True, timeout=300)
        print(B, P, mode, "rc", r.returncode, "|", r.stdout.strip().replace("\n", " / "), "|", r.stderr.strip().splitlines()[-1][:150] if r.returncode else "")
