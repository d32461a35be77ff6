"""discriminators, GAN losses, WGAN-GP gradient penalty (double backward through the HIP convolutions)
and the Consis-CycleGAN training step against the same composition on plain torch.nn twins on the CPU (the reference's
render_model/transfer.py IS plain torch; oracle.nets twins are bit-identical to it, tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(build, seed):
    from oracle import nets
    torch.manual_seed(seed)
    cpu = nets.build(build)
    gpu = build().cuda()
    gpu.load_state_dict(cpu.state_dict())
    return cpu, gpu


def _grad_cos(cpu, gpu):
    a = torch.cat([p.grad.flatten().double() for p in cpu.parameters() if p.grad is not None])
    b = torch.cat([p.grad.cpu().flatten().double() for p in gpu.parameters() if p.grad is not None])
    assert a.numel() == b.numel() and a.numel() > 0
    return float((a * b).sum() / (a.norm() * b.norm())), float((a - b).norm() / a.norm())


@pytest.mark.parametrize("netD,norm", [("basic", "instance"), ("pixel", "batch"), ("n_layers", "batch")])
def test_gradient_penalty_double_backward_vs_torch(netD, norm):
    """cal_gradient_penalty (reference transfer.py:356-391): value, the input gradients it returns, and the gradient of the
    penalty w.r.t. every discriminator parameter (second order: backward of the HIP convolutions' backward)."""
    from dsf_amd.render_model.transfer import define_D, cal_gradient_penalty
    cpu, gpu = _pair(lambda: define_D(1, 16, netD, 2, norm, 'normal', 0.2), 3)
    g = torch.Generator().manual_seed(4)
    real, fake = torch.rand(3, 1, 64, 64, generator=g) * 2 - 1, torch.rand(3, 1, 64, 64, generator=g) * 2 - 1
    alpha = torch.rand(3, 1, generator=g)
    gp_c, gr_c = cal_gradient_penalty(cpu, real, fake, 'cpu', 'mixed', 1.0, 10.0, alpha=alpha)
    gp_c.backward()
    gp_g, gr_g = cal_gradient_penalty(gpu, real.cuda(), fake.cuda(), 'cuda', 'mixed', 1.0, 10.0, alpha=alpha.cuda())
    gp_g.backward()
    assert abs(float(gp_g) - float(gp_c)) <= 2e-3 * abs(float(gp_c))
    assert (gr_g.detach().cpu() - gr_c.detach()).abs().max() <= 2e-3 * gr_c.detach().abs().max()
    cos, rel = _grad_cos(cpu, gpu)
    assert cos > 0.9995 and rel < 2e-2, (cos, rel)


GP_CASES = (("basic", (1, 8, "basic", 3, "instance", "normal", 0.2), "mixed"), ("nl2", (1, 8, "n_layers", 2, "batch", "normal", 0.2), "mixed"),
            ("pixel", (1, 8, "pixel", 3, "batch", "normal", 0.2), "mixed"),
            ("basic_real", (1, 8, "basic", 3, "instance", "normal", 0.2), "real"),
            ("basic_fake", (1, 8, "basic", 3, "instance", "normal", 0.2), "fake"))        # as tests/golden/make_golden_gp.py


@pytest.mark.parametrize("case", range(5))
def test_gradient_penalty_vs_reference_golden(case):
    """The HIP discriminators through the product's ``cal_gradient_penalty`` against the REFERENCE's own call of its
    ``cal_gradient_penalty`` (render_model/transfer.py:356-391; tests/golden/reference_gp.npz, mixing draw recorded):
    value and returned input gradients within 2e-3, second-order parameter gradients cosine > 0.9995."""
    import os
    from dsf_amd.render_model.transfer import define_D, cal_gradient_penalty
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_gp.npz"))
    tag, args, kind = GP_CASES[case]
    _, gpu = _pair(lambda: define_D(*args), int(g["seed_net"]))
    real, fake = torch.tensor(g["real"]).cuda(), torch.tensor(g["fake"]).cuda()
    gp, grads = cal_gradient_penalty(gpu, real, fake, 'cuda', kind, 1.0, 10.0, alpha=torch.tensor(g[tag + "_alpha"]).cuda())
    gp.backward()
    ref = float(g[tag + "_gp"])
    assert abs(float(gp) - ref) <= 2e-3 * abs(ref), (float(gp), ref)
    gr = g[tag + "_grads"]
    assert np.abs(grads.detach().cpu().numpy() - gr).max() <= 2e-3 * np.abs(gr).max()
    if kind == "mixed":
        a = torch.tensor(g[tag + "_param_grads"]).double()
        b = torch.cat([p.grad.cpu().flatten().double() for p in gpu.parameters() if p.grad is not None])
        assert a.numel() == b.numel()
        cos, rel = float((a * b).sum() / (a.norm() * b.norm())), float((a - b).norm() / a.norm())
        assert cos > 0.9995 and rel < 2e-2, (cos, rel)


@pytest.mark.parametrize("gan_mode", ["lsgan", "wgangp"])
def test_consis_cyclegan_step_vs_torch(gan_mode):
    from dsf_amd.render_model.transfer import define_G, define_D
    from dsf_amd.transfer_step import ConsisCycleGANStep
    mk_g = lambda: define_G(1, 1, 16, 'resnet_6blocks', 'instance', False, 'xavier')
    mk_d = lambda: define_D(1, 16, 'basic', 3, 'instance', 'normal', 0.02)
    nets_c, nets_g = [], []
    for i, mk in enumerate((mk_g, mk_g, mk_d, mk_d)):
        c, gnet = _pair(mk, 10 + i)
        nets_c.append(c); nets_g.append(gnet)
    g = torch.Generator().manual_seed(5)
    real_A = (torch.rand(2, 1, 64, 64, generator=g) * 2 - 1).clamp(max=1.0)
    real_A[:, :, :20] = 1.0                                              # background rows
    real_B = (torch.rand(2, 1, 64, 64, generator=g) * 2 - 1)
    real_B[:, :, 40:] = 1.0
    alphas = (torch.rand(2, 1, generator=g), torch.rand(2, 1, generator=g))
    sc = ConsisCycleGANStep(*nets_c, gan_mode=gan_mode)
    sg = ConsisCycleGANStep(*nets_g, gan_mode=gan_mode)
    # losses and gradients of one step
    fwd_c = sc.forward(real_A, real_B)
    fwd_g = sg.forward(real_A.cuda(), real_B.cuda())
    for a, b in zip(fwd_c, fwd_g):
        assert (a - b.cpu()).abs().max() < 2e-3
    lc, tc = sc.loss_G(real_A, real_B, fwd_c)
    lg, tg = sg.loss_G(real_A.cuda(), real_B.cuda(), fwd_g)
    assert set(tc) == {"idt_A", "idt_B", "G_A", "G_B", "cycle_A", "cycle_B", "consis_A", "consis_B"}
    for k in tc:
        assert abs(float(tc[k]) - float(tg[k])) <= 2e-3 * abs(float(tc[k])) + 1e-5, k
    lc.backward(); lg.backward()
    for c, gnet in zip(nets_c[:2], nets_g[:2]):
        cos, rel = _grad_cos(c, gnet)
        assert cos > 0.9995 and rel < 2e-2, (cos, rel)
    for n in nets_c + nets_g:
        n.zero_grad(set_to_none=True)
    dc = sc.loss_D(nets_c[2], real_B, fwd_c[0], alphas[0])
    dg = sg.loss_D(nets_g[2], real_B.cuda(), fwd_g[0], alphas[0].cuda())
    assert abs(float(dc) - float(dg)) <= 2e-3 * abs(float(dc)) + 1e-5
    dc.backward(); dg.backward()
    cos, rel = _grad_cos(nets_c[2], nets_g[2])
    assert cos > 0.9995 and rel < 2e-2, (cos, rel)
    # whole steps run and move every network
    before = [p.detach().clone() for p in nets_g[0].parameters()]
    for _ in range(2):
        loss, terms = sg(real_A.cuda(), real_B.cuda(), (alphas[0].cuda(), alphas[1].cuda()))
    assert torch.isfinite(loss) and all(torch.isfinite(v) for v in terms.values())
    assert any((p.detach() != q).any() for p, q in zip(nets_g[0].parameters(), before))


@pytest.mark.parametrize("B,C,H,pad,res,relu", [(3, 64, 32, 3, False, True), (2, 256, 16, 1, True, False), (5, 128, 24, 1, False, False),
                                                (2, 1, 20, 3, False, False), (1, 1024, 6, 2, True, True)])
def test_instance_norm_and_reflect_pad_kernels_vs_torch(B, C, H, pad, res, relu):
    """dsf_instnorm_forward / dsf_reflect_pad_nhwc (the frozen generator's inference passes, reference transfer.py:393-448)
    against torch's InstanceNorm2d / ReflectionPad2d on the CPU."""
    from dsf_amd import nn_norm
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, H + 3, generator=g) * 2 + 0.7
    r = torch.randn(B, C, H, H + 3, generator=g) if res else None
    assert torch.equal(nn_norm.reflect_pad(x.cuda(), pad).cpu(), torch.nn.ReflectionPad2d(pad)(x))        # a gather: bit-equal
    if C >= 4:
        ref = torch.nn.InstanceNorm2d(C)(x)
        if res:
            ref = ref + r
        if relu:
            ref = torch.relu(ref)
        got = nn_norm.instance_norm_act(x.cuda(), r.cuda() if res else None, relu).cpu()
        assert (got - ref).abs().max() < 1e-5 * max(1.0, float(ref.abs().max()))


def test_frozen_generator_fused_inference_equals_the_module_path():
    """The no-grad pass of the transfer generator (what Trainer.Pretrain / FinetuneStage run, train_render.py:428-435) on the
    HIP padding / instance-norm passes against the same network through torch's modules, and against the CPU twin."""
    from dsf_amd.render_model import transfer as T
    cpu, gpu = _pair(lambda: T.define_G(1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier'), 8)
    cpu.eval(); gpu.eval()
    x = torch.rand(3, 1, 128, 128, generator=torch.Generator().manual_seed(9)) * 2 - 1
    with torch.no_grad():
        ref = cpu(x)
        fused = gpu(x.cuda()).cpu()
        T.FUSED_INFERENCE[0] = False
        try:
            plain = gpu(x.cuda()).cpu()
        finally:
            T.FUSED_INFERENCE[0] = True
    assert fused.shape == ref.shape == (3, 1, 128, 128)
    assert (fused - plain).abs().max() < 2e-4 and (fused - ref).abs().max() < 2e-3
    # with autograd on the generator runs through the modules (its training step, ConsisCycleGANStep, needs the backward pass)
    y = gpu(x.cuda().requires_grad_(True))
    y.mean().backward()
