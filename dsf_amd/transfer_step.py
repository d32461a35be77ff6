"""Training step of the synthetic->real depth transfer network.

The reference ships only the network / loss definitions of its "Consis-CycleGAN" (render_model/transfer.py:197-391,
393-786: ``define_G``, ``define_D``, ``GANLoss``, ``cal_gradient_penalty``) and loads a generator trained elsewhere
(config.py:60-66 point at checkpoints of the public pytorch-CycleGAN-and-pix2pix code base, which is not vendored).  The
step below is that code base's published CycleGAN objective (``models/cycle_gan_model.py``: two generators, two
discriminators, GAN + cycle (lambda 10) + identity (0.5) terms, generator step with frozen discriminators, then the two
discriminator steps) on the HIP convolutions, plus the consistency term the checkpoint names advertise
("*_consis_cyclegan"): the translated image must keep the depth of the hand region of its source
(``lambda_consis * L1((G(x) - x) * [x < background])``; 0 switches it off and leaves plain CycleGAN).  With
``gan_mode='wgangp'`` the discriminator steps add the reference's gradient penalty (:356-391).
Every random draw (the penalty's mixing coefficients) is an explicit input.  A -> B is synthetic -> real.
"""
import itertools

import torch
import torch.nn.functional as F

from .render_model.transfer import GANLoss, cal_gradient_penalty


class ConsisCycleGANStep:
    def __init__(self, netG_A, netG_B, netD_A, netD_B, gan_mode='lsgan', lambda_A=10.0, lambda_B=10.0, lambda_identity=0.5,
                 lambda_consis=1.0, lambda_gp=10.0, lr=0.0002, beta1=0.5, background=0.99, optimizers=None):
        self.G_A, self.G_B, self.D_A, self.D_B = netG_A, netG_B, netD_A, netD_B
        self.gan = GANLoss(gan_mode).to(next(netG_A.parameters()).device)
        self.gan_mode = gan_mode
        self.lA, self.lB, self.lidt, self.lcon, self.lgp, self.bg = lambda_A, lambda_B, lambda_identity, lambda_consis, lambda_gp, background
        if optimizers is None:
            optimizers = (torch.optim.Adam(itertools.chain(netG_A.parameters(), netG_B.parameters()), lr=lr, betas=(beta1, 0.999)),
                          torch.optim.Adam(itertools.chain(netD_A.parameters(), netD_B.parameters()), lr=lr, betas=(beta1, 0.999)))
        self.opt_G, self.opt_D = optimizers
        from . import nn_conv, optim as _optim
        # torch.optim and FusedAdamW announce their writes (in-place ops bump the version counters / the write epoch): the split
        # weight images may be kept between uses.  Any other optimizer object (a `.data`-style update, an EMA wrapper) could
        # write behind torch's back and be served stale images, so its parameters stay unmanaged (re-split per use); such a
        # caller can opt in with nn_conv.manage_weights(...) + nn_conv.weights_changed() after its writes
        if all(isinstance(o, (torch.optim.Optimizer, _optim.FusedAdamW)) for o in optimizers):
            nn_conv.manage_weights(itertools.chain(netG_A.parameters(), netG_B.parameters(), netD_A.parameters(), netD_B.parameters()))

    @staticmethod
    def _requires_grad(nets, flag):
        for n in nets:
            for p in n.parameters():
                p.requires_grad_(flag)

    def forward(self, real_A, real_B):
        fake_B = self.G_A(real_A)
        rec_A = self.G_B(fake_B)
        fake_A = self.G_B(real_B)
        rec_B = self.G_A(fake_A)
        return fake_B, rec_A, fake_A, rec_B

    def loss_G(self, real_A, real_B, fwd=None):
        fake_B, rec_A, fake_A, rec_B = fwd if fwd is not None else self.forward(real_A, real_B)
        t = {}
        if self.lidt > 0:
            t["idt_A"] = F.l1_loss(self.G_A(real_B), real_B) * self.lB * self.lidt
            t["idt_B"] = F.l1_loss(self.G_B(real_A), real_A) * self.lA * self.lidt
        t["G_A"] = self.gan(self.D_A(fake_B), True)
        t["G_B"] = self.gan(self.D_B(fake_A), True)
        t["cycle_A"] = F.l1_loss(rec_A, real_A) * self.lA
        t["cycle_B"] = F.l1_loss(rec_B, real_B) * self.lB
        if self.lcon > 0:
            t["consis_A"] = F.l1_loss(fake_B * (real_A < self.bg), real_A * (real_A < self.bg)) * self.lcon
            t["consis_B"] = F.l1_loss(fake_A * (real_B < self.bg), real_B * (real_B < self.bg)) * self.lcon
        return sum(t.values()), t

    def loss_D(self, netD, real, fake, alpha=None):
        loss = (self.gan(netD(real), True) + self.gan(netD(fake.detach()), False)) * 0.5
        if self.gan_mode == 'wgangp' and self.lgp > 0:
            gp, _ = cal_gradient_penalty(netD, real, fake.detach(), real.device, 'mixed', 1.0, self.lgp, alpha=alpha)
            loss = loss + gp
        return loss

    def __call__(self, real_A, real_B, alphas=(None, None)):
        fwd = self.forward(real_A, real_B)
        self._requires_grad((self.D_A, self.D_B), False)                  # G step: Ds need no gradients
        self.opt_G.zero_grad(set_to_none=True)
        lg, terms = self.loss_G(real_A, real_B, fwd)
        lg.backward()
        self.opt_G.step()
        self._requires_grad((self.D_A, self.D_B), True)
        self.opt_D.zero_grad(set_to_none=True)
        ldA = self.loss_D(self.D_A, real_B, fwd[0], alphas[0])
        ldA.backward()
        ldB = self.loss_D(self.D_B, real_A, fwd[2], alphas[1])
        ldB.backward()
        self.opt_D.step()
        from . import nn_conv
        nn_conv.weights_changed()                                          # (torch.optim writes in place; belt and braces)
        terms = dict(terms, D_A=ldA.detach(), D_B=ldB.detach())
        return lg.detach(), terms
