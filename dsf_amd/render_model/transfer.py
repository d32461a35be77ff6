"""Frozen Consis-CycleGAN generator used to move synthetic renders to the real-depth domain
(counterpart of the one constructor the trainer calls,
``define_G(1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier')``,
/root/reference/render_model/transfer.py:197-238, 393-448, 547-604) on the HIP convolutions, plus the pieces its
training needs: ``define_D`` / ``NLayerDiscriminator`` / ``PixelDiscriminator`` (:240-284, 709-786) and
``GANLoss`` (:287-353).  U-Nets and the encoder / decoder split are not used by the trainer and not provided."""
import functools

import torch
import torch.nn as nn
from torch.nn import init

from .. import nn_conv



class _Factory:
    """layer classes of the network under construction (the registry ``nn_conv.LAYERS``)"""

    def __getitem__(self, k):
        return nn_conv.LAYERS["Conv2d" if k == "conv" else "ConvTranspose2d"]


_L = _Factory()
import os
FUSED_INFERENCE = [os.environ.get("DSF_GEN_FUSED", "1") == "1"]    # the no-grad generator pass on the HIP padding / instance-norm kernels (False: torch modules)


class ResnetBlock(nn.Module):
    def __init__(self, dim, norm_layer, use_bias):
        super().__init__()
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), _L["conv"](dim, dim, kernel_size=3, padding=0, bias=use_bias), norm_layer(dim), nn.ReLU(True),
            nn.ReflectionPad2d(1), _L["conv"](dim, dim, kernel_size=3, padding=0, bias=use_bias), norm_layer(dim))

    def forward(self, x):
        return x + self.conv_block(x)


def _fusable_norm(m):
    return isinstance(m, nn.InstanceNorm2d) and not m.affine and not m.track_running_stats


def _fused_inference(seq, x, residual=None, pool=None):
    """``seq(x)`` (+ residual behind its last layer) for a stack of [ReflectionPad2d, conv, InstanceNorm2d, ReLU, ResnetBlock, Tanh]
    layers on the HIP inference passes: padding and instance normalisation (+ skip)(+ ReLU) stay channels_last and run as one and
    two launches (nn_norm.reflect_pad / instance_norm_act) where torch needs NCHW kernels between channels_last convolutions
    (config 5: 87 layout copies, 23 two-kernel norms and 20 pads per generator pass = 12 of 97 ms per step).  Used when nothing
    requires a gradient (the generator is frozen inside the trainer steps, train_render.py:428-435); same arithmetic, fp32."""
    from .. import nn_norm
    mods = list(seq)
    if pool is None:
        # one zero fill for the per-sample statistics of every instance norm of the pass (B x 2C doubles each; <= 256 channels here)
        n_norm = sum(1 for m in seq.modules() if _fusable_norm(m))
        pool = [torch.zeros(n_norm * x.shape[0] * 2 * 256, device=x.device, dtype=torch.float64), 0]

    def take(n):
        if pool[1] + n > pool[0].numel():
            return None
        pool[1] += n
        return pool[0][pool[1] - n:pool[1]]
    i = 0
    while i < len(mods):
        m = mods[i]
        last = i == len(mods) - 1
        if isinstance(m, nn.ReflectionPad2d) and len(set(m.padding)) == 1:
            x = nn_norm.reflect_pad(x, m.padding[0])
        elif _fusable_norm(m) and nn_norm.supported(x.shape[1]):
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            tail = last or (relu and i + 2 == len(mods))
            x = nn_norm.instance_norm_act(x, residual if tail else None, relu, m.eps, acc=take(x.shape[0] * 2 * x.shape[1]))
            if tail:
                residual = None
            i += 1 if relu else 0
        elif isinstance(m, ResnetBlock):
            x = _fused_inference(m.conv_block, x, residual=x, pool=pool)
        else:
            x = m(x)
        i += 1
    return x if residual is None else x + residual


class ResnetGenerator(nn.Module):
    def __init__(self, input_nc, output_nc, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=False, n_blocks=6,
                 padding_type='reflect'):
        super().__init__()
        assert padding_type == 'reflect' and not use_dropout
        fn = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
        bias = fn == nn.InstanceNorm2d
        seq = [nn.ReflectionPad2d(3), _L["conv"](input_nc, ngf, kernel_size=7, padding=0, bias=bias), norm_layer(ngf), nn.ReLU(True)]
        ch = ngf
        for _ in range(2):
            seq += [_L["conv"](ch, ch * 2, kernel_size=3, stride=2, padding=1, bias=bias), norm_layer(ch * 2), nn.ReLU(True)]
            ch *= 2
        seq += [ResnetBlock(ch, norm_layer, bias) for _ in range(n_blocks)]
        for _ in range(2):
            seq += [_L["convT"](ch, ch // 2, kernel_size=3, stride=2, padding=1, output_padding=1, bias=bias),
                    norm_layer(ch // 2), nn.ReLU(True)]
            ch //= 2
        seq += [nn.ReflectionPad2d(3), _L["conv"](ngf, output_nc, kernel_size=7, padding=0), nn.Tanh()]
        self.model = nn.Sequential(*seq)

    def forward(self, x):
        if x.is_cuda and not torch.is_grad_enabled() and x.dtype == torch.float32 and FUSED_INFERENCE[0]:
            return _fused_inference(self.model, x)
        return self.model(x)


def define_G(input_nc, output_nc, ngf, netG, norm='batch', use_dropout=False, init_type='normal', init_gain=0.02, gpu_ids=[]):
    if norm == 'instance':
        norm_layer = functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=False)
    elif norm == 'batch':
        norm_layer = functools.partial(nn.BatchNorm2d, affine=True, track_running_stats=True)
    else:
        raise NotImplementedError(norm)
    blocks = {'resnet_9blocks': 9, 'resnet_6blocks': 6}
    if netG not in blocks:
        raise NotImplementedError('only the resnet generators are on the hot path (got %s)' % netG)
    net = ResnetGenerator(input_nc, output_nc, ngf, norm_layer=norm_layer, use_dropout=use_dropout, n_blocks=blocks[netG])

    def init_func(m):
        name = m.__class__.__name__
        if hasattr(m, 'weight') and ('Conv' in name or 'Linear' in name):
            if init_type == 'xavier':
                init.xavier_normal_(m.weight, gain=init_gain)
            elif init_type == 'normal':
                init.normal_(m.weight, 0.0, init_gain)
            else:
                raise NotImplementedError(init_type)
            if getattr(m, 'bias', None) is not None:
                init.constant_(m.bias, 0.0)
        elif 'BatchNorm2d' in name:
            init.normal_(m.weight, 1.0, init_gain)
            init.constant_(m.bias, 0.0)

    net.apply(init_func)            # torch.nn.init writes in place on the parameters (version counters see it)
    nn_conv.weights_changed()
    return net


def _norm_layer(norm):
    if norm == 'instance':
        return functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=False)
    if norm == 'batch':
        return functools.partial(nn.BatchNorm2d, affine=True, track_running_stats=True)
    if norm == 'none':
        return lambda c: nn.Identity()
    raise NotImplementedError(norm)


def _init_net(net, init_type, init_gain):
    def init_func(m):
        name = m.__class__.__name__
        if hasattr(m, 'weight') and ('Conv' in name or 'Linear' in name):
            if init_type == 'xavier':
                init.xavier_normal_(m.weight, gain=init_gain)
            elif init_type == 'normal':
                init.normal_(m.weight, 0.0, init_gain)
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                init.orthogonal_(m.weight, gain=init_gain)
            else:
                raise NotImplementedError(init_type)
            if getattr(m, 'bias', None) is not None:
                init.constant_(m.bias, 0.0)
        elif 'BatchNorm2d' in name:
            init.normal_(m.weight, 1.0, init_gain)
            init.constant_(m.bias, 0.0)
    net.apply(init_func)            # torch.nn.init writes in place on the parameters (version counters see it)
    nn_conv.weights_changed()
    return net


class NLayerDiscriminator(nn.Module):
    """PatchGAN discriminator (reference :709-755): k4 convolutions, LeakyReLU(0.2), stride 2 for the first n_layers."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d):
        super().__init__()
        fn = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
        use_bias = fn == nn.InstanceNorm2d
        C = _L["conv"]
        seq = [C(input_nc, ndf, kernel_size=4, stride=2, padding=1), nn.LeakyReLU(0.2, True)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            seq += [C(ndf * prev, ndf * mult, kernel_size=4, stride=2, padding=1, bias=use_bias), norm_layer(ndf * mult),
                    nn.LeakyReLU(0.2, True)]
        prev, mult = mult, min(2 ** n_layers, 8)
        seq += [C(ndf * prev, ndf * mult, kernel_size=4, stride=1, padding=1, bias=use_bias), norm_layer(ndf * mult),
                nn.LeakyReLU(0.2, True)]
        seq += [C(ndf * mult, 1, kernel_size=4, stride=1, padding=1)]
        self.model = nn.Sequential(*seq)

    def forward(self, input):
        return self.model(input)


class PixelDiscriminator(nn.Module):
    """1x1 PatchGAN (reference :757-786)."""

    def __init__(self, input_nc, ndf=64, norm_layer=nn.BatchNorm2d):
        super().__init__()
        fn = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
        use_bias = fn == nn.InstanceNorm2d
        C = _L["conv"]
        self.net = nn.Sequential(C(input_nc, ndf, kernel_size=1, stride=1, padding=0), nn.LeakyReLU(0.2, True),
                                 C(ndf, ndf * 2, kernel_size=1, stride=1, padding=0, bias=use_bias), norm_layer(ndf * 2),
                                 nn.LeakyReLU(0.2, True), C(ndf * 2, 1, kernel_size=1, stride=1, padding=0, bias=use_bias))

    def forward(self, input):
        return self.net(input)


def define_D(input_nc, ndf, netD, n_layers_D=3, norm='batch', init_type='normal', init_gain=0.02, gpu_ids=[]):
    """reference :240-284."""
    norm_layer = _norm_layer(norm)
    if netD == 'basic':
        net = NLayerDiscriminator(input_nc, ndf, n_layers=3, norm_layer=norm_layer)
    elif netD == 'n_layers':
        net = NLayerDiscriminator(input_nc, ndf, n_layers_D, norm_layer=norm_layer)
    elif netD == 'pixel':
        net = PixelDiscriminator(input_nc, ndf, norm_layer=norm_layer)
    else:
        raise NotImplementedError('Discriminator model name [%s] is not recognized' % netD)
    return _init_net(net, init_type, init_gain)


class GANLoss(nn.Module):
    """The three adversarial objectives of the reference's ``GANLoss`` (:287-353) behind its call signature
    ``criterion(prediction, target_is_real)``: 'lsgan' = mean squared error against the label, 'vanilla' = binary cross
    entropy on logits against the label, 'wgangp' = the critic's signed mean.  The labels are buffers ``real_label`` /
    ``fake_label`` (they follow ``.to(device)`` and appear in the state dict in that order, as in the reference)."""

    _CRITERIA = {'lsgan': nn.MSELoss, 'vanilla': nn.BCEWithLogitsLoss, 'wgangp': None}

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        if gan_mode not in self._CRITERIA:
            raise NotImplementedError('gan mode %s not implemented' % gan_mode)
        for name, value in (('real_label', target_real_label), ('fake_label', target_fake_label)):
            self.register_buffer(name, torch.tensor(value))
        self.gan_mode = gan_mode
        criterion = self._CRITERIA[gan_mode]
        self.loss = criterion() if criterion is not None else None

    def get_target_tensor(self, prediction, target_is_real):
        return (self.real_label if target_is_real else self.fake_label).expand_as(prediction)

    def __call__(self, prediction, target_is_real):
        if self.loss is None:                                           # wgangp: maximise the critic on real, minimise on fake
            return prediction.mean() * (-1.0 if target_is_real else 1.0)
        return self.loss(prediction, self.get_target_tensor(prediction, target_is_real))


def _penalty_points(real_data, fake_data, kind, alpha, device):
    """where the critic's input gradient is taken: the real batch, the fake batch, or one random point per sample on the
    segment between them (``alpha`` (B,1) in [0,1): the weight of the real sample)"""
    if kind == 'real':
        return real_data
    if kind == 'fake':
        return fake_data
    if kind != 'mixed':
        raise NotImplementedError('{} not implemented'.format(kind))
    n = real_data.shape[0]
    w = torch.rand(n, 1, device=device) if alpha is None else alpha
    w = w.reshape((n,) + (1,) * (real_data.dim() - 1))                  # one weight per sample, broadcast over its elements
    return w * real_data + (1 - w) * fake_data


def cal_gradient_penalty(netD, real_data, fake_data, device, type='mixed', constant=1.0, lambda_gp=10.0, alpha=None):
    """WGAN-GP gradient penalty with the reference's signature and return pair (:356-391):
    ``lambda_gp * mean_b (||d netD(x_b) / d x_b + 1e-16||_2 - constant)^2`` at real / fake / mixed points, and the per-sample
    gradients ``(B, -1)``; ``(0.0, None)`` when ``lambda_gp <= 0``.  The input gradient is taken with ``create_graph=True`` so
    that the penalty can be differentiated with respect to the critic's parameters -- the HIP convolutions provide the
    backward of their backward (``nn_conv.Conv2dFunction.backward`` re-expresses itself through differentiable Functions
    when autograd asks).  ``alpha`` (B,1): the mixing draw of type 'mixed' as an explicit input; drawn here
    when None.  Pinned to the reference's own call by tests/golden/reference_gp.npz."""
    if not lambda_gp > 0.0:
        return 0.0, None
    x = _penalty_points(real_data, fake_data, type, alpha, device)
    x.requires_grad_(True)
    score = netD(x)
    grad_x, = torch.autograd.grad(score, x, grad_outputs=torch.ones_like(score), create_graph=True, retain_graph=True)
    per_sample = grad_x.reshape(real_data.size(0), -1)
    excess = torch.linalg.vector_norm(per_sample + 1e-16, ord=2, dim=1) - constant
    return (excess * excess).mean() * lambda_gp, per_sample
