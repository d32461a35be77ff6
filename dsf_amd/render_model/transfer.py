"""Frozen Consis-CycleGAN generator used to move synthetic renders to the real-depth domain
(counterpart of the one constructor the trainer calls,
``define_G(1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier')``,
/root/reference/render_model/transfer.py:197-238, 393-448, 547-604).  Inference-only dense
convolutions -> PyTorch-ROCm / MIOpen.  Discriminators, GAN losses and U-Nets are out of scope."""
import functools

import torch.nn as nn
from torch.nn import init

from .. import nn_conv

_L = {"conv": nn_conv.Conv2d, "convT": nn_conv.ConvTranspose2d}


class ResnetBlock(nn.Module):
    def __init__(self, dim, norm_layer, use_bias):
        super().__init__()
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), _L["conv"](dim, dim, kernel_size=3, padding=0, bias=use_bias), norm_layer(dim), nn.ReLU(True),
            nn.ReflectionPad2d(1), _L["conv"](dim, dim, kernel_size=3, padding=0, bias=use_bias), norm_layer(dim))

    def forward(self, x):
        return x + self.conv_block(x)


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
        return self.model(x)


def define_G(input_nc, output_nc, ngf, netG, norm='batch', use_dropout=False, init_type='normal', init_gain=0.02, gpu_ids=[],
             native=True):
    _L["conv"], _L["convT"] = (nn_conv.Conv2d, nn_conv.ConvTranspose2d) if native else (nn.Conv2d, nn.ConvTranspose2d)
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
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == 'normal':
                init.normal_(m.weight.data, 0.0, init_gain)
            else:
                raise NotImplementedError(init_type)
            if getattr(m, 'bias', None) is not None:
                init.constant_(m.bias.data, 0.0)
        elif 'BatchNorm2d' in name:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)

    net.apply(init_func)
    return net
