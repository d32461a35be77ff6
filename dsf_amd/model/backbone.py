"""Dual-branch backbone (counterpart of the reference's ``model/backbone.py``): a ResNet trunk
regressing 62 MANO parameters plus a 3x-deconv pixel branch producing 84 offset/heat channels at
64x64; with ``refine=True`` a second trunk consumes the stage-1 features fused with the re-encoded
render of the stage-1 MANO estimate.  Convolutions, BatchNorm(+add+ReLU) and the stage-2 bridge
(``render.render`` + ``joint2offset``) all run on the HIP kernels of this package (csrc/conv.hip, norm.hip,
raster.hip, image_ops.hip).
Module / parameter names equal the reference's (``MANO_OCR_stage``, model/backbone.py:188-343) so
its ``latest.pth`` / ``best.pth`` load unchanged."""
import math
import os

import torch
import torch.nn as nn

from . import resnet as _resnet
from .resnet import BasicBlock, Bottleneck
from .. import nn_conv
from ..nn_norm import FusedBatchNorm2d, ConvBN, conv_bn_act, take_twin
from ..util.generateFeature import joint2offset, offset2joint_softmax
from .. import ops as _ops
from ..streams import fork

_BRIDGE_FORK = [os.environ.get("DSF_BRIDGE_FORK", "1") == "1"]
_HEAD_FORK = [os.environ.get("DSF_HEAD_FORK", "1") == "1"]     # _run_trunk: the MANO head beside the decoder (0: in front of it)
# which of the two forked chains of the stage-2 forward the HOST issues first: "bridge" (MANO head -> MANO layer -> rasteriser -> offset
# map, ~0.5 ms of host time for short launches) or "decoder" (three transposed convolutions + heads; the default: the main queue has its
# long launches before the host turns to the short ones -- config 2 16.141 -> 16.107 ms in a same-box A/B, 12 blocks of 10, config 4
# equal); see streams.fork.mark
_BRIDGE_ORDER = [os.environ.get("DSF_BRIDGE_ORDER", "decoder")]

BN_MOMENTUM = 0.1
resnet = {18: (BasicBlock, [2, 2, 2, 2]), 50: (Bottleneck, [3, 4, 6, 3]), 101: (Bottleneck, [3, 4, 23, 3]),
          152: (Bottleneck, [3, 8, 36, 3])}
N_MANO = 3 + 45 + 10 + 4


class _Layers:
    """Layer factories of one network under construction, taken from the registry ``nn_conv.LAYERS`` (the HIP
    implicit-GEMM convolutions and the fused BN kernels)."""

    def __init__(self):
        # fused BN(+add+ReLU) kernels (csrc/norm.hip): 3 launches each way instead of MIOpen's 3 + add + ReLU
        # (B=32 ResNet-18 step: 32.6 vs 33.0 ms).  DSF_FUSED_BN=0 keeps torch's BatchNorm2d / ReLU modules.
        self.fused_bn = bool(nn_conv.LAYERS["fused_bn"]) and os.environ.get("DSF_FUSED_BN", "1") == "1"
        self.Conv2d = nn_conv.LAYERS["Conv2d"]
        self.ConvTranspose2d = nn_conv.LAYERS["ConvTranspose2d"]
        self.MaxPool2d = nn_conv.LAYERS["MaxPool2d"]

    def bn_relu(self, c, **kw):
        """[BatchNorm, ReLU] pair of an nn.Sequential: fused into one module (the ReLU slot becomes an
        Identity so indices and state-dict keys do not move)."""
        if self.fused_bn:
            return [FusedBatchNorm2d(c, fuse_relu=True, **kw), nn.Identity()]
        return [nn.BatchNorm2d(c, **kw), nn.ReLU(inplace=True)]

    def bn(self, c, **kw):
        return FusedBatchNorm2d(c, **kw) if self.fused_bn else nn.BatchNorm2d(c, **kw)

    def __enter__(self):
        self._saved = (_resnet._CONV[0], _resnet._FUSED_BN[0])
        _resnet._CONV[0] = self.Conv2d
        _resnet._FUSED_BN[0] = self.fused_bn
        return self

    def __exit__(self, *a):
        _resnet._CONV[0], _resnet._FUSED_BN[0] = self._saved


def _head(seq, x):
    """``mano_regress`` = Sequential(AdaptiveAvgPool2d(1), Flatten, Linear): pooling + product as one launch on the GPU (ops.PoolLinear:
    torch runs a mean reduction, a hipBLASLt GEMM and five backward launches for a 32 x 512 x 62 product), the modules themselves elsewhere"""
    out = _ops.pool_linear(x, seq[2]) if (len(seq) == 3 and isinstance(seq[2], nn.Linear) and isinstance(seq[0], nn.AdaptiveAvgPool2d)) else None
    return out if out is not None else seq(x)


def _stem(seq, x):
    """``pre`` = Sequential(Conv2d, BatchNorm, ReLU slot, MaxPool2d) (reference model/backbone.py:200-204): in training the BatchNorm's
    apply pass writes the POOLED map only and the pooling's backward runs inside the BatchNorm backward
    (FusedBatchNorm2d.forward_pooled; the 134 MB full-resolution output and its gradient never exist); the modules themselves elsewhere"""
    if len(seq) == 4 and isinstance(seq[1], FusedBatchNorm2d) and seq[1].fuse_relu and isinstance(seq[2], nn.Identity) \
            and isinstance(seq[3], nn.MaxPool2d) and seq[1].training:
        mp = seq[3]
        geom = [v if isinstance(v, int) else (v[0] if v[0] == v[1] else None)
                for v in (mp.kernel_size, mp.stride if mp.stride is not None else mp.kernel_size, mp.padding, mp.dilation)]
        if None not in geom and geom[3] == 1 and not mp.ceil_mode and not mp.return_indices:
            return conv_bn_act(seq[0], seq[1], x, pool=(geom[0], geom[1], geom[2], mp))
    return seq(x)


def convtranspose_bn_relu(cin, cout, kernel, L):
    seq = ConvBN if L.fused_bn else nn.Sequential
    return seq(L.ConvTranspose2d(cin, cout, kernel, stride=2, padding=1, output_padding=0, bias=False), *L.bn_relu(cout, momentum=0.1))


class _TwoBranchNet(nn.Module):
    """Trunk builder shared by the 1-stage and 2-stage nets; ``suffix`` = '' or '_s2'."""

    def _stem(self):
        L = self._L
        self.pre = nn.Sequential(L.Conv2d(1, 64, kernel_size=5, stride=1, padding=2, bias=False),
                                 *L.bn_relu(64, momentum=BN_MOMENTUM),
                                 L.MaxPool2d(kernel_size=3, stride=2, padding=1))

    def _make_layer(self, block, planes, blocks, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = (ConvBN if self._L.fused_bn else nn.Sequential)(
                self._L.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                self._L.bn(planes * block.expansion, momentum=BN_MOMENTUM))
        layers = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def _trunk(self, block, layers, suffix):
        for i, (planes, stride) in enumerate(zip((64, 128, 256, 512), (1, 2, 2, 2))):
            setattr(self, 'layer%d%s' % (i + 1, suffix), self._make_layer(block, planes, layers[i], stride))
        setattr(self, 'mano_regress' + suffix,
                nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(self.inplanes, N_MANO)))
        L = self._L
        setattr(self, 'deconv_layer4' + suffix, convtranspose_bn_relu(self.inplanes, 256, 4, L))
        setattr(self, 'deconv_layer3' + suffix, convtranspose_bn_relu(256, 256, 4, L))
        setattr(self, 'deconv_layer2' + suffix, convtranspose_bn_relu(256, 256, 4, L))
        heads = nn.ModuleList([L.Conv2d(256, self.joint_num * 3, kernel_size=1, stride=1),
                               L.Conv2d(256, self.joint_num, kernel_size=1, stride=1)])
        setattr(self, 'finals' + suffix, heads)

    def _run_trunk(self, x, suffix):
        g = lambda n: getattr(self, n + suffix)
        c4, c4b = take_twin(g('layer4')(g('layer3')(g('layer2')(g('layer1')(x)))))    # c4 is read twice: a handle each (nn_norm.take_twin)
        heads = g('finals')
        if _HEAD_FORK[0] and c4.is_cuda:
            # the pooled MANO head is ONE workgroup per sample (65 us at B = 32 with an eighth of the chip busy, 2 x 29 us backward):
            # beside the decoder on the branch stream instead of in front of it; the decoder's launches are issued first
            f = fork(c4.device, params=self).mark()
            feat = g('deconv_layer2')(g('deconv_layer3')(g('deconv_layer4')(c4)))
            pix = nn_conv.fused_heads(feat, heads)
            with f.branch(0, c4b):
                mano = _head(g('mano_regress'), c4b)
            f.join()
            return c4, feat, pix, mano
        mano = _head(g('mano_regress'), c4b)
        feat = g('deconv_layer2')(g('deconv_layer3')(g('deconv_layer4')(c4)))
        pix = nn_conv.fused_heads(feat, heads)
        return c4, feat, pix, mano

    @torch.no_grad()
    def init_weights(self):
        # in-place ops on the parameters themselves (not `.data`): torch's version counters see them; the write epoch is
        # bumped as well so that nothing keyed on parameter contents (conv_x6 weight images) can outlive a re-initialisation
        nn_conv.weights_changed()
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.normal_(0, math.sqrt(2. / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.fill_(1)
                m.bias.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.normal_(0, 0.001)
            elif isinstance(m, nn.ConvTranspose2d):
                nn.init.normal_(m.weight, std=0.001)
        for name in ('finals', 'finals_s2'):
            for m in getattr(self, name, nn.ModuleList()).modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.normal_(m.weight, std=0.001)
                    nn.init.constant_(m.bias, 0)


class MANO_OCR(_TwoBranchNet):
    def __init__(self, backbone, joint_num):
        super().__init__()
        self._L = _Layers()
        self.joint_num = joint_num
        self.feature_dim = [joint_num * 3, joint_num]
        block, layers = resnet[int(backbone.split('_')[-1])]
        with self._L:
            self._stem()
            self.inplanes = 64
            self._trunk(block, layers, '')
        self.init_weights()

    def forward(self, img):
        _, _, pix, mano = self._run_trunk(_stem(self.pre, img), '')
        return [[pix, mano]]


class MANO_OCR_stage(_TwoBranchNet):
    def __init__(self, backbone, joint_num, refine=False, coord='xyz'):
        super().__init__()
        self._L = _Layers()
        self.joint_num = joint_num
        self.feature_dim = [joint_num * 3, joint_num]
        self.refine = refine
        self.coord_type = coord
        self.pool = nn.AdaptiveAvgPool2d(1)
        block, layers = resnet[int(backbone.split('_')[-1])]
        with self._L:
            self._stem()
            self.inplanes = 64
            self._trunk(block, layers, '')
            if refine:
                self.fusion = nn.Sequential(self._L.Conv2d(256 + joint_num * 4 * 2 + 64, 256, 3, 1, 1), *self._L.bn_relu(256))
                self.inplanes = 256
                self._trunk(block, layers, '_s2')
        self.init_weights()

    def forward(self, img, render=None, center=None, cube=None, M=None):
        c0 = _stem(self.pre, img)
        if not self.refine:
            _, feat, pix, mano = self._run_trunk(c0, '')
            return [[pix, mano]]
        if _BRIDGE_FORK[0]:
            # the stage-2 bridge (MANO head -> MANO layer -> rasteriser -> offset map: a chain of short, latency-bound launches,
            # forward and backward) needs only c4; the decoder (three transposed convolutions + heads: large launches) needs
            # nothing of it until the stage-2 `cat` -- beside each other on forked streams (streams.py)
            c4, c4b = take_twin(self.layer4(self.layer3(self.layer2(self.layer1(c0)))))
            f = fork(c4.device, params=self)
            first = _BRIDGE_ORDER[0] == "decoder"           # the decoder's launches are ISSUED first; both chains start at the fork point
            if first:
                f.mark()
                feat = self.deconv_layer2(self.deconv_layer3(self.deconv_layer4(c4)))
                pix = nn_conv.fused_heads(feat, self.finals)
            with f.branch(0, c4b):
                mano = _head(self.mano_regress, c4b)
                mano_img, mano_uvd, _, _ = render.render(mano, center, cube)
                remap = joint2offset(mano_uvd, mano_img, 0.8, 64)
            if not first:
                feat = self.deconv_layer2(self.deconv_layer3(self.deconv_layer4(c4)))
                pix = nn_conv.fused_heads(feat, self.finals)
            f.join()
        else:
            _, feat, pix, mano = self._run_trunk(c0, '')
            # stage-2 bridge: render the stage-1 MANO estimate, re-encode it as an offset map (HIP kernels)
            mano_img, mano_uvd, _, _ = render.render(mano, center, cube)
            remap = joint2offset(mano_uvd, mano_img, 0.8, 64)
        _, _, pix2, mano2 = self._run_trunk(self.fusion(_ops.cat_channels((c0, feat, pix, remap))), '_s2')
        return [[pix, mano], [pix2, mano2]]

    def encoder(self, img):
        c0 = _stem(self.pre, img)
        c4, _, pix, _ = self._run_trunk(c0, '')
        return self.pool(c4).squeeze(), offset2joint_softmax(pix, img, 0.8)
