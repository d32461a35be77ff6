"""Residual blocks of the backbone (counterpart of the reference's ``model/resnet.py:18-98``) on the HIP convolution
and fused BN(+add+ReLU) kernels of this package.  Attribute names follow the reference so its checkpoints (state-dict
keys) load unchanged."""
import torch.nn as nn

from ..nn_conv import Conv2d as _HipConv2d
from ..nn_norm import FusedBatchNorm2d, conv_bn_act, take_twin
from ..streams import fork
import os
_DS_FORK = [os.environ.get("DSF_DS_FORK", "1") == "1"]

_CONV = [_HipConv2d]      # layer factory of the network under construction (set by model/backbone.py::_Layers)
_FUSED_BN = [False]       # fused BN+add+ReLU kernels (see model/backbone.py::_Layers)


def _norm(c, **kw):
    return FusedBatchNorm2d(c, **kw) if _FUSED_BN[0] else nn.BatchNorm2d(c, **kw)


def _conv(cin, cout, k, stride=1):
    return _CONV[0](cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 3, stride)
        self.bn1 = _norm(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv(planes, planes, 3)
        self.bn2 = _norm(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # x is read twice (conv1, identity / downsample): one handle of the producing BatchNorm's twin output each, so that its
        # backward gets the two gradients separately and adds them in its own sums pass (nn_norm.take_twin); this block's output is
        # handed out the same way
        x, x2 = take_twin(x)
        if self.downsample is not None and _DS_FORK[0]:
            f = fork(x.device, params=self)
            with f.branch(0, x2):
                identity = self.downsample(x2)
            y = conv_bn_act(self.conv1, self.bn1, x, relu=True)
            f.join()
            return conv_bn_act(self.conv2, self.bn2, y, residual=identity, relu=True, twin=True)
        y = conv_bn_act(self.conv1, self.bn1, x, relu=True)                       # (BN statistics from the conv epilogue)
        identity = x2 if self.downsample is None else self.downsample(x2)
        return conv_bn_act(self.conv2, self.bn2, y, residual=identity, relu=True, twin=True)    # bn + skip + relu in one pass


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = _norm(planes)
        self.conv2 = _conv(planes, planes, 3, stride)
        self.bn2 = _norm(planes)
        self.conv3 = _conv(planes, planes * self.expansion, 1)
        self.bn3 = _norm(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        x, x2 = take_twin(x)                                 # (see BasicBlock.forward)
        if self.downsample is not None and _DS_FORK[0]:
            f = fork(x.device, params=self)
            with f.branch(0, x2):
                identity = self.downsample(x2)
            y = conv_bn_act(self.conv1, self.bn1, x, relu=True)
            y = conv_bn_act(self.conv2, self.bn2, y, relu=True)
            f.join()
            return conv_bn_act(self.conv3, self.bn3, y, residual=identity, relu=True, twin=True)
        y = conv_bn_act(self.conv1, self.bn1, x, relu=True)
        y = conv_bn_act(self.conv2, self.bn2, y, relu=True)
        identity = x2 if self.downsample is None else self.downsample(x2)
        return conv_bn_act(self.conv3, self.bn3, y, residual=identity, relu=True, twin=True)
