"""Residual blocks of the backbone (counterpart of the reference's ``model/resnet.py:18-98``).
Dense convolutions stay on PyTorch-ROCm (MIOpen / hipBLASLt MFMA kernels): the north star keeps
MFMA for the backbone's conv GEMMs only.  Attribute names follow the reference so its
checkpoints (state-dict keys) load unchanged."""
import torch.nn as nn

from ..nn_conv import Conv2d as _HipConv2d

_CONV = [_HipConv2d]      # layer factory switch: [nn.Conv2d] builds the plain-torch CPU twin (oracle / tests)


def _conv(cin, cout, k, stride=1):
    return _CONV[0](cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 3, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv(planes, planes, 3)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        y += x if self.downsample is None else self.downsample(x)
        return self.relu(y)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * self.expansion, 1)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        y += x if self.downsample is None else self.downsample(x)
        return self.relu(y)
