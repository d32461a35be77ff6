"""Stacked-hourglass pixel network (counterpart of the reference's ``model/hourglass.py:62-259``) on the HIP
convolutions, the fused BatchNorm(+ReLU) kernels (csrc/norm.hip) and the HIP max pooling of this package.
``PoseNetMANO`` adds the 62-d MANO head that BASELINE config 3 needs (the reference's PoseNet has none and is not used
by its trainer).  Module / parameter names equal the reference's."""
import math

import torch
from torch import nn

from .. import nn_conv
from ..nn_norm import FusedBatchNorm2d, conv_bn_act
from ..streams import fork


def _bn_relu(c, relu=True):
    """(BatchNorm2d, ReLU) pair: the fused kernel (ReLU inside, an Identity keeps the attribute) unless the layer registry
    asks for plain torch modules (oracle.nets twins)"""
    if nn_conv.LAYERS["fused_bn"]:
        return FusedBatchNorm2d(c, fuse_relu=relu), (nn.Identity() if relu else None)
    return nn.BatchNorm2d(c), (nn.ReLU() if relu else None)



class Conv(nn.Module):
    def __init__(self, inp_dim, out_dim, kernel_size=3, stride=1, bn=False, relu=True):
        super().__init__()
        self.inp_dim = inp_dim
        self.conv = nn_conv.LAYERS["Conv2d"](inp_dim, out_dim, kernel_size, stride, padding=(kernel_size - 1) // 2, bias=True)
        if bn:
            self.bn, self.relu = _bn_relu(out_dim, relu)
        else:
            self.bn, self.relu = None, (nn.ReLU() if relu else None)

    def forward(self, x):
        if isinstance(self.bn, FusedBatchNorm2d):            # (the 1-channel stem: one autograd node, see nn_norm._StemFunction)
            return conv_bn_act(self.conv, self.bn, x)
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return self.relu(x) if self.relu is not None else x


class Residual(nn.Module):
    """pre-activation bottleneck: BN-ReLU-1x1, BN-ReLU-3x3, BN-ReLU-1x1 (+1x1 skip when widths differ)."""

    def __init__(self, inp_dim, out_dim):
        super().__init__()
        mid = int(out_dim / 2)
        self.bn1, self.relu1 = _bn_relu(inp_dim)
        self.conv1 = Conv(inp_dim, mid, 1, relu=False)
        self.bn2, self.relu2 = _bn_relu(mid)
        self.conv2 = Conv(mid, mid, 3, relu=False)
        self.bn3, self.relu3 = _bn_relu(mid)
        self.conv3 = Conv(mid, out_dim, 1, relu=False)
        self.skip_layer = Conv(inp_dim, out_dim, 1, relu=False)
        self.need_skip = inp_dim != out_dim

    def forward(self, x):
        y = self.conv1(self.relu1(self.bn1(x)))
        y = self.conv2(self.relu2(self.bn2(y)))
        y = self.conv3(self.relu3(self.bn3(y)))
        return y + (self.skip_layer(x) if self.need_skip else x)


class Hourglass(nn.Module):
    def __init__(self, n, f, bn=None, increase=0):
        super().__init__()
        nf = f + increase
        self.up1 = Residual(f, f)
        self.pool1 = nn_conv.LAYERS["MaxPool2d"](2, 2)
        self.low1 = Residual(f, nf)
        self.n = n
        self.low2 = Hourglass(n - 1, nf, bn=bn) if n > 1 else Residual(nf, nf)
        self.low3 = Residual(nf, f)
        self.up2 = nn.Upsample(scale_factor=2, mode='nearest')

    def forward(self, x):
        # the two arms are independent between x and the add: ``up1`` works on this level's own (large) maps, the other arm on
        # maps a quarter of that size and smaller (8 x 8 ... 2 x 2: 32 ... 2 tiles for 256 CUs) -- a stream per level for ``up1``, so
        # that the chain of small launches runs beside the large ones instead of between them (streams.py; config 3 +5 %)
        f = fork(x.device, params=self)
        with f.branch(self.n, x):
            up = self.up1(x)
        low = self.up2(self.low3(self.low2(self.low1(self.pool1(x)))))
        f.join()
        return up + low


class Merge(nn.Module):
    def __init__(self, x_dim, y_dim):
        super().__init__()
        self.conv = Conv(x_dim, y_dim, 1, relu=False, bn=False)

    def forward(self, x):
        return self.conv(x)


class PoseNet(nn.Module):
    def __init__(self, nstack, joint_num, inp_dim=256, bn=False, increase=0, **kwargs):
        super().__init__()
        self._build(nstack, joint_num, inp_dim, bn, increase)

    def _build(self, nstack, joint_num, inp_dim, bn, increase):
        self.nstack = nstack
        self.joint_num = joint_num
        self.pre = nn.Sequential(Conv(1, 64, 7, 2, bn=True, relu=True), Residual(64, 128), nn_conv.LAYERS["MaxPool2d"](2, 2),
                                 Residual(128, 256), Residual(256, inp_dim))
        self.hgs = nn.ModuleList([Hourglass(4, inp_dim, bn, increase) for _ in range(nstack)])
        self.features = nn.ModuleList([nn.Sequential(Residual(inp_dim, inp_dim), Conv(inp_dim, inp_dim, 1, bn=True, relu=True))
                                       for _ in range(nstack)])
        head = lambda c: nn.ModuleList([nn_conv.LAYERS["Conv2d"](inp_dim, c, kernel_size=1, stride=1, padding=0) for _ in range(nstack)])
        self.outs_1 = head(joint_num * 3)
        self.outs_2 = head(joint_num)
        self.outs_3 = head(joint_num)
        self.merge_features = nn.ModuleList([Merge(inp_dim, inp_dim) for _ in range(nstack - 1)])
        self.merge_preds = nn.ModuleList([Merge(joint_num * 5, inp_dim) for _ in range(nstack - 1)])
        self.merge_all = nn.ModuleList([Merge(inp_dim * 2, inp_dim) for _ in range(nstack - 1)])
        self.init_weights()

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
        for heads in (self.outs_1, self.outs_2):
            for m in heads:
                nn.init.normal_(m.weight, std=0.001)
                nn.init.constant_(m.bias, 0)

    def forward(self, imgs):
        x = self.pre(imgs)
        preds_all, hg = [], None
        for i in range(self.nstack):
            hg = self.hgs[i](x)
            feature = self.features[i](hg)
            preds = torch.cat((self.outs_1[i](feature), self.outs_2[i](feature), self.outs_3[i](feature)), dim=1)
            preds_all.append(preds)
            if i < self.nstack - 1:
                x = x + self.merge_preds[i](preds) + self.merge_features[i](feature)
        return preds_all, hg


class _GlobalAvgPoolFunction(torch.autograd.Function):
    """the two single-pass sums below with a CHANNELS_LAST gradient: autograd's own backward of ``sum`` expands the pooled
    gradient in torch's standard order, and that order then travels down every skip connection of the last hourglass (each
    add takes it over, each upsampling backward runs its strided path: 4 x 55 us at 2 ... 128 workgroups) until a HIP kernel
    converts it.  Same values."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return (x.sum(3).sum(2) * (1.0 / (x.shape[2] * x.shape[3]))).reshape(x.shape[0], x.shape[1], 1, 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        gx = torch.empty((B, C, H, W), device=g.device, dtype=g.dtype, memory_format=torch.channels_last)
        gx.copy_((g.reshape(B, C, 1, 1) * (1.0 / (H * W))).expand(B, C, H, W))
        return gx


class GlobalAvgPool2d(nn.Module):
    """nn.AdaptiveAvgPool2d(1) as two single-pass sums (over W, then over H).  torch's one-shot mean over a large H x W
    to B x C outputs is a multi-block reduction whose semaphores are zeroed by a hipMemsetAsync -- a node a captured HIP
    graph does not replay correctly on ROCm 7.2 (train_step.GraphedStep refuses such graphs)."""

    def forward(self, x):
        if x.is_cuda and x.dim() == 4:
            return _GlobalAvgPoolFunction.apply(x)
        return (x.sum(3).sum(2) * (1.0 / (x.shape[2] * x.shape[3]))).reshape(x.shape[0], x.shape[1], 1, 1)


class PoseNetMANO(nn.Module):
    """PoseNet + a pooled linear head regressing the 62 MANO parameters (BASELINE config 3)."""

    def __init__(self, nstack=2, joint_num=21):
        super().__init__()
        self.body = PoseNet(nstack, joint_num)
        self.mano_regress = nn.Sequential(GlobalAvgPool2d(), nn.Flatten(), nn.Linear(256, 62))
        nn.init.normal_(self.mano_regress[2].weight, std=0.001)
        nn.init.zeros_(self.mano_regress[2].bias)

    def forward(self, img):
        preds, hg = self.body(img)
        from .. import ops
        head = ops.pool_linear(hg, self.mano_regress[2])          # pooling + Linear in one launch (ops.PoolLinear)
        return preds, (head if head is not None else self.mano_regress(hg))
