"""nn.MaxPool2d on the NHWC HIP kernels of dsf_amd/csrc/pool.hip (the stem pooling of the reference's backbone,
model/backbone.py:200-204, and the 2x2 poolings of model/hourglass.py:131).  Drop-in subclass: same constructor, no
parameters; channels_last fp32 GPU tensors with C % 4 == 0 take the HIP path (1-byte argmax, gather backward), anything
else raises -- there is no CPU path."""
import ctypes

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L
from ._lib import I, check, stream_ptr

CL = torch.channels_last


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad):
        x = x.contiguous(memory_format=CL)
        B, C, Hi, Wi = x.shape
        Ho, Wo = (Hi + 2 * pad - k) // stride + 1, (Wi + 2 * pad - k) // stride + 1
        y = torch.empty((B, C, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=CL)
        need = ctx.needs_input_grad[0]
        arg = torch.empty((B, Ho, Wo, C), device=x.device, dtype=torch.uint8) if need else None
        if B:
            check(L.lib().dsf_maxpool_forward(_p(x), _p(y), _p(arg) if need else ctypes.c_void_p(0), I(B), I(Hi), I(Wi), I(C),
                                              I(Ho), I(Wo), I(k), I(stride), I(pad), stream_ptr()), "dsf_maxpool_forward")
        if need:
            ctx.save_for_backward(arg)
        ctx.geom = (B, Hi, Wi, C, Ho, Wo, k, stride, pad)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        arg, = ctx.saved_tensors
        B, Hi, Wi, C, Ho, Wo, k, stride, pad = ctx.geom
        gy = gy.contiguous(memory_format=CL)
        gx = torch.empty((B, C, Hi, Wi), device=gy.device, dtype=torch.float32, memory_format=CL)
        if B:
            check(L.lib().dsf_maxpool_backward(_p(gy), _p(arg), _p(gx), I(B), I(Hi), I(Wi), I(C), I(Ho), I(Wo), I(k), I(stride),
                                               I(pad), stream_ptr()), "dsf_maxpool_backward")
        return gx, None, None, None


class MaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        one = lambda v: v if isinstance(v, int) else (v[0] if v[0] == v[1] else None)
        k, s, p, d = one(self.kernel_size), one(self.stride if self.stride is not None else self.kernel_size), one(self.padding), \
            one(self.dilation)
        if not x.is_cuda:
            raise RuntimeError("dsf_amd MaxPool2d runs on the GPU only (got a %s tensor); there is no CPU path" % x.device)
        if None in (k, s, p) or d != 1 or self.ceil_mode or self.return_indices or x.dtype != torch.float32 or x.dim() != 4 \
                or x.shape[1] % 4 or k > 15 or 2 * p > k:
            return super().forward(x)                      # geometries the reference never uses: torch's kernel
        return _MaxPool.apply(x, k, s, p)
