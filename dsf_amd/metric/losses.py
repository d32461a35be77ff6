"""Counterpart of the reference's ``metric/losses.py`` (Huber loss with delta 0.01)."""
import torch


class SmoothL1Loss(torch.nn.Module):
    """0.5 z^2 for |z| < 0.01 else 0.01 (|z| - 0.005); mean over the last dim, then mean (or sum)
    over the rest (/root/reference/metric/losses.py:6-30)."""

    def __init__(self, size_average=True):
        super().__init__()
        self.size_average = size_average

    def forward(self, x, y, weight=None):
        """``weight``: optional loss weight folded into the fused kernel's scale factor (the trainer multiplies every
        term by a constant, train_render.py:444-466; folding it saves a scalar multiply kernel each way)."""
        assert x.shape == y.shape
        if not x.is_cuda:
            raise RuntimeError("dsf_amd SmoothL1Loss runs on the GPU only (got a %s tensor); there is no CPU path" % x.device)
        from .. import ops
        fused = ops.huber_mean(x, y, 0.01, self.size_average, 1.0 if weight is None else float(weight))
        if fused is not None:
            return fused
        # target needs a gradient (or non-fp32 inputs): the reference's own formula, on the device
        z = (x - y).float()
        a = z.abs()
        per = torch.where(a < 0.01, 0.5 * z * z, 0.01 * (a - 0.005)).mean(dim=-1)
        out = per.mean() if self.size_average else per.sum()
        return out if weight is None else out * weight


class WeightSmoothL1Loss(torch.nn.Module):
    """Same with a per-row weight applied to z before the two branches (reference :32-57)."""

    def __init__(self, size_average=True):
        super().__init__()
        self.size_average = size_average

    def forward(self, x, y, weight):
        assert x.shape == y.shape
        z = (x - y).float()
        small = z.abs() < 0.01
        zw = z * weight.unsqueeze(-1)
        per = torch.where(small, 0.5 * zw * zw, 0.01 * (zw.abs() - 0.005)).mean(dim=-1)
        return per.mean() if self.size_average else per.sum()
