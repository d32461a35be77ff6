"""Counterpart of the reference's ``render_model/render_loss.py`` (``depth_loss``).
``surface_loss`` (pytorch3d chamfer) is never called by the trainer and is out of scope."""
import torch


class depth_loss(torch.nn.Module):
    """Mean |real - synth| over pixels that are foreground (< 0.99) in BOTH images
    (/root/reference/render_model/render_loss.py:9-21; its random sub-mask is always true)."""

    def __init__(self, beta=0.4, smooth=False):
        super().__init__()
        self.smooth = smooth
        self.smoothLoss = torch.nn.SmoothL1Loss(beta=beta)

    def forward(self, real, synth):
        both = (real.lt(0.99) & synth.lt(0.99)).to(real.dtype)
        n = both.sum()
        if not self.smooth:
            return ((real - synth).abs() * both).sum() / n
        d = (synth - real) * both
        beta = self.smoothLoss.beta
        per = torch.where(d.abs() < beta, 0.5 * d * d / beta, d.abs() - 0.5 * beta)
        return per.sum() / n


def m2d_loss(real_crop, synth_crop):
    """Model-to-data depth term written inline in the trainer (train_render.py:728-732): L1 over the
    UNION of the two foreground masks, normalised per sample, batch mean, x0.1."""
    if real_crop.is_cuda:
        from .. import ops
        fused = ops.m2d(real_crop, synth_crop)               # one reduction launch pair + one backward launch (csrc/step_ops.hip)
        if fused is not None:
            return fused[0]
    union = (real_crop.lt(0.99) | synth_crop.lt(0.99)).to(real_crop.dtype)
    per = ((real_crop - synth_crop).abs() * union).sum(-1).sum(-1) / (union.sum(-1).sum(-1) + 1e-8)
    return per.mean() * 0.1
