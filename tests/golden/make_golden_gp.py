#!/usr/bin/env python
"""Golden vectors for the WGAN-GP gradient penalty of the reference (``cal_gradient_penalty``,
render_model/transfer.py:356-391), made by IMPORTING the reference's transfer module and calling its own function on its
own ``define_D`` discriminators.  The reference draws the mixing coefficients with ``torch.rand`` inside the call; the draw is
recovered by replaying the same seed (recorded as ``*_alpha``) so that the product, whose penalty takes ``alpha`` as an
explicit input, can be run on the same numbers.  Writes tests/golden/reference_gp.npz (arrays only).

    python tests/golden/make_golden_gp.py        # build container only: /root/reference is not on the GPU box
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402

CASES = (("basic", (1, 8, "basic", 3, "instance", "normal", 0.2), "mixed"),
         ("nl2", (1, 8, "n_layers", 2, "batch", "normal", 0.2), "mixed"),
         ("pixel", (1, 8, "pixel", 3, "batch", "normal", 0.2), "mixed"),
         ("basic_real", (1, 8, "basic", 3, "instance", "normal", 0.2), "real"),
         ("basic_fake", (1, 8, "basic", 3, "instance", "normal", 0.2), "fake"))
SEED_NET, SEED_DATA, SEED_ALPHA, B, S = 21, 22, 23, 3, 64


def main():
    mg._install_stubs()
    sys.path.insert(0, mg.REF)
    from render_model import transfer as rt
    out = {"seed_net": np.int64(SEED_NET), "B": np.int64(B), "S": np.int64(S)}
    g = torch.Generator().manual_seed(SEED_DATA)
    real = torch.rand(B, 1, S, S, generator=g) * 2 - 1
    fake = torch.rand(B, 1, S, S, generator=g) * 2 - 1
    out["real"], out["fake"] = real.numpy(), fake.numpy()
    for tag, args, kind in CASES:
        torch.manual_seed(SEED_NET)
        D = rt.define_D(*args)                                   # training mode, as the reference calls it
        out[tag + "_keys"] = np.array(list(D.state_dict().keys()))
        # checksums of the seeded state (the product's twin built from the same seed must reproduce them: same weights)
        out[tag + "_state_sums"] = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in D.state_dict().values()
                                             if v.dtype.is_floating_point])
        torch.manual_seed(SEED_ALPHA)
        alpha = torch.rand(B, 1)                                 # the draw the call below makes first (type 'mixed')
        torch.manual_seed(SEED_ALPHA)
        gp, grads = rt.cal_gradient_penalty(D, real.clone(), fake.clone(), "cpu", kind, 1.0, 10.0)
        gp.backward()
        out[tag + "_alpha"] = alpha.numpy()
        out[tag + "_gp"] = np.float64(gp.item())
        out[tag + "_grads"] = grads.detach().numpy()
        pg = np.concatenate([p.grad.flatten().numpy() for p in D.parameters() if p.grad is not None])
        assert np.isfinite(pg).all() and np.abs(pg).max() > 0
        if kind == "mixed":
            out[tag + "_param_grads"] = pg
    # lambda_gp = 0: the reference returns (0.0, None)
    z = rt.cal_gradient_penalty(D, real, fake, "cpu", "mixed", 1.0, 0.0)
    assert z == (0.0, None)
    np.savez_compressed(os.path.join(HERE, "reference_gp.npz"), **out)
    print("reference_gp.npz", os.path.getsize(os.path.join(HERE, "reference_gp.npz")), "bytes")


if __name__ == "__main__":
    main()
