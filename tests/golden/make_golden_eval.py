#!/usr/bin/env python
"""Golden vectors for the evaluation metric of the reference (``Trainer.xyz2error``, train_render.py:826-864), made by
IMPORTING the reference's trainer module (third-party modules the container lacks are empty import-time stand-ins, as in
make_golden.py; no arithmetic of the reference is stubbed).  Writes tests/golden/reference_eval.npz (arrays only).

    python tests/golden/make_golden_eval.py        # build container only: /root/reference is not on the GPU box
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402


def main():
    mg._install_stubs()
    pmd = types.ModuleType("pytorch3d.loss.point_mesh_distance")
    sys.modules["pytorch3d.loss"].point_mesh_distance = pmd
    sys.modules["pytorch3d.loss.point_mesh_distance"] = pmd
    sys.path.insert(0, mg.REF)
    cwd = os.getcwd()
    os.chdir("/tmp")                                   # the trainer module creates log directories relative to cwd
    import train_render as tr
    os.chdir(cwd)
    rng = np.random.default_rng(77)
    out = {}
    for ds, J in (("nyu", 14), ("msra", 21), ("icvl", 16)):
        B = 5
        pred = rng.uniform(-1, 1, (B, J, 3)).astype(np.float32)
        gt = (pred + rng.normal(size=(B, J, 3)).astype(np.float32) * 0.05).astype(np.float32)
        center = np.stack([rng.uniform(-40, 40, B), rng.uniform(-40, 40, B), rng.uniform(500, 1200, B)], 1).astype(np.float32)
        cube = np.full((B, 3), 250.0, dtype=np.float32)
        cube[1] = (200.0, 300.0, 250.0)
        fake = types.SimpleNamespace(config=types.SimpleNamespace(dataset=ds), phase="train")
        T = torch.tensor
        out[ds + "_pred"], out[ds + "_gt"], out[ds + "_center"], out[ds + "_cube"] = pred, gt, center, cube
        out[ds + "_err"] = np.float64(tr.Trainer.xyz2error(fake, T(pred), T(gt), T(center), T(cube)))
        out[ds + "_err_batch"] = tr.Trainer.xyz2error(fake, T(pred), T(gt), T(center), T(cube), keep_batch=True)
        out[ds + "_err_joint"] = tr.Trainer.xyz2error(fake, T(pred), T(gt), T(center), T(cube), keep_joint=True)
    # ---- Render.mask_img (mano_layer.py:1326-1340) with its random draws made explicit: the draws are reproduced by
    #      replaying the same RNG calls from the same seeds, then the reference runs from those seeds
    from render_model import mano_layer as ml
    R = ml.Render.__new__(ml.Render)
    torch.nn.Module.__init__(R)
    xx, yy = np.meshgrid(np.arange(128), np.arange(128))
    R.xy_mesh = torch.from_numpy(np.stack((2 * (xx + 0.5) / 128 - 1.0, 2 * (yy + 0.5) / 128 - 1.0), axis=-1).reshape([1, -1, 2])).float()
    B, J = 3, 21
    img = rng.uniform(-1, 1, (B, 1, 128, 128)).astype(np.float32)
    img[img > 0.2] = 1.0
    juvd = rng.uniform(-0.7, 0.7, (B, J, 3)).astype(np.float32)
    np.random.seed(123); torch.manual_seed(456)
    k = np.random.choice(np.arange(3, 10), 1, replace=False)[0]
    joint_id = np.random.choice(np.arange(0, J), k, replace=False)
    uvd_offset = (torch.rand(B, k, 3) - 0.5) * 0.15 * 2
    mask_range = torch.rand([B, k]) * 0.3
    np.random.seed(123); torch.manual_seed(456)
    masked = R.mask_img(torch.tensor(img), torch.tensor(juvd), 0.15, 0.3)
    out["mask_img"], out["mask_juvd"], out["mask_joint_id"] = img, juvd, joint_id.astype(np.int64)
    out["mask_offset"], out["mask_radius"], out["mask_out"] = uvd_offset.numpy(), mask_range.numpy(), masked.numpy()
    assert (masked.numpy() != img).any()
    # ---- discriminators and GAN losses of render_model/transfer.py ----
    from render_model import transfer as rt
    xg = torch.tensor(rng.uniform(-1, 1, (2, 1, 128, 128)).astype(np.float32))
    for tag, args in (("basic", (1, 64, "basic", 3, "instance", "normal", 0.02)), ("pixel", (1, 64, "pixel", 3, "batch", "xavier", 0.02)),
                      ("nl2", (1, 32, "n_layers", 2, "batch", "normal", 0.02))):
        torch.manual_seed(11)
        D = rt.define_D(*args)
        D.eval()
        with torch.no_grad():
            o = D(xg)
        out["D_%s_keys" % tag] = np.array(list(D.state_dict().keys()))
        out["D_%s_out" % tag] = o.numpy()
    out["D_x"] = xg.numpy()
    pred = torch.tensor(rng.normal(size=(2, 1, 14, 14)).astype(np.float32))
    out["gan_pred"] = pred.numpy()
    for mode in ("lsgan", "vanilla", "wgangp"):
        L = rt.GANLoss(mode)
        out["gan_%s_real" % mode] = np.float64(L(pred, True))
        out["gan_%s_fake" % mode] = np.float64(L(pred, False))
    np.savez_compressed(os.path.join(HERE, "reference_eval.npz"), **out)
    print("reference_eval.npz", os.path.getsize(os.path.join(HERE, "reference_eval.npz")), "bytes")


if __name__ == "__main__":
    main()
