#!/usr/bin/env python
"""Golden vectors for the ResNet-50 / ``Bottleneck`` backbone (model/resnet.py:58-98, model/backbone.py:188-343) made by
IMPORTING the reference: state-dict keys, eval-mode outputs of the one-stage net, and a TRAINING-mode forward + backward of
the same net (batch statistics, gradients of a fixed functional w.r.t. a handful of named parameters and the input).
Writes tests/golden/reference_r50.npz (arrays only).

    python tests/golden/make_golden_r50.py        # build container only: /root/reference is not on the GPU box
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402

PROBES = ["pre.0.weight", "layer1.0.conv1.weight", "layer1.0.conv3.weight", "layer1.0.downsample.0.weight",
          "layer1.0.bn3.weight", "layer2.0.conv2.weight", "layer3.5.conv2.weight", "layer4.2.conv3.weight",
          "layer4.0.downsample.1.bias", "mano_regress.2.weight", "deconv_layer4.0.weight", "finals.0.weight"]


def main():
    mg._install_stubs()
    sys.path.insert(0, mg.REF)
    from model.backbone import MANO_OCR_stage
    out = {}
    torch.manual_seed(7)
    net = MANO_OCR_stage("ResNet_stage_50", 21, False)
    out["keys"] = np.array(list(net.state_dict().keys()))
    out["nparams"] = np.array([sum(p.numel() for p in net.parameters())])
    rng = np.random.default_rng(15)
    x = torch.tensor(rng.uniform(-1, 1, (2, 1, 128, 128)).astype(np.float32))
    out["x"] = x.numpy()
    net.eval()
    with torch.no_grad():
        (pix, par), = net(x)
    out["eval_pix_sub"], out["eval_par"] = pix.numpy()[:, :, ::8, ::8].copy(), par.numpy()
    # training mode: batch statistics + backward of  sum(pix * gw_pix) + sum(par * gw_par)
    net.train()
    xg = x.clone().requires_grad_(True)
    (pix, par), = net(xg)
    gw_pix = torch.tensor(rng.normal(size=tuple(pix.shape)).astype(np.float32))
    gw_par = torch.tensor(rng.normal(size=tuple(par.shape)).astype(np.float32))
    ((pix * gw_pix).sum() + (par * gw_par).sum()).backward()
    out["train_pix_sub"], out["train_par"] = pix.detach().numpy()[:, :, ::8, ::8].copy(), par.detach().numpy()
    # (gw_pix / gw_par are not stored: the tests redraw them from default_rng(15) after x, in this order)
    out["gw_pix_checksum"] = np.array([float(gw_pix.double().sum()), float(gw_par.double().sum())])
    out["grad_x_sub"] = xg.grad.numpy()[:, :, ::4, ::4].copy()
    named = dict(net.named_parameters())
    out["probe_names"] = np.array(PROBES)
    for i, n in enumerate(PROBES):
        g = named[n].grad.numpy()
        out["probe%d_norm" % i] = np.array([np.sqrt((g.astype(np.float64) ** 2).sum())])
        out["probe%d_head" % i] = g.reshape(-1)[:64].copy()
    out["running_mean_layer4_2_bn3_head"] = net.layer4[2].bn3.running_mean.numpy()[:32].copy()
    np.savez_compressed(os.path.join(HERE, "reference_r50.npz"), **out)
    print("reference_r50.npz", os.path.getsize(os.path.join(HERE, "reference_r50.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
