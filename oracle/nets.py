"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by dsf_amd/).

Plain ``torch.nn`` twins of the product networks for the CPU side of the tests and for bench.py's ``cpu_baseline`` leg:
inside ``torch_twin()`` the layer registry the model builders construct from (``dsf_amd.nn_conv.LAYERS``) holds
``torch.nn.Conv2d`` / ``torch.nn.ConvTranspose2d`` and plain ``BatchNorm2d`` + ``ReLU`` modules, so the same builder code
yields a network with identical parameters and state-dict keys that runs on CPU tensors through torch's own kernels.
The product never builds these: its registry always holds the HIP layers."""
import contextlib

import torch.nn as nn


@contextlib.contextmanager
def torch_twin():
    from dsf_amd import nn_conv
    saved = dict(nn_conv.LAYERS)
    nn_conv.LAYERS.update(Conv2d=nn.Conv2d, ConvTranspose2d=nn.ConvTranspose2d, fused_bn=False, MaxPool2d=nn.MaxPool2d)
    try:
        yield
    finally:
        nn_conv.LAYERS.update(saved)


def build(fn, *args, **kwargs):
    """``build(MANO_OCR_stage, 'ResNet_stage_18', 21, True)`` -> the torch.nn twin of that product network"""
    with torch_twin():
        return fn(*args, **kwargs)
