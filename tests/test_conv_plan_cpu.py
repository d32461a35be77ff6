"""Host-side dispatch of the split-operand convolutions (no GPU: dsf_conv_x6_forward_plan only reads its arguments and the
environment).  Which layers take the patch-staged kernel (variant 2), which keep the per-tap gather (1: weight fragments direct,
0: both operands through LDS), and how many K splits the launcher picks -- the rules DESIGN.md section 4 (K11p) states."""
import ctypes

import pytest

from dsf_amd import _lib as L

I = ctypes.c_int


def plan(B, Hi, Wi, Ci, Ho, Wo, Co, K, stride=1, dil=1, pad=1):
    v, k = ctypes.c_int(-9), ctypes.c_int(-9)
    rc = L.lib().dsf_conv_x6_forward_plan(I(B), I(Hi), I(Wi), I(Ci), I(Ho), I(Wo), I(Co), I(K), I(K), I(stride), I(dil), I(pad), I(pad),
                                          ctypes.byref(v), ctypes.byref(k))
    assert rc == 0
    return v.value, k.value


@pytest.fixture(autouse=True)
def _default_mode(monkeypatch):
    for name in ("DSF_X6_PATCH", "DSF_X6_BDIRECT", "DSF_X6_BM"):
        monkeypatch.delenv(name, raising=False)
    was = bool(L.lib().dsf_set_deterministic(I(0)))
    yield
    L.lib().dsf_set_deterministic(I(1 if was else 0))


@pytest.mark.parametrize("shape,variant,splits", [
    # 3 x 3, stride 1, pad 1 at B = 32: every map width of the ResNets takes the patch kernel, unsplit from 256 tiles
    ((32, 64, 64, 488, 64, 64, 256, 3), 2, 1),
    ((32, 64, 64, 64, 64, 64, 64, 3), 2, 1),
    ((32, 32, 32, 128, 32, 32, 128, 3), 2, 1),
    ((32, 16, 16, 256, 16, 16, 256, 3), 2, 1),              # 256 tiles of 64 rows: unsplit
    ((32, 8, 8, 512, 8, 8, 512, 3), 2, 4),                  # 128 tiles: four splits of eight channel chunks
    ((192, 8, 8, 512, 8, 8, 512, 3), 2, 1),                 # 8-wide maps keep 64-row tiles (one image each) at any batch
    # other geometries stay on the gather kernels
    ((32, 4, 4, 128, 4, 4, 128, 3), 1, None),               # 4-wide maps
    ((32, 64, 64, 64, 32, 32, 128, 3, 2), 1, 1),            # stride 2
    ((32, 64, 64, 256, 64, 64, 84, 1, 1, 1, 0), 1, 1),      # 1 x 1
    ((32, 16, 16, 256, 8, 8, 512, 3, 2), 0, None),          # 64-row tiles with several n tiles: weights through LDS
])
def test_which_kernel_a_layer_takes(shape, variant, splits):
    v, k = plan(*shape)
    assert v == variant, (shape, v, k)
    if splits is not None:
        assert k == splits, (shape, v, k)


def test_transposed_4x4_stride_2_and_reflection_padded_layers_take_the_patch_kernel():
    # ConvTranspose2d(k 4, s 2, p 1) as a dilation-2 gather: Ho = 2 Hi, pad 2; one 2 x 2 convolution per output parity class
    assert plan(32, 32, 32, 256, 64, 64, 256, 4, 1, 2, 2)[0] == 2
    assert plan(192, 8, 8, 2048, 16, 16, 256, 4, 1, 2, 2)[0] == 2
    assert plan(32, 64, 64, 128, 128, 128, 64, 4, 1, 2, 2)[0] != 2          # 64-wide inputs: not built
    assert plan(32, 32, 32, 128, 64, 64, 64, 3, 1, 2, 1)[0] != 2            # 3 x 3 under dilation 2 (backward of a stride-2 layer)
    # ReflectionPad2d(1) + Conv2d(3, padding=0): Hi = Ho + 2
    assert plan(64, 34, 34, 256, 32, 32, 256, 3, 1, 1, 0)[0] == 2
    assert plan(64, 33, 34, 256, 31, 32, 256, 3, 1, 1, 0)[0] != 2 or (31 * 32) % 128 == 0      # tiles must be whole rows of one image


def test_switches_and_deterministic_mode(monkeypatch):
    shape = (32, 8, 8, 512, 8, 8, 512, 3)
    assert plan(*shape) == (2, 4)
    monkeypatch.setenv("DSF_X6_PATCH", "1")                 # 64-wide maps only
    assert plan(*shape)[0] != 2 and plan(32, 64, 64, 64, 64, 64, 64, 3)[0] == 2
    monkeypatch.setenv("DSF_X6_PATCH", "0")
    assert plan(32, 64, 64, 64, 64, 64, 64, 3)[0] == 1
    monkeypatch.delenv("DSF_X6_PATCH")
    L.lib().dsf_set_deterministic(I(1))                     # never splits a reduction
    try:
        assert plan(*shape) == (2, 1)
    finally:
        L.lib().dsf_set_deterministic(I(0))


def test_bad_arguments_are_refused():
    v = ctypes.c_int(0)
    rc = L.lib().dsf_conv_x6_forward_plan(I(0), I(8), I(8), I(16), I(8), I(8), I(16), I(3), I(3), I(1), I(1), I(1), I(1), ctypes.byref(v), None)
    assert rc != 0
    rc = L.lib().dsf_conv_x6_forward_plan(I(1), I(8), I(8), I(16), I(8), I(8), I(16), I(3), I(3), I(1), I(3), I(1), I(1), ctypes.byref(v), None)
    assert rc != 0


def test_stem_node_launches_have_names_and_unsupported_geometries_are_refused():
    """the record kinds of nn_norm._StemFunction map onto the kernels rocprofv3 prints (bench.py's per-kernel table), and the stem entry
    points refuse what they do not implement before touching any pointer (status codes only: no GPU here)"""
    from dsf_amd import nn_conv
    rec = lambda kind: (kind, 32, 128, 128, 1, 128, 128, 64, 5, 5, 1, 1, 2, 2)
    assert nn_conv.kernel_name(rec("c1_fwd_bn")) == "conv_c1_fwd_stats_kernel<5, 1>"
    assert nn_conv.kernel_name(rec("c1_wrw_bn1")) == "conv_c1_wrw_bn_kernel<5, 1, 1>"
    assert nn_conv.kernel_name(("c1_wrw_bn0", 64, 256, 256, 1, 128, 128, 64, 7, 7, 2, 1, 3, 3)) == "conv_c1_wrw_bn_kernel<7, 2, 0>"
    lib = L.lib()
    one = ctypes.c_void_p(16)                                # a non-NULL address that is never dereferenced: the geometry is refused first
    UNSUPPORTED, BAD = lib.dsf_conv_c1_wrw_bn(one, one, one, one, one, one, one, one, one, I(8), I(1), I(3), I(1), I(1), one, one, one, I(0), one, I(2),
                                              I(16), I(16), I(16), I(16), I(64), I(5), I(1), I(2), ctypes.c_void_p(0)), \
        lib.dsf_conv_c1_wrw_bn(one, one, one, None, one, one, one, one, one, I(8), I(1), I(3), I(2), I(1), one, one, one, I(0), one, I(2),
                               I(16), I(16), I(16), I(16), I(64), I(5), I(1), I(2), ctypes.c_void_p(0))
    assert UNSUPPORTED != 0 and BAD != 0 and UNSUPPORTED != BAD          # pooling (3, 1, 1): not implemented; a pooled call without argmax: bad argument
    # 5 x 5 windows, more than 2 x 2 windows per pixel, a channel count the kernels do not cover
    for k, s, p, C in ((5, 2, 2, 64), (3, 1, 1, 64), (3, 2, 1, 6)):
        assert lib.dsf_bn_relu_pool_forward(one, one, one, I(2), I(16), I(16), I(C), I(k), I(s), I(p), ctypes.c_float(1e-5), ctypes.c_float(0.1), None, None,
                                            one, one, one, one, one, I(0), ctypes.c_void_p(0)) == UNSUPPORTED
