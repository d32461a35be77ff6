"""Edge cases through the product path: empty batches, single samples, shapes at the limits of the vectorised kernels,
off-screen / behind-camera meshes, all-background images."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CAM = (588.03, 587.07, 320.0, 240.0)


@pytest.fixture(scope="module")
def render():
    from dsf_amd.render_model.mano_layer import Render
    return Render("synthetic", "nyu", CAM, (640, 480)).cuda()


def test_empty_batches(render):
    from dsf_amd import ops, nn_conv
    from dsf_amd.metric.meshLoss import ICPLoss
    from dsf_amd.metric.losses import SmoothL1Loss
    from dsf_amd.util.generateFeature import GFM
    mano = render.mano_layer
    z = lambda *s: torch.zeros(*s, device="cuda")
    v, j = mano.get_mano_vertices(z(0, 3), z(0, 45), z(0, 10), z(0, 4), 1 / 125)
    assert v.shape == (0, 779, 3) and j.shape == (0, 21, 3)
    img, juvd, jxyz, mesh = render.render(z(0, 62), z(0, 3), z(0, 3))
    assert img.shape == (0, 1, 128, 128) and juvd.shape == (0, 21, 3)
    assert ICPLoss(z(0, 779, 3), z(0, 16, 3), mano.faces).shape == (0,)
    conv = nn_conv.Conv2d(8, 16, 3, padding=1).cuda()
    assert conv(z(0, 8, 12, 12)).shape == (0, 16, 12, 12)
    maps = GFM().joint2offset(z(0, 21, 3), z(0, 1, 128, 128), 0.8, 64)
    assert maps.shape == (0, 84, 64, 64)
    l = SmoothL1Loss()(z(0, 3), z(0, 3))                      # the reference's mean over nothing is nan as well
    assert torch.isnan(l) or float(l) == 0.0


def test_single_sample_and_odd_batch_match_larger_batch(render):
    """per-sample independence: sample i of a batch of 5 equals the same sample run alone (bit-exact integer outputs)."""
    from dsf_amd.train_step import synthetic_batch
    p, c, cube = synthetic_batch(5, "cuda", seed=9)
    img5, juvd5, _, mesh5 = render.render(p, c, cube)
    for i in (0, 4):
        img1, juvd1, _, mesh1 = render.render(p[i:i + 1], c[i:i + 1], cube[i:i + 1])
        assert torch.equal(img1[0], img5[i]) and torch.equal(mesh1[0], mesh5[i]) and torch.equal(juvd1[0], juvd5[i])


def test_mesh_off_screen_behind_camera_and_all_background(render):
    from dsf_amd import ops
    from dsf_amd.train_step import synthetic_batch
    p, c, cube = synthetic_batch(4, "cuda", seed=3)
    mano = render.mano_layer
    v, _ = mano.get_mano_vertices(p[:, :3], p[:, 3:48], p[:, 48:58], p[:, 58:62], 1 / 125)
    verts = v * cube.unsqueeze(1) / 2 + c.unsqueeze(1)
    verts[1, :, 2] = -verts[1, :, 2]                        # behind the camera: skipped, background
    verts[2, :, 0] += 5000.0                                # far outside the frame
    c2, M, _, _ = ops.crop_setup(c, cube, render.cam, 128)
    minv = torch.linalg.inv_ex(M)[0].contiguous()
    img, p2f = ops.RenderCropFunction.apply(verts.contiguous(), mano.faces_i32, minv, render.resize_rowmap, c2[:, 2].contiguous(),
                                            cube[:, 2].contiguous(), render.cam, 640, 128)
    assert (p2f[1] == -1).all() and (p2f[2] == -1).all() and (img[1] == 1).all() and (img[2] == 1).all()
    assert (p2f[0] >= 0).any() and (p2f[3] >= 0).any()
    # gradient of an all-background image is exactly zero, and finite everywhere
    vg = verts.clone().requires_grad_(True)
    img, _ = ops.RenderCropFunction.apply(vg, mano.faces_i32, minv, render.resize_rowmap, c2[:, 2].contiguous(),
                                          cube[:, 2].contiguous(), render.cam, 640, 128)
    img.sum().backward()
    assert torch.isfinite(vg.grad).all() and not vg.grad[1].any() and not vg.grad[2].any()


@pytest.mark.parametrize("cin,cout,k,s,h", [(4, 4, 3, 1, 5), (36, 132, 3, 1, 9), (32, 4, 1, 1, 7), (260, 64, 3, 2, 17), (64, 68, 4, 2, 16)])
def test_conv_shapes_at_kernel_limits(cin, cout, k, s, h):
    """channel counts that are multiples of 4 but not of the 32 / 64 / 128 tile sizes, odd maps, ragged last tiles"""
    from dsf_amd import nn_conv
    torch.manual_seed(cin * 7 + cout)
    conv = nn_conv.Conv2d(cin, cout, k, stride=s, padding=k // 2).cuda()
    ref = torch.nn.Conv2d(cin, cout, k, stride=s, padding=k // 2)
    ref.load_state_dict(conv.state_dict())
    x = torch.randn(3, cin, h, h)
    xg = x.cuda().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y, yr = conv(xg), ref(xr)
    assert torch.allclose(y.cpu(), yr, rtol=1e-4, atol=1e-4)
    g = torch.randn_like(yr)
    y.backward(g.cuda()); yr.backward(g)
    assert torch.allclose(xg.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(conv.weight.grad.cpu(), ref.weight.grad, rtol=1e-3, atol=1e-3)
    assert torch.allclose(conv.bias.grad.cpu(), ref.bias.grad, rtol=1e-3, atol=1e-3)


def test_new_convolution_entry_points_follow_the_error_convention():
    """Status codes of the split-operand / 1-channel entry points for arguments they do not take: an error status
    (never a wrong answer), which the Python layer turns into RuntimeError."""
    import ctypes
    from dsf_amd import _lib as L
    from dsf_amd._lib import I, ptr, stream_ptr
    lib = L.lib()
    x = torch.randn(2, 8, 8, 18, device="cuda")                    # NHWC, Ci = 18: not a multiple of 4
    img = torch.zeros(lib.dsf_conv_x6_image_bytes(I(3), I(3), I(20), I(16)), dtype=torch.uint8, device="cuda")
    y = torch.empty(2, 8, 8, 16, device="cuda")
    st = lib.dsf_conv_x6_forward(ptr(x), ptr(img), ptr(None), ptr(y), I(2), I(8), I(8), I(18), I(8), I(8), I(16), I(3), I(3), I(1),
                                 I(1), I(1), I(1), I(0), stream_ptr())
    assert st != 0 and b"argument" in lib.dsf_status_string(st).lower()
    x4 = torch.randn(2, 8, 8, 20, device="cuda")
    y_odd = torch.empty(2, 15, 15, 16, device="cuda")              # dil 2 needs even output sizes
    st = lib.dsf_conv_x6_forward(ptr(x4), ptr(img), ptr(None), ptr(y_odd), I(2), I(8), I(8), I(20), I(15), I(15), I(16), I(3), I(3),
                                 I(1), I(2), I(1), I(1), I(0), stream_ptr())
    assert st == 2                                                  # DSF_ERR_UNSUPPORTED
    assert lib.dsf_conv_c1_supported(I(64), I(5), I(5), I(1)) == 1 and lib.dsf_conv_c1_supported(I(65), I(5), I(5), I(1)) == 0
    assert lib.dsf_conv_c1_supported(I(64), I(3), I(3), I(1)) == 0
    st = lib.dsf_conv_c1_forward(ptr(x), ptr(x), ptr(None), ptr(y), I(2), I(8), I(8), I(8), I(8), I(16), I(3), I(1), I(1), stream_ptr())
    assert st == 2
    with pytest.raises(RuntimeError):
        L.check(st, "dsf_conv_c1_forward")
