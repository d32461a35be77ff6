"""K11 fp32 MFMA implicit-GEMM convolution vs torch's CPU convolution (fp32), forward and all
three gradients.  Tolerance: 1e-4 of the largest reference magnitude (accumulation-order noise only;
the f32 MFMA is an exact fmaf chain)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # Ci, Co, K, stride, pad, H, bias
    (64, 64, 3, 1, 1, 16, False),
    (1, 64, 5, 1, 2, 32, False),          # stem (scalar gather path)
    (64, 128, 3, 2, 1, 16, False),
    (64, 128, 1, 2, 0, 16, False),        # downsample 1x1 stride 2
    (488, 256, 3, 1, 1, 8, True),         # stage-2 fusion conv (Ci % 32 != 0)
    (256, 63, 1, 1, 0, 8, True),          # offset head (Co % 4 != 0)
    (256, 21, 1, 1, 0, 8, True),
    (1, 64, 7, 2, 3, 32, True),           # hourglass stem
    (64, 1, 7, 1, 0, 22, True),           # generator output conv (Co = 1)
    (128, 256, 3, 1, 1, 9, False),        # odd spatial size -> M tail
    (1, 36, 5, 2, 2, 29, True),           # 1-channel direct kernels (conv_c1.hip): partial lanes, ragged row segments
    (1, 64, 7, 1, 3, 21, False),
    (1, 8, 5, 1, 0, 13, True),
]


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.mark.parametrize("Ci,Co,K,s,p,H,bias", CASES)
def test_conv2d_matches_torch_cpu(Ci, Co, K, s, p, H, bias):
    from dsf_amd.nn_conv import Conv2dFunction
    g = torch.Generator().manual_seed(Ci * 131 + Co * 7 + K)
    B = 3
    x = torch.randn(B, Ci, H, H, generator=g, requires_grad=True)
    w = (torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5).requires_grad_(True)
    b = torch.randn(Co, generator=g).requires_grad_(True) if bias else None
    y = F.conv2d(x, w, b, stride=s, padding=p)
    gy = torch.randn(y.shape, generator=g)
    grads = torch.autograd.grad((y * gy).sum(), [x, w] + ([b] if bias else []))
    xg = x.detach().cuda().requires_grad_(True)
    wg = w.detach().cuda().requires_grad_(True)
    bg = b.detach().cuda().requires_grad_(True) if bias else None
    yg = Conv2dFunction.apply(xg, wg, bg, s, (p, p))
    assert yg.shape == y.shape
    assert _rel(yg.cpu(), y.detach()) < 1e-4
    gg = torch.autograd.grad((yg * gy.cuda()).sum(), [xg, wg] + ([bg] if bias else []))
    for a, r in zip(gg, grads):
        assert _rel(a.cpu(), r) < 1e-4


@pytest.mark.parametrize("Ci,Co,K,s,p,op,H", [(512, 256, 4, 2, 1, 0, 4), (256, 128, 3, 2, 1, 1, 8), (64, 32, 4, 2, 1, 0, 7)])
def test_conv_transpose2d_matches_torch_cpu(Ci, Co, K, s, p, op, H):
    from dsf_amd.nn_conv import ConvTranspose2dFunction
    g = torch.Generator().manual_seed(Ci + Co + K)
    B = 2
    x = torch.randn(B, Ci, H, H, generator=g, requires_grad=True)
    w = (torch.randn(Ci, Co, K, K, generator=g) / (Ci * K * K) ** 0.5).requires_grad_(True)
    b = torch.randn(Co, generator=g).requires_grad_(True)
    y = F.conv_transpose2d(x, w, b, stride=s, padding=p, output_padding=op)
    gy = torch.randn(y.shape, generator=g)
    grads = torch.autograd.grad((y * gy).sum(), [x, w, b])
    xg, wg, bg = (t.detach().cuda().requires_grad_(True) for t in (x, w, b))
    yg = ConvTranspose2dFunction.apply(xg, wg, bg, s, (p, p), (op, op))
    assert yg.shape == y.shape
    assert _rel(yg.cpu(), y.detach()) < 1e-4
    for a, r in zip(torch.autograd.grad((yg * gy.cuda()).sum(), [xg, wg, bg]), grads):
        assert _rel(a.cpu(), r) < 1e-4


@pytest.mark.parametrize("acc", [False, True])
@pytest.mark.parametrize("C,H,res,relu", [(64, 16, True, True), (256, 8, False, True), (512, 4, True, False), (8, 5, False, False),
                                          (1024, 3, True, True), (2048, 4, True, True), (2048, 3, False, False),
                                          # one channel quad shared by all 256 threads; ragged row batches; > 2 M float4 (several batches per thread)
                                          (4, 7, True, True), (128, 37, False, True), (64, 153, True, True)])
def test_fused_batchnorm_matches_torch_cpu(C, H, res, relu, acc):
    """bn(x) (+ residual) (relu) in training mode vs torch CPU: output, running stats, all gradients -- on the ordered-partials
    path (reduce, finalise, apply) and, ``acc``, on the finalise-free path (float-atomic accumulation rows from an open
    ``stat_pool``, folded in the apply kernels' prologue: dsf_bn_forward_acc / dsf_bn_backward_acc)."""
    from dsf_amd import nn_norm, _lib as L
    from dsf_amd.nn_norm import FusedBatchNorm2d
    import contextlib
    if acc and L.deterministic():
        pytest.skip("deterministic mode keeps the ordered-partials path")
    pool = nn_norm.stat_pool(2 * nn_norm.acc_rows() * 2 * C, "cuda") if acc else contextlib.nullcontext()
    with pool:
        _bn_case(C, H, res, relu, FusedBatchNorm2d)
        if acc:
            assert nn_norm._ACC_POOL is not None and nn_norm._ACC_POOL[1] == 2 * nn_norm.acc_rows() * 2 * C     # forward + backward blocks taken
    assert nn_norm._ACC_POOL is None


@pytest.mark.parametrize("acc", [False, True])
@pytest.mark.parametrize("M,C,res,relu", [(255, 64, True, True), (256, 4, False, True), (257, 64, True, False), (1024, 128, False, True),
                                          (1025, 8, True, True), (600, 256, True, True), (17, 2048, False, False)])
def test_small_map_batchnorm_one_launch(M, C, res, relu, acc):
    """Maps of up to 1024 rows take ONE launch per pass (a workgroup per channel quad holds the column in registers: 1 or 4 float4
    per thread): the boundaries of the two instantiations and the first size beyond them, against torch CPU."""
    from dsf_amd import nn_norm, _lib as L
    from dsf_amd.nn_norm import FusedBatchNorm2d
    import contextlib
    if acc and L.deterministic():
        pytest.skip("deterministic mode keeps the ordered-partials path")
    with (nn_norm.stat_pool(2 * nn_norm.acc_rows() * 2 * C, "cuda") if acc else contextlib.nullcontext()):
        _bn_case(C, M, res, relu, FusedBatchNorm2d, shape=(1, C, 1, M))


def _bn_case(C, H, res, relu, FusedBatchNorm2d, shape=None):
    g = torch.Generator().manual_seed(C + H)
    B = 6
    shape = shape or (B, C, H, H)
    x = (torch.randn(shape, generator=g) * 2 + 0.5).requires_grad_(True)
    r = torch.randn(shape, generator=g).requires_grad_(True) if res else None
    ref = torch.nn.BatchNorm2d(C, momentum=0.1)
    with torch.no_grad():
        ref.weight.copy_(torch.randn(C, generator=g)); ref.bias.copy_(torch.randn(C, generator=g))
    fused = FusedBatchNorm2d(C, momentum=0.1).cuda()
    fused.load_state_dict(ref.state_dict())
    y = ref(x)
    if res:
        y = y + r
    if relu:
        y = F.relu(y)
    gy = torch.randn(y.shape, generator=g)
    inputs = [x, ref.weight, ref.bias] + ([r] if res else [])
    grads = torch.autograd.grad((y * gy).sum(), inputs)
    xg = x.detach().cuda().requires_grad_(True)
    rg = r.detach().cuda().requires_grad_(True) if res else None
    yg = fused(xg, rg, relu)
    assert _rel(yg.cpu(), y.detach()) < 1e-5
    gin = [xg, fused.weight, fused.bias] + ([rg] if res else [])
    gg = torch.autograd.grad((yg * gy.cuda()).sum(), gin)
    for a, b_ in zip(gg, grads):
        assert _rel(a.cpu(), b_) < 1e-4
    assert _rel(fused.running_mean.cpu(), ref.running_mean) < 1e-5
    assert _rel(fused.running_var.cpu(), ref.running_var) < 1e-5
    assert int(fused.state_dict()["num_batches_tracked"]) == 1            # deferred add, folded in when the state is read
    # eval mode: frozen statistics
    ref.eval(); fused.eval()
    with torch.no_grad():
        ye = ref(x)
        assert _rel(fused(xg.detach()).cpu(), ye) < 1e-5


# ---- conv_x6: fp32 products on the bf16 matrix cores by exact three-way operand splitting --------------------------------
X6_CASES = [
    # Ci, Co, K, stride, pad, H, B
    (488, 256, 3, 1, 1, 16, 4),           # stage-2 fusion conv: Ci % 16 != 0 (zero-padded last chunk), two n tiles
    (64, 64, 3, 1, 1, 64, 2),
    (256, 84, 1, 1, 0, 32, 2),            # merged heads: ragged n tile
    (20, 36, 3, 2, 1, 17, 3),             # ragged everything, stride 2
    (512, 512, 3, 1, 1, 8, 32),           # small map: split-K with float atomics
    (256, 256, 4, 2, 1, 32, 2),           # ConvTranspose2d backward-data geometry
]


@pytest.mark.parametrize("Ci,Co,K,s,p,H,B", X6_CASES)
def test_x6_matches_float64_as_closely_as_the_fp32_mfma(Ci, Co, K, s, p, H, B, monkeypatch):
    """Both kernels against a float64 convolution: the split path must be in the same error class as the fp32 MFMA
    (its products are exact to 2^-26; what is left is fp32 accumulation order), far inside the 1e-4 parity bar."""
    from dsf_amd import nn_conv
    g = torch.Generator().manual_seed(Ci + 3 * Co + K)
    x = torch.randn(B, Ci, H, H, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5).cuda().requires_grad_(True)
    gy = None
    out = {}
    for math in ("x6", "f32"):
        monkeypatch.setattr(nn_conv, "MATH", math)
        nn_conv.RECORD = []
        y = nn_conv.Conv2dFunction.apply(x, w, None, s, (p, p))
        gy = torch.randn(y.shape, generator=g).cuda() if gy is None else gy
        gx, gw = torch.autograd.grad((y * gy).sum(), [x, w])
        kinds = {r[0] for r in nn_conv.RECORD}
        nn_conv.RECORD = None
        assert ("x6" in kinds) == (math == "x6")
        out[math] = (y.detach().double().cpu(), gx.double().cpu(), gw.double().cpu())
    xd = x.detach().double().cpu().requires_grad_(True)
    wd = w.detach().double().cpu().requires_grad_(True)
    yd = F.conv2d(xd, wd, None, stride=s, padding=p)
    gxd, gwd = torch.autograd.grad((yd * gy.double().cpu()).sum(), [xd, wd])
    from dsf_amd import _lib as L
    # deterministic mode (DSF_DETERMINISTIC=1 in the environment) does not split the reduction: one fp32 accumulation chain
    # over all of K instead of several shorter ones, hence a slightly larger -- still accumulation-order -- error
    # (round 6: the small-map weight gradients split their pixels towards ONE workgroup per CU instead of two -- chains twice as long:
    #  2.0e-6 on the 8 x 8 x 512 layer, was 1.6e-6)
    bar = 4e-6 if L.deterministic() else 3e-6
    for i, ref in enumerate((yd.detach(), gxd, gwd)):
        e6, e32 = _rel(out["x6"][i], ref), _rel(out["f32"][i], ref)
        assert e6 < bar, (i, e6)
        assert e6 < 3 * e32 + 1e-7, (i, e6, e32)


@pytest.mark.parametrize("Ci,Co,H,W,B,bias", [
    (64, 64, 8, 64, 2, False),        # 256-pixel tiles, two per image: top and bottom image borders inside the patch
    (20, 36, 12, 64, 1, True),        # ragged chunk (20 channels), ragged n tile, three tiles per image
    (36, 130, 64, 64, 6, False),      # 128-wide n tiles (two, the second ragged), odd chunk count
    (488, 256, 64, 64, 6, True),      # the stage-2 fusion layer: half-filled last chunk
    (36, 130, 2, 64, 3, False),       # 64-row tiles of one image row each
    (64, 64, 64, 64, 4, False),
    (128, 128, 32, 32, 48, False),    # 32-wide maps, 128-row tiles (four image rows)
    (20, 130, 32, 32, 2, True),       # 32-wide maps, 64-row tiles
    (36, 130, 16, 16, 3, False),      # 16-wide maps: a fragment block covers two image rows
    (40, 200, 8, 8, 5, True),         # 8-wide maps: a tile is one image, patch rows padded to 24 granules
    (256, 256, 12, 16, 2, False),     # non-square: three tiles per image
])
def test_patch_staged_3x3_kernel_is_bit_identical_to_the_per_tap_gather(Ci, Co, H, W, B, bias, monkeypatch):
    """igemm_x6p_kernel (input staged once per 16-channel chunk as a patch with halo, read by all nine taps) against the per-tap
    gather kernels (DSF_X6_PATCH=0) on 64 / 32 / 16 / 8-wide maps: same reduction order, so with an unsplit reduction
    (deterministic mode) forward and input gradient (the mode-1 image through the same kernel) are BITWISE equal; with the K
    splits the launcher chooses otherwise (float atomics; splits finer than the channel chunks fall back to the gather kernels)
    the result stays at accumulation-order distance from float64.  Unsplit, the shape really takes the patch kernel
    (dsf_conv_x6_forward_plan: variant 2)."""
    import ctypes
    from dsf_amd import nn_conv, _lib as L
    I = ctypes.c_int
    monkeypatch.setattr(nn_conv, "MATH", "x6")
    g = torch.Generator().manual_seed(Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(Co, generator=g).cuda().requires_grad_(True) if bias else None
    gy = torch.randn(B, Co, H, W, generator=g).cuda()
    def variant(ci, co):
        v = ctypes.c_int(-1)
        assert L.lib().dsf_conv_x6_forward_plan(I(B), I(H), I(W), I(ci), I(H), I(W), I(co), I(3), I(3), I(1), I(1), I(1), I(1), ctypes.byref(v), None) == 0
        return v.value
    xd, wd = x.detach().double().cpu().requires_grad_(True), w.detach().double().cpu()
    ref = F.conv2d(xd, wd, b.detach().double().cpu() if bias else None, padding=1)
    gxd, = torch.autograd.grad((ref * gy.double().cpu()).sum(), [xd])
    was = L.set_deterministic(True)
    try:
        out = {}
        for patch in ("0", "2"):
            monkeypatch.setenv("DSF_X6_PATCH", patch)
            assert (variant(Ci, Co) == 2) == (patch == "2")           # (the backward-data launch is planned on its own shape)
            assert patch == "2" or variant(Co, Ci) != 2
            y = nn_conv.Conv2dFunction.apply(x, w, b, 1, (1, 1))
            gx, = torch.autograd.grad((y * gy).sum(), [x])
            out[patch] = (y.detach(), gx)
        for a, r in zip(out["2"], out["0"]):
            assert torch.equal(a, r)
        L.set_deterministic(False)
        y = nn_conv.Conv2dFunction.apply(x, w, b, 1, (1, 1))            # the launcher's own K splits
        gx, = torch.autograd.grad((y * gy).sum(), [x])
        assert _rel(y.detach().double().cpu(), ref.detach()) < 4e-6 and _rel(gx.double().cpu(), gxd) < 4e-6
    finally:
        L.set_deterministic(was)


@pytest.mark.parametrize("Ci,Co,H,W,B", [
    (36, 132, 8, 64, 2),          # ragged 32-channel block (4 of 32), two n tiles (the second 4 wide), splits of four rows
    (20, 200, 5, 64, 1),          # one map of five rows: the last split is a single row
    (40, 72, 7, 32, 3),           # 32-wide maps (two chunks per row), maps of odd height
    (36, 132, 16, 16, 5),         # 16-wide maps: rows in pairs, an odd row count in the last split
    (32, 128, 3, 16, 3),
    (36, 60, 8, 64, 2),           # <= 64 output channels: two channel blocks x two tap groups
    (64, 64, 6, 32, 3),
    (200, 520, 64, 64, 4),        # takes the row kernel by the launcher's own rule (splits of 1024+ pixels)
])
def test_row_staged_3x3_backward_weights_against_float64(Ci, Co, H, W, B, monkeypatch):
    """igemm_wrw_x6p_kernel (input rows with halo staged once in a ring, all nine taps per workgroup) against float64 and against
    igemm_wrw_x6_kernel: both within accumulation-order distance of the float64 weight gradient; in deterministic mode (partial
    tiles per split, added in order) two runs are bitwise equal.  DSF_X6_WRW_PATCH=2 takes the row kernel wherever its
    geometry fits, so that the small shapes here exercise it; the last case takes it by the launcher's own rule."""
    import ctypes
    from dsf_amd import nn_conv, _lib as L
    monkeypatch.setattr(nn_conv, "MATH", "x6")
    g = torch.Generator().manual_seed(Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, Co, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    wd = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    ref, = torch.autograd.grad((F.conv2d(x.double(), wd, None, padding=1) * gy.double()).sum(), [wd])
    ref = ref.permute(2, 3, 1, 0).contiguous()                      # the kernel layout [KH][KW][Ci][Co]
    out = {}
    for level in ("0", "2" if (Ci, Co) != (200, 520) else "1"):
        monkeypatch.setenv("DSF_X6_WRW_PATCH", level)
        nn_conv.RECORD = []
        try:
            out[level] = nn_conv._wrw(x, gy, 3, 3, 1, (1, 1))
            name = nn_conv.kernel_name(nn_conv.RECORD[0])
        finally:
            nn_conv.RECORD = None
        assert name.startswith("igemm_wrw_x6p_kernel" if level != "0" else "igemm_wrw_x6_kernel"), name
        assert _rel(out[level].double(), ref) < 3e-6, (level, _rel(out[level].double(), ref))
    was = L.set_deterministic(True)
    try:
        a = nn_conv._wrw(x, gy, 3, 3, 1, (1, 1))
        b = nn_conv._wrw(x, gy, 3, 3, 1, (1, 1))
        assert torch.equal(a, b) and _rel(a.double(), ref) < 3e-6
    finally:
        L.set_deterministic(was)


def test_whole_network_weight_split_launch_writes_the_per_layer_images_bit_for_bit():
    """dsf_conv_x6_split_weights_multi (one launch after an optimizer step; round 6: four granules per thread, 16-byte loads along n
    (forward images) / along k (backward-data images)) against dsf_conv_x6_split_weights layer by layer: the same bytes, for full,
    ragged (partial chunk, partial n tile) and 1 x 1 layers in both modes."""
    import ctypes
    from dsf_amd import _lib as L
    lib = L.lib()
    I, I64, P = ctypes.c_int, ctypes.c_int64, lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(3)
    layers = [(3, 3, 64, 64), (3, 3, 488, 256), (1, 1, 256, 84), (4, 4, 256, 256), (3, 3, 36, 132), (5, 5, 20, 36), (1, 1, 2048, 512)]
    rows, ref, outs, total = [], [], [], 0
    for (KH, KW, Ci, Co) in layers:
        w = torch.randn(KH, KW, Ci, Co, generator=g).cuda()
        for mode in (0, 1):
            Ck, Cn = (Co, Ci) if mode else (Ci, Co)
            n = lib.dsf_conv_x6_image_bytes(I(KH), I(KW), I(Ck), I(Cn))
            a = torch.zeros(n, dtype=torch.uint8, device="cuda")
            b = torch.full((n,), 0xAB, dtype=torch.uint8, device="cuda")
            assert lib.dsf_conv_x6_split_weights(P(w), P(a), I(KH), I(KW), I(Ci), I(Co), I(mode), st) == 0
            rows.append((w.data_ptr(), b.data_ptr(), KH, KW, Ci, Co, mode, total))
            total += lib.dsf_conv_x6_image_granules(I(KH), I(KW), I(Ck), I(Cn))
            ref.append(a); outs.append(b); layers_keep = w
            rows[-1] = rows[-1] + (w,)
    table = torch.tensor([r[:8] for r in rows] + [(0, 0, 0, 0, 0, 0, 0, total)], dtype=torch.int64).cuda()
    assert lib.dsf_conv_x6_split_weights_multi(P(table), I(len(rows)), I64(total), st) == 0
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(ref, outs)):
        assert torch.equal(a, b), (i, rows[i][2:7])


@pytest.mark.parametrize("B,Ci,Co,Ho,bias", [
    (2, 256, 256, 32, False),     # the decoder's largest input gradient (64 x 64 x 256 -> 32 x 32 x 256), 128-row tiles of four output rows
    (5, 64, 128, 16, True),       # 16-wide output, bias
    (4, 512, 256, 8, False),      # 8-wide output: 64-row tiles (1 x 4 waves), several channel chunks
    (3, 36, 130, 16, False),      # ragged: a partial last chunk (36 channels), a partial second n tile
    (32, 256, 256, 8, False),     # few tiles: the launcher splits K -- ranges of (class, chunk) pairs that cut through a class
])
def test_4x4_stride_2_convolution_by_input_parity_classes_against_the_gather_kernel_and_float64(B, Ci, Co, Ho, bias, monkeypatch):
    """igemm_x6p_kernel<.., 4, true> (round 6): a 4 x 4, stride 2, pad 1 convolution -- the input gradient of ConvTranspose2d(4, 2, 1) --
    with the input taken apart into its four parity classes, each a 2 x 2 stride-1 convolution through the patch-staged kernel,
    all four accumulating into one output tile.  Against the per-tap gather kernel (DSF_X6_PATCH=4) and float64: the classes change
    the ORDER of the fp32 accumulation (class-major instead of tap-major), so the two kernels agree to accumulation-order distance,
    not bitwise; the unsplit launch is deterministic run to run."""
    from dsf_amd import nn_conv, _lib as L
    monkeypatch.setattr(nn_conv, "MATH", "x6")
    g = torch.Generator().manual_seed(B + Ci + Co + Ho)
    x = torch.randn(B, Ci, 2 * Ho, 2 * Ho, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, Ci, 4, 4, generator=g) / (Ci * 16) ** 0.5).cuda()
    b = torch.randn(Co, generator=g).cuda() if bias else None
    ref = F.conv2d(x.double(), w.double(), b.double() if bias else None, stride=2, padding=1)
    out = {}
    for level in ("2", "4"):
        monkeypatch.setenv("DSF_X6_PATCH", level)
        nn_conv.RECORD = []
        try:
            out[level] = nn_conv.Conv2dFunction.apply(x, w, b, 2, (1, 1))
            name = nn_conv.kernel_name(nn_conv.RECORD[0])
        finally:
            nn_conv.RECORD = None
        assert (name.startswith("igemm_x6p_kernel") and name.endswith("4, true>")) == (level == "2"), (level, name)
        assert _rel(out[level].double(), ref) < 2e-6, (level, _rel(out[level].double(), ref))
    assert _rel(out["2"].double(), out["4"].double()) < 2e-6
    monkeypatch.setenv("DSF_X6_PATCH", "2")
    was = L.set_deterministic(True)                                # (no K split: plain stores)
    try:
        a = nn_conv.Conv2dFunction.apply(x, w, b, 2, (1, 1))
        c = nn_conv.Conv2dFunction.apply(x, w, b, 2, (1, 1))
        assert torch.equal(a, c) and _rel(a.double(), ref) < 4e-6
    finally:
        L.set_deterministic(was)


def test_transposed_convolution_input_gradient_takes_the_input_parity_launch(monkeypatch):
    """ConvTranspose2d(256, 256, 4, 2, 1) of the decoder (reference model/backbone.py:30-43): its backward-data pass is the 4 x 4
    stride-2 convolution above; gradients against torch's float64 transposed convolution."""
    from dsf_amd import nn_conv
    monkeypatch.setattr(nn_conv, "MATH", "x6")
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 256, 16, 16, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(256, 256, 4, 4, generator=g) / 64).cuda().requires_grad_(True)
    nn_conv.RECORD = []
    try:
        y = nn_conv.ConvTranspose2dFunction.apply(x, w, None, 2, (1, 1), (0, 0))
        gy = torch.randn(y.shape, generator=g).cuda()
        gx, gw = torch.autograd.grad((y * gy).sum(), [x, w])
        names = [nn_conv.kernel_name(r) for r in nn_conv.RECORD]
    finally:
        nn_conv.RECORD = None
    assert any(n.startswith("igemm_x6p_kernel") and n.endswith("4, true>") for n in names), names
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    yd = F.conv_transpose2d(xd, wd, None, stride=2, padding=1)
    gxd, gwd = torch.autograd.grad((yd * gy.double()).sum(), [xd, wd])
    assert _rel(y.detach().double(), yd.detach()) < 2e-6 and _rel(gx.double(), gxd) < 2e-6 and _rel(gw.double(), gwd) < 3e-6


@pytest.mark.parametrize("Ci,Co,H,W,B,K,stride,pad", [
    (128, 128, 32, 32, 4, 3, 1, 1),     # small-map 3 x 3: the layers with the largest share of a config-2 step
    (64, 256, 16, 16, 5, 3, 2, 1),      # stride 2: input and output maps differ
    (256, 256, 16, 16, 3, 4, 2, 1),     # the transposed convolutions' weight gradient: 4 x 4, stride 2
    (36, 60, 8, 8, 7, 3, 1, 1),         # ragged channels (partial k and n tiles), 8-wide rows: a chunk spans two rows
    (32, 40, 4, 4, 9, 1, 1, 0),         # 4 x 4 maps: a 16-pixel chunk is a whole image; 9 chunks unsplit: a tail of 3
    (32, 40, 4, 4, 11, 1, 1, 0),        # 11 chunks: a tail of 5
    (64, 64, 8, 16, 11, 3, 1, 1),       # non-square, a pixel count that leaves 1-5 chunks for the tail of the six-chunk loop
    (48, 64, 12, 16, 3, 3, 1, 1),       # 12 rows: not a power of two
])
def test_backward_weights_three_chunks_ahead_pipeline_against_float64(Ci, Co, H, W, B, K, stride, pad, monkeypatch):
    """igemm_wrw_x6_kernel (round 6: loads three chunks ahead from three register sets over two LDS stages -- a six-chunk pattern
    with a straight-line main loop and a nested tail of 0-5 chunks) against float64 for workgroup targets that move the split
    boundaries, so that every tail length occurs; in deterministic mode (ordered partial tiles) two runs agree to the bit."""
    from dsf_amd import nn_conv, _lib as L
    monkeypatch.setattr(nn_conv, "MATH", "x6")
    monkeypatch.setenv("DSF_X6_WRW_PATCH", "0")
    g = torch.Generator().manual_seed(Ci * 7 + Co + H + K)
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    x = torch.randn(B, Ci, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, Co, Ho, Wo, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    wd = torch.zeros(Co, Ci, K, K, dtype=torch.float64, device="cuda", requires_grad=True)
    ref, = torch.autograd.grad((F.conv2d(x.double(), wd, None, stride=stride, padding=pad) * gy.double()).sum(), [wd])
    ref = ref.permute(2, 3, 1, 0).contiguous()
    for wgs in ("512", "256", "96", "24", "1"):        # (splits of 2 k chunks and a last one of any length: tails 0-5 over the cases)
        monkeypatch.setenv("DSF_X6_WRW_WGS", wgs)
        out = nn_conv._wrw(x, gy, K, K, stride, (pad, pad))                                # float atomics
        assert _rel(out.double(), ref) < 3e-6, (wgs, _rel(out.double(), ref))
        was = L.set_deterministic(True)
        try:
            a = nn_conv._wrw(x, gy, K, K, stride, (pad, pad)).clone()
            b = nn_conv._wrw(x, gy, K, K, stride, (pad, pad))
            assert torch.equal(a, b) and _rel(a.double(), ref) < 3e-6, wgs
        finally:
            L.set_deterministic(was)


@pytest.mark.parametrize("Ci,Co,H,W,B", [
    (64, 128, 16, 16, 3),         # R18 / R50 shortcut: 1 x 1, stride 2
    (256, 512, 32, 32, 8),        # enough rows for 128-row tiles
    (36, 200, 10, 6, 2),          # ragged channels, non-square, a partial last tile
])
def test_backward_data_of_1x1_stride_2_launches_the_live_parity_class_only(Ci, Co, H, W, B, monkeypatch):
    """The input gradient of a 1 x 1 stride-2 convolution is a dilation-2 gather in which one pixel in four has a tap.  The
    launch over that parity class alone (its epilogue stores the three siblings' zeros) against the launch over all four classes
    (DSF_X6_LIVE=0): bitwise equal -- every pixel written exactly once, zeros where there is no tap -- and equal to float64."""
    from dsf_amd import nn_conv, _lib as L
    monkeypatch.setattr(nn_conv, "MATH", "x6")
    g = torch.Generator().manual_seed(Ci + Co)
    x = torch.randn(B, Ci, H, W, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5).cuda().requires_grad_(True)
    gy = torch.randn(B, Co, H // 2, W // 2, generator=g).cuda()
    was = L.set_deterministic(True)                                      # unsplit launches on these small shapes
    try:
        out = {}
        for live in ("0", "1"):
            monkeypatch.setenv("DSF_X6_LIVE", live)
            y = nn_conv.Conv2dFunction.apply(x, w, None, 2, (0, 0))
            gx, = torch.autograd.grad((y * gy).sum(), [x])
            out[live] = gx
    finally:
        L.set_deterministic(was)
    assert torch.equal(out["0"], out["1"])
    assert float(out["1"][:, :, 1::2, :].abs().max()) == 0.0 and float(out["1"][:, :, :, 1::2].abs().max()) == 0.0
    xd, wd = x.detach().double().cpu().requires_grad_(True), w.detach().double().cpu()
    gxd, = torch.autograd.grad((F.conv2d(xd, wd, None, stride=2) * gy.double().cpu()).sum(), [xd])
    assert _rel(out["1"].double().cpu(), gxd) < 3e-6


@pytest.mark.parametrize("Ci,Co,H,W,B", [
    (256, 256, 32, 32, 4),        # 64-row tiles (1 x 4 waves)
    (64, 128, 32, 32, 8),         # 128-row tiles
    (36, 132, 16, 16, 3),         # ragged channel chunk and n tile
    (132, 72, 8, 8, 5),           # 8-wide inputs: a class image is one tile
    (40, 130, 12, 16, 2),         # non-square input
])
def test_patch_staged_transposed_4x4_stride_2_is_bit_identical_to_the_gather(Ci, Co, H, W, B, monkeypatch):
    """ConvTranspose2d(k = 4, s = 2, p = 1) runs as a dilation-2 gather in which every output parity class is a 2 x 2 convolution
    over the input grid.  igemm_x6p_kernel with four taps (one input patch per 16-channel chunk, read by the class's four taps)
    against igemm_x6b / x6_kernel (DSF_X6_PATCH=3): same reduction order, unsplit (deterministic mode) => BITWISE equal, forward
    and the input gradient of the matching stride-2 convolution (the same gather with the mode-1 image); both against float64."""
    import ctypes
    from dsf_amd import nn_conv, _lib as L
    I = ctypes.c_int
    monkeypatch.setattr(nn_conv, "MATH", "x6")
    g = torch.Generator().manual_seed(Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=g).cuda().requires_grad_(True)
    wt = (torch.randn(Ci, Co, 4, 4, generator=g) / (4 * Ci) ** 0.5).cuda().requires_grad_(True)          # transposed layout
    xs = torch.randn(B, Co, 2 * H, 2 * W, generator=g).cuda().requires_grad_(True)                      # for the stride-2 convolution
    ws = (torch.randn(Ci, Co, 4, 4, generator=g) / (16 * Co) ** 0.5).cuda().requires_grad_(True)         # (Co_out = Ci, Ci_in = Co)
    gys = torch.randn(B, Ci, H, W, generator=g).cuda()

    def variant():
        v = ctypes.c_int(-1)
        assert L.lib().dsf_conv_x6_forward_plan(I(B), I(H), I(W), I(Ci), I(2 * H), I(2 * W), I(Co), I(4), I(4), I(1), I(2), I(2), I(2),
                                                ctypes.byref(v), None) == 0
        return v.value
    was = L.set_deterministic(True)
    try:
        out = {}
        for level in ("3", "2"):
            monkeypatch.setenv("DSF_X6_PATCH", level)
            assert (variant() == 2) == (level == "2")
            y = nn_conv.ConvTranspose2dFunction.apply(x, wt, None, 2, (1, 1), (0, 0))
            ys = nn_conv.Conv2dFunction.apply(xs, ws, None, 2, (1, 1))
            gxs, = torch.autograd.grad((ys * gys).sum(), [xs])
            out[level] = (y.detach(), gxs)
    finally:
        L.set_deterministic(was)
    for a, r in zip(out["2"], out["3"]):
        assert torch.equal(a, r)
    ref = F.conv_transpose2d(x.detach().double().cpu(), wt.detach().double().cpu(), None, stride=2, padding=1)
    assert _rel(out["2"][0].double().cpu(), ref) < 4e-6
    xsd = xs.detach().double().cpu().requires_grad_(True)
    gxd, = torch.autograd.grad((F.conv2d(xsd, ws.detach().double().cpu(), None, stride=2, padding=1) * gys.double().cpu()).sum(), [xsd])
    assert _rel(out["2"][1].double().cpu(), gxd) < 4e-6


@pytest.mark.parametrize("Ci,Co,H,W,B", [(256, 256, 32, 32, 6), (36, 132, 16, 16, 3), (64, 64, 8, 64, 2), (40, 200, 8, 8, 5)])
def test_patch_staged_3x3_over_a_reflection_padded_input(Ci, Co, H, W, B, monkeypatch):
    """The generator's residual blocks are ReflectionPad2d(1) + Conv2d(3, padding=0): the input carries its own border, every
    patch pixel is inside it.  igemm_x6p_kernel against the gather kernels (DSF_X6_PATCH=0), unsplit: bitwise equal, forward and
    input gradient (a pad-2 gather: stays on the gather kernel either way); against float64."""
    import ctypes
    from dsf_amd import nn_conv, _lib as L
    I = ctypes.c_int
    monkeypatch.setattr(nn_conv, "MATH", "x6")
    g = torch.Generator().manual_seed(Ci + Co + H)
    x = torch.randn(B, Ci, H + 2, W + 2, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5).cuda().requires_grad_(True)
    gy = torch.randn(B, Co, H, W, generator=g).cuda()
    was = L.set_deterministic(True)
    try:
        out = {}
        for level in ("0", "2"):
            monkeypatch.setenv("DSF_X6_PATCH", level)
            v = ctypes.c_int(-1)
            assert L.lib().dsf_conv_x6_forward_plan(I(B), I(H + 2), I(W + 2), I(Ci), I(H), I(W), I(Co), I(3), I(3), I(1), I(1), I(0), I(0),
                                                    ctypes.byref(v), None) == 0
            assert (v.value == 2) == (level == "2")
            y = nn_conv.Conv2dFunction.apply(x, w, None, 1, (0, 0))
            gx, = torch.autograd.grad((y * gy).sum(), [x])
            out[level] = (y.detach(), gx)
    finally:
        L.set_deterministic(was)
    for a, r in zip(out["2"], out["0"]):
        assert torch.equal(a, r)
    xd = x.detach().double().cpu().requires_grad_(True)
    ref = F.conv2d(xd, w.detach().double().cpu(), None)
    gxd, = torch.autograd.grad((ref * gy.double().cpu()).sum(), [xd])
    assert _rel(out["2"][0].double().cpu(), ref.detach()) < 4e-6 and _rel(out["2"][1].double().cpu(), gxd) < 4e-6


@pytest.mark.parametrize("Ci,K,p,H,W,B,bias", [
    (64, 7, 0, 26, 70, 2, True),          # the generator's last layer on a reflection-padded input (tiles with ragged edges)
    (64, 7, 3, 19, 33, 3, False),         # zero padding instead
    (16, 5, 2, 9, 130, 1, True),          # three tiles wide, one and a bit high
    (8, 3, 1, 8, 64, 4, False),
])
def test_one_output_channel_convolution_matches_float64(Ci, K, p, H, W, B, bias):
    """conv_co1_fwd_kernel (lane = output pixel, fp32 FMAs over an LDS-staged patch) against float64, and its gradients (which
    take the general kernels) through the layer."""
    from dsf_amd import nn_conv
    g = torch.Generator().manual_seed(Ci + K + H)
    x = torch.randn(B, Ci, H, W, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(1, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(1, generator=g).cuda().requires_grad_(True) if bias else None
    nn_conv.RECORD = []
    try:
        y = nn_conv.Conv2dFunction.apply(x, w, b, 1, (p, p))
        kinds = [r[0] for r in nn_conv.RECORD]
    finally:
        nn_conv.RECORD = None
    assert kinds == ["co1_fwd"], kinds
    gy = torch.randn(y.shape, generator=g).cuda()
    gx, gw = torch.autograd.grad((y * gy).sum(), [x, w])
    xd, wd = x.detach().double().cpu().requires_grad_(True), w.detach().double().cpu().requires_grad_(True)
    yd = F.conv2d(xd, wd, b.detach().double().cpu() if bias else None, padding=p)
    gxd, gwd = torch.autograd.grad((yd * gy.double().cpu()).sum(), [xd, wd])
    assert yd.shape == y.shape
    assert _rel(y.detach().double().cpu(), yd.detach()) < 2e-6
    assert _rel(gx.double().cpu(), gxd) < 3e-6 and _rel(gw.double().cpu(), gwd) < 3e-6


@pytest.mark.parametrize("Ci,Co,K,s,p,H,W,B,rows", [
    (36, 132, 3, 1, 1, 8, 64, 2, "2"),        # row-staged kernel (forced), 128-wide tiles: two n tiles, ragged
    (64, 60, 3, 1, 1, 6, 32, 3, "2"),         # row-staged kernel, <= 64 output channels (two tap groups stage the same dY)
    (128, 128, 3, 1, 1, 4, 4, 64, "1"),       # old kernel: hourglass-sized maps
    (256, 128, 1, 1, 0, 8, 8, 64, "1"),       # 1 x 1
    (40, 72, 3, 2, 1, 9, 11, 5, "1"),         # stride 2, ragged everything
    (16, 20, 5, 1, 2, 7, 7, 3, "1"),
])
def test_bias_gradient_from_the_weight_gradient_launch(Ci, Co, K, s, p, H, W, B, rows, monkeypatch):
    """dsf_conv_x6_wrw_bias: the workgroups that stage a dY tile for the first K tile add its column sums into dbias -- the bias
    gradient without the separate column-sum launches.  Through Conv2dFunction's backward (the launch runs on the calling stream
    for layers below the side-stream threshold): the column-sum path is NOT taken where the launcher accepts the layer (the old
    kernel, <= 64 pixel splits) and IS taken where it declines (the row-staged kernel's layers); either way db equals the
    float64 sum and dW is what the plain launch gives; with the gradient pool the vectors come out of the pooled zeros."""
    from dsf_amd import nn_conv, _lib as L
    if nn_conv.MATH != "x6" or L.deterministic():
        pytest.skip("split kernels in float-atomic mode only")
    monkeypatch.setenv("DSF_X6_WRW_PATCH", rows)
    monkeypatch.setattr(nn_conv, "WRW_MIN_WORK", [1e30])              # every layer's dW on the calling stream
    g = torch.Generator().manual_seed(Ci + Co + K)
    x = torch.randn(B, Ci, H, W, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(Co, generator=g).cuda().requires_grad_(True)
    out = {}
    calls = []
    plain_bias_grad = nn_conv._bias_grad
    monkeypatch.setattr(nn_conv, "_bias_grad", lambda gy_: (calls.append(1), plain_bias_grad(gy_))[1])
    for fused in (True, False):
        monkeypatch.setattr(nn_conv, "BIAS_IN_WRW", [fused])
        for pooled in (False, True):
            del calls[:]
            y = nn_conv.Conv2dFunction.apply(x, w, b, s, (p, p))
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).cuda()
            if pooled:
                with nn_conv.grad_pool(w.numel() + Co + 16, x.device):
                    gw, gb = torch.autograd.grad((y * gy).sum(), [w, b])
                    gw, gb = gw.clone(), gb.clone()
            else:
                gw, gb = torch.autograd.grad((y * gy).sum(), [w, b])
            assert len(calls) == (0 if (fused and rows == "1") else 1), (fused, pooled, rows, len(calls))
            out[(fused, pooled)] = (gw, gb)
    ref_b = gy.double().sum((0, 2, 3))
    for key, (gw, gb) in out.items():
        assert _rel(gb.double(), ref_b) < 3e-6, (key, _rel(gb.double(), ref_b))
        assert _rel(gw.double(), out[(False, False)][0].double()) < 3e-6, key


def test_x6_weight_images_follow_the_weights(monkeypatch):
    """The split image of a weight is kept from one use to the next only for MANAGED parameters (FusedAdamW's, EvalStep's):
    in-place torch updates (version counter) and FusedAdamW's raw-pointer updates (nn_conv.weights_changed) both invalidate
    it, an unchanged managed weight is not split again.  An unmanaged parameter is re-split at every use, so a write
    through ``.data`` -- invisible to torch's version counter -- can never be served from a stale image."""
    from dsf_amd import nn_conv
    from dsf_amd.optim import FusedAdamW
    monkeypatch.setattr(nn_conv, "MATH", "x6")            # (the suite also runs under DSF_CONV_MATH=f32)
    torch.manual_seed(0)
    conv = nn_conv.Conv2d(32, 48, 3, padding=1, bias=False).cuda()
    x = torch.randn(2, 32, 12, 12, device="cuda")
    ref = lambda: F.conv2d(x.double(), conv.weight.detach().double(), padding=1)
    # unmanaged: forward, write through .data, forward must change
    y0 = conv(x)
    v0 = conv.weight._version
    conv.weight.data.mul_(2.0)
    assert conv.weight._version == v0                       # torch did not see the write ...
    y1 = conv(x)
    assert _rel(y1.double(), 2.0 * y0.double()) < 1e-6 and _rel(y1.double(), ref()) < 2e-6      # ... the layer did
    # init_weights-style re-initialisation after a forward
    with torch.no_grad():
        conv.weight.normal_(0, 0.1)
    assert _rel(conv(x).double(), ref()) < 2e-6
    # managed: cached until announced
    opt = FusedAdamW(conv.parameters(), lr=0.1)
    assert conv.weight.__dict__["_dsf_managed"]
    y0 = conv(x)
    img = conv.weight.__dict__["_dsf_x6"][0]
    conv(x)
    assert conv.weight.__dict__["_dsf_x6"][0] is img and _rel(y0.double(), ref()) < 2e-6     # same cache entry: no re-split
    with torch.no_grad():
        conv.weight.mul_(-2.0)
    assert _rel(conv(x).double(), ref()) < 2e-6
    conv(x).square().mean().backward()
    opt.step()
    assert _rel(conv(x).double(), ref()) < 2e-6
    conv.weight.data.mul_(0.5)
    nn_conv.weights_changed()                               # the documented way to announce a .data write on managed weights
    assert _rel(conv(x).double(), ref()) < 2e-6
    # a frozen, managed layer beside the optimizer's (the transfer generator of FinetuneStageStep): the optimizer announces ITS
    # parameters only, so the bystander's image survives the step (round 5 re-split all 43 generator layers every step) ...
    frozen = nn_conv.Conv2d(32, 32, 3, padding=1, bias=False).cuda().requires_grad_(False)
    nn_conv.manage_weights(frozen.parameters())
    yf = frozen(x)
    entry = frozen.weight.__dict__["_dsf_x6"][0]
    conv(x).square().mean().backward()
    opt.step()
    assert torch.equal(frozen(x), yf) and frozen.weight.__dict__["_dsf_x6"][0] is entry
    # ... and an unannounced write somewhere (weights_changed() without arguments) still invalidates it
    frozen.weight.data.mul_(3.0)
    nn_conv.weights_changed()
    assert _rel(frozen(x).double(), F.conv2d(x.double(), frozen.weight.detach().double(), padding=1)) < 2e-6


def test_x6_random_geometries_against_float64():
    """Seeded sweep over ragged geometries (non-square maps, odd sizes, every channel-count class the split kernels accept,
    kernel 1..5, stride 1 / 2, transposed convolutions): forward and all gradients against float64."""
    from dsf_amd import nn_conv
    rng = np.random.RandomState(7)
    saved, nn_conv.RECORD = nn_conv.RECORD, []
    try:
        for case in range(36):
            Ci = int(rng.choice([16, 20, 36, 64, 100, 132]))
            Co = int(rng.choice([1, 4, 36, 64, 72, 130]))
            K = int(rng.choice([1, 2, 3, 4, 5]))
            s = int(rng.choice([1, 2]))
            p = int(rng.randint(0, K))
            H, W, B = int(rng.randint(K + 1, 20)), int(rng.randint(K + 1, 20)), int(rng.choice([1, 3]))
            transposed = case % 3 == 2 and Co % 4 == 0
            g = torch.Generator().manual_seed(case)
            x = torch.randn(B, Ci, H, W, generator=g).cuda().requires_grad_(True)
            if transposed:
                op = int(rng.randint(0, s))
                w = (torch.randn(Ci, Co, K, K, generator=g) / (Ci * K * K) ** 0.5).cuda().requires_grad_(True)
                if (H - 1) * s - 2 * p + K + op < 1 or (W - 1) * s - 2 * p + K + op < 1:
                    continue
                y = nn_conv.ConvTranspose2dFunction.apply(x, w, None, s, (p, p), (op, op))
                ref_fn = lambda xd, wd: F.conv_transpose2d(xd, wd, None, stride=s, padding=p, output_padding=op)
            else:
                w = (torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5).cuda().requires_grad_(True)
                y = nn_conv.Conv2dFunction.apply(x, w, None, s, (p, p))
                ref_fn = lambda xd, wd: F.conv2d(xd, wd, None, stride=s, padding=p)
            gy = torch.randn(y.shape, generator=g).cuda()
            gx, gw = torch.autograd.grad((y * gy).sum(), [x, w])
            xd = x.detach().double().cpu().requires_grad_(True)
            wd = w.detach().double().cpu().requires_grad_(True)
            yd = ref_fn(xd, wd)
            assert yd.shape == y.shape, (case, yd.shape, y.shape)
            gxd, gwd = torch.autograd.grad((yd * gy.double().cpu()).sum(), [xd, wd])
            for name, a, r in (("y", y, yd), ("gx", gx, gxd), ("gw", gw, gwd)):
                assert _rel(a.detach().double().cpu(), r.detach()) < 3e-6, (case, name, Ci, Co, K, s, p, H, W, B, transposed)
        kinds = [r[0] for r in nn_conv.RECORD]
        assert nn_conv.MATH != "x6" or kinds.count("x6") >= 30, kinds          # the sweep really exercised the split kernels
    finally:
        nn_conv.RECORD = saved


@pytest.mark.parametrize("kind,Ci,Co,K,s,p,H,B,res", [
    ("conv", 64, 64, 3, 1, 1, 64, 16, True),         # BasicBlock conv2 + skip (BN 64: the 256-row tile)
    ("conv", 64, 128, 3, 2, 1, 64, 32, False),       # stride 2 (64-row tiles)
    ("conv", 64, 128, 1, 2, 0, 32, 4, False),        # downsample 1x1
    ("conv", 128, 128, 3, 1, 1, 33, 31, True),       # M tail: rows of the last tile past M must not count
    ("conv", 256, 512, 3, 1, 1, 32, 12, False),      # four column tiles
    ("conv", 512, 2048, 1, 1, 0, 32, 3, False),      # Bottleneck expansion (C = 2048: two BN column blocks)
    ("deconv", 512, 256, 4, 2, 1, 32, 6, False),     # decoder transposed convolution (dil-2 gather)
    ("deconv", 128, 64, 4, 2, 1, 32, 13, False),
])
def test_bn_statistics_from_the_conv_epilogue_equal_the_reduction_pass(kind, Ci, Co, K, s, p, H, B, res):
    """dsf_conv_x6_forward_bn + dsf_bn_forward_from_stats (statistics as per-tile partial rows written by the convolution
    epilogue) against conv -> dsf_bn_forward: same output / saved statistics / running buffers to accumulation-order noise
    and the same gradients; the request is honoured (rows > 0) on these shapes."""
    from dsf_amd import nn_conv, nn_norm
    if nn_conv.MATH != "x6":
        pytest.skip("DSF_CONV_MATH=f32: the fp32-MFMA kernels have no statistics epilogue")
    torch.manual_seed(3)
    conv = (nn_conv.Conv2d(Ci, Co, K, s, p, bias=False) if kind == "conv"
            else nn_conv.ConvTranspose2d(Ci, Co, K, stride=s, padding=p, output_padding=0, bias=False)).cuda()
    nn_conv.weights_changed()
    outs = []
    import contextlib
    from dsf_amd import _lib as L_
    variants = (True, False) if L_.deterministic() else (True, False, "acc")
    for fused in variants:
        bn = nn_norm.FusedBatchNorm2d(Co, momentum=0.1).cuda()
        torch.manual_seed(7)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
        torch.manual_seed(11)
        x = (torch.randn(B, Ci, H, H, device="cuda") + 0.5).requires_grad_(True)
        with torch.no_grad():
            Ho = conv(x).shape[-1]
        r = torch.randn(B, Co, Ho, Ho, device="cuda").requires_grad_(True) if res else None
        conv.weight.grad = None
        nn_norm.EPILOGUE_STATS[0] = bool(fused)
        pool = nn_norm.stat_pool(2 * nn_norm.acc_rows() * 2 * Co, "cuda") if fused == "acc" else contextlib.nullcontext()
        try:
            if fused == "acc":                               # finalise-free path: the epilogue must fill the accumulation rows
                with nn_norm.stat_pool(nn_norm.acc_rows() * 2 * Co, "cuda"):
                    req = nn_conv.StatsRequest()
                    req.acc = nn_norm._acc_take(Co, x.device)
                    nn_conv.STATS = req
                    try:
                        with torch.no_grad():
                            yc = conv(x)
                    finally:
                        nn_conv.STATS = None
                    assert req.filled == 1 and req.acc is not None
                    tot = req.acc.view(nn_norm.acc_rows(), 2, Co).double().sum(0)
                    ref_s = torch.stack((yc.double().sum((0, 2, 3)), (yc.double() ** 2).sum((0, 2, 3))))
                    assert ((tot - ref_s).abs() <= 1e-5 * ref_s.abs() + 1e-3).all()
            elif fused:                                      # the request must be honoured, not silently dropped
                req = nn_conv.StatsRequest()
                nn_conv.STATS = req
                try:
                    with torch.no_grad():
                        conv(x)
                finally:
                    nn_conv.STATS = None
                assert req.rows > 0 and req.part is not None
                rows_max = nn_conv.L.lib().dsf_conv_x6_bn_stats_rows(B, Ho, Ho)
                assert req.rows <= rows_max and req.part.numel() == rows_max * 2 * Co
            with pool:
                y = nn_norm.conv_bn_act(conv, bn, x, residual=r, relu=True)
                (y * torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)).sum().backward()
                if fused == "acc":
                    assert nn_norm._ACC_POOL[1] == 2 * nn_norm.acc_rows() * 2 * Co
        finally:
            nn_norm.EPILOGUE_STATS[0] = True
        outs.append((y.detach(), bn.running_mean.clone(), bn.running_var.clone(), x.grad.clone(), conv.weight.grad.clone(),
                     bn.weight.grad.clone(), bn.bias.grad.clone(), None if r is None else r.grad.clone(),
                     int(bn.num_batches_tracked)))
    a, b = outs[0], outs[1]
    assert a[8] == b[8] == 1
    for other in outs[1:]:
        assert other[8] == 1
        for i, (u, v) in enumerate(zip(a[:8], other[:8])):
            if u is None:
                continue
            # outputs and statistics to 2e-5 of the largest value; gradients (i >= 3) ALSO pass on their relative L2 error: a
            # statistic that differs in its last bit flips the recomputed ReLU mask of an element whose pre-activation is ~0
            ok = _rel(u, v) < 2e-5 or (i >= 3 and float((u - v).norm() / v.norm()) < 1e-4)
            assert ok, (i, _rel(u, v), float((u - v).norm() / v.norm()))
    # and against float64 statistics of the convolution output itself
    with torch.no_grad():
        yc = conv(x).double()
    m = yc.mean((0, 2, 3))
    v = yc.var((0, 2, 3), unbiased=True)
    assert _rel(a[1].double(), 0.1 * m) < 1e-5
    assert _rel(a[2].double(), 0.9 + 0.1 * v) < 1e-5


def test_bn_epilogue_request_is_declined_where_the_launch_cannot_serve_it():
    """A convolution with a bias (the BN would have to see y + bias) never gets a statistics request; a launch that splits K
    leaves rows = 0 and conv_bn_act takes the reduction pass -- same numbers either way."""
    from dsf_amd import nn_conv, nn_norm
    torch.manual_seed(5)
    conv = nn_conv.Conv2d(512, 512, 3, 1, 1, bias=False).cuda()      # tiny M, long K: the launcher splits K
    bn = nn_norm.FusedBatchNorm2d(512).cuda()
    nn_conv.weights_changed()
    x = torch.randn(2, 512, 4, 4, device="cuda")
    y = nn_norm.conv_bn_act(conv, bn, x, relu=False)
    with torch.no_grad():
        yc = conv(x).double()
    ref = (yc - yc.mean((0, 2, 3), keepdim=True)) / torch.sqrt(yc.var((0, 2, 3), unbiased=False, keepdim=True) + bn.eps)
    assert _rel(y.detach().double(), ref) < 1e-4
    cb = nn_conv.Conv2d(64, 64, 3, 1, 1, bias=True).cuda()
    bb = nn_norm.FusedBatchNorm2d(64).cuda()
    xb = torch.randn(2, 64, 16, 16, device="cuda")
    yb = nn_norm.conv_bn_act(cb, bb, xb, relu=False)
    with torch.no_grad():
        yc = cb(xb).double()
    ref = (yc - yc.mean((0, 2, 3), keepdim=True)) / torch.sqrt(yc.var((0, 2, 3), unbiased=False, keepdim=True) + bb.eps)
    assert _rel(yb.detach().double(), ref) < 1e-4


@pytest.mark.parametrize("kind,Ci,Co,K,s,p,H,B,res,bias", [
    ("conv", 64, 64, 3, 1, 1, 64, 16, True, False),       # BasicBlock conv2 + skip + ReLU
    ("conv", 64, 128, 3, 2, 1, 64, 32, False, False),
    ("conv", 128, 128, 3, 1, 1, 33, 31, True, False),     # M tail
    ("conv", 488, 256, 3, 1, 1, 32, 12, False, True),     # the stage-2 fusion convolution: bias AND BatchNorm
    ("conv", 512, 2048, 1, 1, 0, 32, 3, True, False),     # Bottleneck expansion + skip
    ("deconv", 512, 256, 4, 2, 1, 32, 6, False, False),
    ("conv", 512, 512, 3, 1, 1, 4, 2, True, False),       # split reduction: the epilogue is declined, same numbers
])
def test_eval_mode_batchnorm_rides_in_the_conv_epilogue(kind, Ci, Co, K, s, p, H, B, res, bias):
    """dsf_conv_x6_forward_affine: conv -> frozen-statistics BatchNorm (+ residual) (+ ReLU) in one launch (nn_norm.conv_bn_act
    under torch.no_grad() in eval mode) against the two-pass composition and against float64; the cached folded (scale, shift)
    follow in-place updates of the statistics / affine parameters, training steps and FusedAdamW-style raw writes."""
    from dsf_amd import nn_conv, nn_norm
    torch.manual_seed(4)
    conv = (nn_conv.Conv2d(Ci, Co, K, s, p, bias=bias) if kind == "conv"
            else nn_conv.ConvTranspose2d(Ci, Co, K, stride=s, padding=p, output_padding=0, bias=False)).cuda()
    bn = nn_norm.FusedBatchNorm2d(Co).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
        bn.running_mean.uniform_(-0.5, 0.5); bn.running_var.uniform_(0.3, 2.0)
    nn_conv.weights_changed()
    bn.eval()
    x = torch.randn(B, Ci, H, H, device="cuda") + 0.3

    def both():
        with torch.no_grad():
            Ho = conv(x).shape[-1]
            r = torch.randn(B, Co, Ho, Ho, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9)) if res else None
            nn_norm.EPILOGUE_AFFINE[0] = True
            fused = nn_norm.conv_bn_act(conv, bn, x, residual=r, relu=True)
            nn_norm.EPILOGUE_AFFINE[0] = False
            try:
                plain = nn_norm.conv_bn_act(conv, bn, x, residual=r, relu=True)
            finally:
                nn_norm.EPILOGUE_AFFINE[0] = True
            yc = conv(x).double()
        ref = (yc - bn.running_mean.double().view(1, -1, 1, 1)) / torch.sqrt(bn.running_var.double().view(1, -1, 1, 1) + bn.eps) \
            * bn.weight.double().view(1, -1, 1, 1) + bn.bias.double().view(1, -1, 1, 1)
        if r is not None:
            ref = ref + r.double()
        return fused, plain, ref.clamp_min(0)
    fused, plain, ref = both()
    assert fused.is_contiguous(memory_format=torch.channels_last)
    assert _rel(fused, plain) < 2e-6 and _rel(fused.double(), ref) < 5e-6
    if Ci * K * K > 4000 and H <= 4:
        return                                                         # (declined launch: nothing cached to invalidate)
    # the folded transform follows every kind of update
    with torch.no_grad():
        bn.running_mean.add_(0.25)                                     # ordinary in-place write: version counter
    fused, plain, ref = both()
    assert _rel(fused.double(), ref) < 5e-6
    bn.train()
    with torch.no_grad():
        nn_norm.conv_bn_act(conv, bn, x, relu=True)                    # a training step rewrites the statistics through raw pointers
    bn.eval()
    fused, plain, ref = both()
    assert _rel(fused, plain) < 2e-6 and _rel(fused.double(), ref) < 5e-6
    # with autograd on, the epilogue path must not be taken (it records no graph): gradients still flow
    xg = x.clone().requires_grad_(True)
    y = nn_norm.conv_bn_act(conv, bn, xg, relu=True)
    y.sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


@pytest.mark.parametrize("Ci,Co,K,H,B", [(128, 128, 3, 4, 16), (256, 128, 1, 2, 64), (256, 256, 3, 8, 8)])
def test_split_launch_adds_into_a_pooled_zero_output(Ci, Co, K, H, B):
    """A small layer is split along K and its partial sums meet in Y by float atomics: Y comes from the step's pooled zero fill
    (nn_conv.zero_pool -> dsf_conv_x6_forward_into) instead of a fill launch per layer.  Same result as the self-filling launch up
    to atomic order, forward and through the backward-data pass; the first pass under an owner only records the demand."""
    import ctypes
    from dsf_amd import nn_conv, _lib as L
    if nn_conv.MATH != "x6":
        pytest.skip("DSF_CONV_MATH=f32")
    if L.deterministic():
        pytest.skip("deterministic mode never splits the reduction")
    I = ctypes.c_int
    assert int(L.lib().dsf_conv_x6_forward_splits(I(B), I(H), I(H), I(Ci), I(Co), I(K), I(K), I(1))) > 1
    bwd_split = int(L.lib().dsf_conv_x6_forward_splits(I(B), I(H), I(H), I(Co), I(Ci), I(K), I(K), I(1))) > 1    # the backward-data launch
    g = torch.Generator().manual_seed(Ci + Co + K + H)
    x = torch.randn(B, Ci, H, H, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(Co, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(B, Co, H, H, generator=g).cuda()

    def run():
        y = nn_conv.Conv2dFunction.apply(x, w, b, 1, (K // 2, K // 2))
        gx, = torch.autograd.grad((y * gy).sum(), [x])
        return y.detach(), gx

    y0, gx0 = run()

    class Owner:
        pass
    owner = Owner()
    with nn_conv.zero_pool(owner, x.device):
        y1, gx1 = run()                                          # records the demand, fills per layer
        assert nn_conv._ZERO[0] is None
    want = y0.numel() + (gx0.numel() if bwd_split else 0)
    assert owner._zero_pool_floats == want
    with nn_conv.zero_pool(owner, x.device):
        pool = nn_conv._ZERO[0]
        y2, gx2 = run()
        assert nn_conv._ZERO[1] == want                          # every split launch took its output from the pool
    lo, hi = pool.data_ptr(), pool.data_ptr() + pool.numel() * 4
    assert lo <= y2.data_ptr() < hi and (lo <= gx2.data_ptr() < hi) == bwd_split and not (lo <= y1.data_ptr() < hi)
    ref = F.conv2d(x.detach().double().cpu(), w.detach().double().cpu(), b.detach().double().cpu(), padding=K // 2)
    for y in (y0, y1, y2):
        assert _rel(y.double().cpu(), ref) < 2e-6
    assert _rel(gx2, gx0) < 2e-6 and _rel(gx1, gx0) < 2e-6
    # an unsplit request is refused: the unsplit kernels store Y
    out = torch.zeros_like(y0)
    rc = L.lib().dsf_conv_x6_forward_into(nn_conv.ptr_nhwc(x.detach().contiguous(memory_format=torch.channels_last)), None,
                                          None, nn_conv.ptr_nhwc(out), I(B), I(H), I(H), I(Ci), I(H), I(H), I(Co), I(K), I(K), I(1),
                                          I(1), I(K // 2), I(K // 2), I(1), None)
    assert rc != 0


@pytest.mark.parametrize("M,C", [(256, 128), (4096, 256), (1000, 12), (64, 256), (3, 64), (16384, 64), (5000, 128), (4096, 63),
                                 (70000, 21), (100, 7)])
def test_bias_gradient_column_sums(M, C):
    """dsf_col_sum: per-channel sums of an (M, C) matrix -- the one-launch kernel for small inputs, the two-launch form above
    1 M elements -- against float64, and bitwise run to run (fixed-order folds)."""
    from dsf_amd import nn_conv
    g = torch.Generator().manual_seed(M + C)
    gy = (torch.randn(1, M, 1, C, generator=g) * 3 + 0.5).cuda().permute(0, 3, 1, 2)          # (1, C, M, 1), channels_last memory
    out = nn_conv._bias_grad(gy)
    ref = gy.double().sum((0, 2, 3))
    assert (out.double() - ref).abs().max().item() < 1e-6 * gy.abs().double().sum((0, 2, 3)).max().item()
    assert torch.equal(out, nn_conv._bias_grad(gy))
