"""Conv2d / ConvTranspose2d on the hand-written implicit-GEMM kernels (dsf_amd/csrc/conv.hip: fp32 MFMA;
dsf_amd/csrc/conv_x6.hip: the same fp32 products on the bf16 matrix cores by exact three-way operand splitting).
Drop-in subclasses of the torch modules: same parameters, state-dict keys and initialisation; only ``forward``
differs.  Activations are kept channels_last (NHWC in memory), which is what the kernels read and write.  GPU only:
a CPU tensor raises.
"""
import os

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L
from ._lib import I, ptr, check, stream_ptr

CL = torch.channels_last
RECORD = None        # set to a list to log every igemm launch (bench.py replays them for the roofline)


def _nhwc(x):
    if x.dtype != torch.float32:
        x = x.float()
    return x.contiguous(memory_format=CL)


def _fwd(x, wk, bias, out_hw, Co, KH, KW, stride, dil, pad):
    """x: (B,Ci,Hi,Wi) channels_last; wk: [KH][KW][Ci][Co] contiguous -> (B,Co,Ho,Wo) channels_last."""
    B, Ci, Hi, Wi = x.shape
    Ho, Wo = out_hw
    if RECORD is not None:
        RECORD.append(("fwd", B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, dil, pad[0], pad[1]))
    y = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=CL)
    check(L.lib().dsf_conv_igemm_forward(ptr_nhwc(x), ptr(wk), ptr(bias), ptr_nhwc(y), I(B), I(Hi), I(Wi), I(Ci), I(Ho),
                                         I(Wo), I(Co), I(KH), I(KW), I(stride), I(dil), I(pad[0]), I(pad[1]),
                                         stream_ptr()), "dsf_conv_igemm_forward")
    return y


# DSF_CONV_MATH=f32 keeps every convolution on the fp32 MFMA (v_mfma_f32_32x32x2_f32); the default routes the dilation-1
# forward / backward-data GEMMs through conv_x6.hip (six bf16 MFMAs per fp32 product, fp32-grade results, ~1.5x faster).
MATH = os.environ.get("DSF_CONV_MATH", "x6")
_EPOCH = 0


def weights_changed(params=None):
    """Called by whoever rewrites parameters behind torch's version counters (optim.FusedAdamW's raw-pointer kernel, the
    model builders' init_weights, user code that writes through ``.data``): bumps the global write epoch that every cache
    keyed on parameter contents includes (the MANO layer's result memo, BatchNorm's folded affine) and invalidates the split
    weight images of conv_x6 -- all of them, or with ``params`` (the writer knows exactly what it wrote: an optimizer) only
    theirs, so that the images of parameters nobody touched (a frozen transfer generator beside the network being trained: 43
    re-split launches per config-5 step until round 6) survive the step."""
    global _EPOCH
    L.WRITE_EPOCH[0] += 1
    if params is None:
        _EPOCH += 1
    else:
        for p in params:
            p.__dict__["_dsf_wepoch"] = p.__dict__.get("_dsf_wepoch", 0) + 1


def manage_weights(params, managed=True):
    """Declares that every write to ``params`` from now on is announced (in-place torch ops on the parameter itself, which
    bump its version counter, or ``weights_changed()``): only then are the split weight images of conv_x6 kept from one
    use to the next.  optim.FusedAdamW manages the parameters it updates, EvalStep the frozen net it scores.  Unmanaged
    parameters (the default) are re-split at every use, so that a write torch cannot see -- ``w.data.mul_(2)``, a
    ``.data``-style optimizer or EMA -- is never served from a stale image."""
    for p in params:
        p.__dict__["_dsf_managed"] = bool(managed)


_X6_MAX_ELEMS = 0xFFFFFFF0 // 4        # the x6 kernels address their fp32 inputs with 32-bit byte offsets (buffer loads)


def _x6_ok(Ck, n_in=0):
    """reduction width Ck and input element count the conv_x6 forward-type kernel takes (else: the fp32 MFMA kernels)"""
    return MATH == "x6" and Ck % 4 == 0 and Ck >= 16 and n_in <= _X6_MAX_ELEMS


def _wrw_x6_ok(Ci, Co, n_x=0, n_gy=0):
    return MATH == "x6" and Ci % 4 == 0 and Co % 4 == 0 and Ci >= 16 and max(n_x, n_gy) <= _X6_MAX_ELEMS


def _x6_image(weight, wk, mode):
    """bf16x3 image of ``wk`` = [KH][KW][Ci][Co] (a view of ``weight``'s memory): mode 0 the forward operand, mode 1
    the stride-1 backward-data operand.  Cached on the parameter until it changes; temporaries are split per call."""
    KH, KW, Ci, Co = wk.shape
    key = (weight._version, _EPOCH, weight.__dict__.get("_dsf_wepoch", 0), wk.data_ptr())
    cache = weight.__dict__.get("_dsf_x6")
    if cache is not None and mode in cache and cache[mode][0] == key and weight.__dict__.get("_dsf_managed", False):
        return cache[mode][1]
    Ck, Cn = (Co, Ci) if mode else (Ci, Co)
    n = L.lib().dsf_conv_x6_image_bytes(I(KH), I(KW), I(Ck), I(Cn))
    img = cache[mode][1] if (cache is not None and mode in cache and cache[mode][1].numel() == n) else \
        torch.empty(n, device=wk.device, dtype=torch.uint8)
    check(L.lib().dsf_conv_x6_split_weights(ptr(wk), ptr(img), I(KH), I(KW), I(Ci), I(Co), I(mode), stream_ptr()),
          "dsf_conv_x6_split_weights")
    if cache is None:
        cache = weight.__dict__.setdefault("_dsf_x6", {})
    cache[mode] = (key, img, (KH, KW, Ci, Co))
    return img


_JOBS = {}           # id(params list owner) -> (rows, device table)


def refresh_images(params, owner=None):
    """Re-splits every cached conv_x6 image of ``params`` in ONE launch (instead of one small launch per layer and
    direction at their next use): what an optimizer calls after it has rewritten the weights."""
    rows, entries, total = [], [], 0
    for p in params:
        cache = p.__dict__.get("_dsf_x6")
        if not cache or not p.is_cuda:
            continue
        for mode, (key, img, (KH, KW, Ci, Co)) in cache.items():
            if key[-1] != p.data_ptr():                 # the parameter moved (re-laid out): its next use splits it lazily
                continue
            Ck, Cn = (Co, Ci) if mode else (Ci, Co)
            rows.append((p.data_ptr(), img.data_ptr(), KH, KW, Ci, Co, mode, total))
            entries.append((p, mode))
            total += L.lib().dsf_conv_x6_image_granules(I(KH), I(KW), I(Ck), I(Cn))
    if not rows:
        return
    slot = _JOBS.get(id(owner))
    if slot is None or slot[0] != rows:
        dev = entries[0][0].device
        table = torch.tensor(rows + [(0, 0, 0, 0, 0, 0, 0, total)], dtype=torch.int64).to(dev)
        slot = _JOBS[id(owner)] = (rows, table)
    from ._lib import I64
    check(L.lib().dsf_conv_x6_split_weights_multi(ptr(slot[1]), I(len(rows)), I64(total), stream_ptr()),
          "dsf_conv_x6_split_weights_multi")
    for p, mode in entries:
        cache = p.__dict__["_dsf_x6"]
        _, img, geom = cache[mode]
        cache[mode] = ((p._version, _EPOCH, p.__dict__.get("_dsf_wepoch", 0), p.data_ptr()), img, geom)


class StatsRequest:
    """Side channel between a convolution and the BatchNorm that follows it (nn_norm.conv_bn_act): while ``STATS`` holds a
    request, a bias-free conv_x6 forward also writes the per-tile partial rows of the BatchNorm batch statistics
    (dsf_conv_x6_forward_bn) and leaves them here; ``rows`` stays 0 when the launch it chose cannot (split K, ...)."""
    part, rows = None, 0
    acc, filled = None, 0      # acc: a zeroed block of accumulation rows (nn_norm.stat_pool) -> the epilogue ADDS its tile sums there


class AffineRequest:
    """The same side channel for the evaluation-mode epilogue: y = act((conv + bias) * scale + shift (+ residual)) in the
    convolution's own launch (dsf_conv_x6_forward_affine); ``applied`` tells nn_norm.conv_bn_act whether it happened."""

    def __init__(self, scale, shift, residual, relu):
        self.scale, self.shift, self.residual, self.relu, self.applied = scale, shift, residual, bool(relu), False


STATS = None
C1_STATS = [os.environ.get("DSF_C1_STATS", "1") == "1"]     # the 1-channel stem kernels serve a StatsRequest too (0: the BatchNorm reduces itself)


def _fwd_x6(x, image, bias, out_hw, Co, KH, KW, stride, pad, dil=1):
    """_fwd with the weight operand given as a conv_x6 image whose reduction width is x's channel count (dil 1, or the
    dil-2 transposed-convolution gather with even output sizes)."""
    B, Ci, Hi, Wi = x.shape
    Ho, Wo = out_hw
    if RECORD is not None:
        RECORD.append(("x6", B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, dil, pad[0], pad[1]))
    y = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=CL)
    req = STATS
    if isinstance(req, AffineRequest) and B > 0:
        import ctypes
        done = ctypes.c_int(0)
        res = req.residual
        check(L.lib().dsf_conv_x6_forward_affine(ptr_nhwc(x), ptr(image), ptr(bias), ptr_nhwc(y), I(B), I(Hi), I(Wi), I(Ci), I(Ho),
                                                 I(Wo), I(Co), I(KH), I(KW), I(stride), I(dil), I(pad[0]), I(pad[1]), ptr(req.scale),
                                                 ptr(req.shift), ptr_nhwc(res) if res is not None else None, I(int(req.relu)),
                                                 ctypes.byref(done), stream_ptr()), "dsf_conv_x6_forward_affine")
        req.applied = bool(done.value)
        return y
    if isinstance(req, StatsRequest) and req.acc is not None and bias is None and B > 0:
        import ctypes
        filled = ctypes.c_int(0)
        check(L.lib().dsf_conv_x6_forward_bn_acc(ptr_nhwc(x), ptr(image), ptr_nhwc(y), I(B), I(Hi), I(Wi), I(Ci), I(Ho), I(Wo), I(Co),
                                                 I(KH), I(KW), I(stride), I(dil), I(pad[0]), I(pad[1]), ptr(req.acc),
                                                 I(int(L.lib().dsf_bn_acc_rows())), ctypes.byref(filled), stream_ptr()),
              "dsf_conv_x6_forward_bn_acc")
        req.filled = filled.value
        return y
    if isinstance(req, StatsRequest) and req.acc is None and bias is None and B > 0:
        import ctypes
        rows_max = int(L.lib().dsf_conv_x6_bn_stats_rows(I(B), I(Ho), I(Wo)))
        part = torch.empty(rows_max * 2 * Co, device=x.device, dtype=torch.float32)
        rows = ctypes.c_int(0)
        check(L.lib().dsf_conv_x6_forward_bn(ptr_nhwc(x), ptr(image), ptr_nhwc(y), I(B), I(Hi), I(Wi), I(Ci), I(Ho), I(Wo), I(Co),
                                             I(KH), I(KW), I(stride), I(dil), I(pad[0]), I(pad[1]), ptr(part), ctypes.byref(rows),
                                             stream_ptr()), "dsf_conv_x6_forward_bn")
        if rows.value > 0:
            req.part, req.rows = part, rows.value
        return y
    if _ZERO is not None and B > 0 and y.numel() * 4 <= ZERO_POOL_MAX_BYTES:
        # a small layer that the launcher splits along K (partial sums meet in Y by float atomics): Y from the step's pooled zero
        # fill instead of a fill launch of its own in front of every such layer
        import ctypes
        k = ctypes.c_int(1)
        check(L.lib().dsf_conv_x6_forward_plan(I(B), I(Hi), I(Wi), I(Ci), I(Ho), I(Wo), I(Co), I(KH), I(KW), I(stride), I(dil), I(pad[0]),
                                               I(pad[1]), None, ctypes.byref(k)), "dsf_conv_x6_forward_plan")
        splits = k.value
        off = _zero_take(B * Ho * Wo * Co, x.device) if splits > 1 else None
        if off is not None:
            # (a tensor of its own on the pool's storage, not a view of the pool: views share ONE version counter, and an
            #  in-place op on any pooled output would then invalidate every other one that autograd has saved)
            y = torch.empty(0, device=x.device, dtype=torch.float32).set_(_ZERO[0].untyped_storage(), off, (B, Co, Ho, Wo),
                                                                          (Ho * Wo * Co, 1, Wo * Co, Co))
            check(L.lib().dsf_conv_x6_forward_into(ptr_nhwc(x), ptr(image), ptr(bias), ptr_nhwc(y), I(B), I(Hi), I(Wi), I(Ci), I(Ho),
                                                   I(Wo), I(Co), I(KH), I(KW), I(stride), I(dil), I(pad[0]), I(pad[1]), I(splits),
                                                   stream_ptr()), "dsf_conv_x6_forward_into")
            return y
    check(L.lib().dsf_conv_x6_forward(ptr_nhwc(x), ptr(image), ptr(bias), ptr_nhwc(y), I(B), I(Hi), I(Wi), I(Ci), I(Ho), I(Wo),
                                      I(Co), I(KH), I(KW), I(stride), I(dil), I(pad[0]), I(pad[1]), I(0), stream_ptr()),
          "dsf_conv_x6_forward")
    return y


# ---- zero pool: one fill per step for the outputs of the small split-K layers -----------------------------------------
# ``with zero_pool(owner, device):`` around one forward + backward of a step.  The pool is as large as the demand the PREVIOUS
# pass under the same owner recorded (the first pass records only and every layer fills its own Y); outputs above
# ZERO_POOL_MAX_BYTES keep their own fill (a pooled fill of the 16 x 16 maps' outputs costs what it saves).  The slices are
# views of one tensor: it lives as long as any of them (activations saved for the backward pass included).
ZERO_POOL = [os.environ.get("DSF_ZERO_POOL", "1") == "1"]
ZERO_POOL_MAX_BYTES = int(os.environ.get("DSF_ZERO_POOL_MAX_MB", "4")) << 20
_ZERO = None         # [flat zeroed tensor or None, next offset, demand of this pass]


class zero_pool:
    def __init__(self, owner, device):
        self.owner, self.device = owner, torch.device(device)

    def __enter__(self):
        global _ZERO
        self.saved = _ZERO
        if ZERO_POOL[0] and self.device.type == "cuda" and not L.deterministic():
            n = int(self.owner.__dict__.get("_zero_pool_floats", 0))
            _ZERO = [torch.zeros(n, device=self.device, dtype=torch.float32) if n > 0 else None, 0, 0]
        else:
            _ZERO = None
        return self

    def __exit__(self, *a):
        global _ZERO
        if _ZERO is not None:
            self.owner.__dict__["_zero_pool_floats"] = _ZERO[2]
        _ZERO = self.saved


def _zero_take(n, device):
    n = (n + 63) & ~63                                    # 256-byte slices: rows of float atomics stay inside their cache lines
    _ZERO[2] += n
    buf, off = _ZERO[0], _ZERO[1]
    if buf is None or buf.device != device or off + n > buf.numel():
        return None
    _ZERO[1] = off + n
    return off


def _x6_dil_ok(Ck, Ho, Wo, dil, n_in=0):
    """transposed-convolution gathers conv_x6 implements: dil 1, or dil 2 with even output sizes"""
    return _x6_ok(Ck, n_in) and (dil == 1 or (dil == 2 and Ho % 2 == 0 and Wo % 2 == 0))


def _c1_ok(Ci, Co, KH, KW, stride, pad):
    """1-channel direct kernels (conv_c1.hip): the stems"""
    return Ci == 1 and pad[0] == pad[1] and bool(L.lib().dsf_conv_c1_supported(I(Co), I(KH), I(KW), I(stride)))


def _co1_ok(Ci, Co, KH, KW, stride, pad):
    """one output channel (conv_c1.hip: conv_co1_fwd_kernel): the generator's last layer"""
    return Co == 1 and KH == KW and KH in (3, 5, 7) and stride == 1 and Ci % 8 == 0 and pad[0] == pad[1]


def _fwd_co1(x, wk, bias, out_hw, K, pad):
    B, Ci, Hi, Wi = x.shape
    Ho, Wo = out_hw
    if RECORD is not None:
        RECORD.append(("co1_fwd", B, Hi, Wi, Ci, Ho, Wo, 1, K, K, 1, 1, pad, pad))
    y = torch.empty((B, 1, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=CL)
    check(L.lib().dsf_conv_co1_forward(ptr_nhwc(x), ptr(wk), ptr(bias), ptr_nhwc(y), I(B), I(Hi), I(Wi), I(Ci), I(Ho), I(Wo), I(K), I(pad),
                                       stream_ptr()), "dsf_conv_co1_forward")
    return y


def _fwd_c1(x, wk, bias, out_hw, Co, K, stride, pad):
    B, _, Hi, Wi = x.shape
    Ho, Wo = out_hw
    if RECORD is not None:
        RECORD.append(("c1_fwd", B, Hi, Wi, 1, Ho, Wo, Co, K, K, stride, 1, pad, pad))
    y = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=CL)
    req = STATS
    if isinstance(req, StatsRequest) and req.acc is not None and bias is None and B > 0 and C1_STATS[0] and not L.deterministic():
        # the BatchNorm behind the stem asked for its batch statistics: lane = channel, the sums ride in two registers
        check(L.lib().dsf_conv_c1_forward_bn_acc(ptr_nhwc(x), ptr(wk), None, ptr_nhwc(y), I(B), I(Hi), I(Wi), I(Ho), I(Wo), I(Co), I(K), I(stride),
                                                 I(pad), ptr(req.acc), I(int(L.lib().dsf_bn_acc_rows())), stream_ptr()),
              "dsf_conv_c1_forward_bn_acc")
        req.filled = 1
        return y
    check(L.lib().dsf_conv_c1_forward(ptr_nhwc(x), ptr(wk), ptr(bias), ptr_nhwc(y), I(B), I(Hi), I(Wi), I(Ho), I(Wo), I(Co), I(K),
                                      I(stride), I(pad), stream_ptr()), "dsf_conv_c1_forward")
    return y


def _wrw_c1(x, gy, K, stride, pad):
    B, _, Hi, Wi = x.shape
    _, Co, Ho, Wo = gy.shape
    if RECORD is not None:
        RECORD.append(("c1_wrw", B, Hi, Wi, 1, Ho, Wo, Co, K, K, stride, 1, pad, pad))
    dw = _pool_take(K * K * Co, x.device)
    pooled = dw is not None
    dw = dw.view(K, K, 1, Co) if pooled else torch.empty((K, K, 1, Co), device=x.device, dtype=torch.float32)
    ws = torch.empty(L.lib().dsf_conv_c1_workspace_bytes(I(K), I(K)) // 4, device=x.device, dtype=torch.float32)
    check(L.lib().dsf_conv_c1_wrw(ptr_nhwc(x), ptr_nhwc(gy), ptr(dw), ptr(ws), I(B), I(Hi), I(Wi), I(Ho), I(Wo), I(Co), I(K),
                                  I(stride), I(pad), I(1 if pooled else 0), stream_ptr()), "dsf_conv_c1_wrw")
    return dw


def _wrw(x, gy, KH, KW, stride, pad, out=None, dbias=None, param=None):
    """-> dW [KH][KW][Ci][Co] for x (B,Ci,Hi,Wi), gy (B,Co,Ho,Wo), both channels_last; ``out``: add into this dW instead.
    ``param``: the weight this dW is the gradient of (kernel layout): under data parallelism dW is written into the weight's slot
    of its all-reduce bucket (parallel.GradAllReducer.grad_slot) instead of the step's gradient pool.
    ``dbias``: a one-element list holding None -- when the split kernels run the launch (not in deterministic mode) it receives the
    bias gradient (Co floats from the zeroed gradient pool or a fresh zero vector), computed by the same launch."""
    B, Ci, Hi, Wi = x.shape
    _, Co, Ho, Wo = gy.shape
    if RECORD is not None:
        RECORD.append(("wrw", B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, 1, pad[0], pad[1]))
    dw = _grad_out(KH * KW * Ci * Co, x.device, param) if out is None else out
    pooled = dw is not None
    dw = dw.view(KH, KW, Ci, Co) if pooled else torch.empty((KH, KW, Ci, Co), device=x.device, dtype=torch.float32)
    # both operands are activations: conv_x6 splits them on the fly (Ci % 4 == 0 and Co % 4 == 0), else the fp32 MFMA kernel
    if _wrw_x6_ok(Ci, Co, x.numel(), gy.numel()):
        nws = int(L.lib().dsf_conv_x6_wrw_workspace_bytes(I(B), I(Ho), I(Wo), I(Ci), I(Co), I(KH), I(KW)))    # > 0: deterministic mode
        ws = torch.empty(nws // 4, device=x.device, dtype=torch.float32) if nws else None
        if dbias is not None and not nws:
            save = _POOL[1] if _POOL is not None else None
            db = _pool_take(Co, x.device)
            db = db if db is not None else torch.zeros(Co, device=x.device, dtype=torch.float32)
            rc = L.lib().dsf_conv_x6_wrw_bias(ptr_nhwc(x), ptr_nhwc(gy), ptr(dw), ptr(db), I(B), I(Hi), I(Wi), I(Ci), I(Ho), I(Wo), I(Co),
                                              I(KH), I(KW), I(stride), I(pad[0]), I(pad[1]), I(1 if pooled else 0), stream_ptr())
            if rc == 0:
                dbias[0] = db
                return dw
            if rc != L.ERR_UNSUPPORTED:
                check(rc, "dsf_conv_x6_wrw_bias")
            if save is not None:                        # declined (a layer with many pixel splits): nothing was launched
                _POOL[1] = save
        check(L.lib().dsf_conv_x6_wrw_ws(ptr_nhwc(x), ptr_nhwc(gy), ptr(dw), I(B), I(Hi), I(Wi), I(Ci), I(Ho), I(Wo), I(Co), I(KH),
                                         I(KW), I(stride), I(pad[0]), I(pad[1]), I(1 if pooled else 0), ptr(ws), stream_ptr()),
              "dsf_conv_x6_wrw_ws")
        return dw
    check(L.lib().dsf_conv_igemm_wrw(ptr_nhwc(x), ptr_nhwc(gy), ptr(dw), I(B), I(Hi), I(Wi), I(Ci), I(Ho), I(Wo), I(Co), I(KH),
                                     I(KW), I(stride), I(pad[0]), I(pad[1]), I(1 if pooled else 0), stream_ptr()), "dsf_conv_igemm_wrw")
    return dw


# ---- weight-gradient pool: one zero-fill per step instead of one memset per layer ------------------------------------
_POOL = None         # [flat zeroed tensor, next offset]


class grad_pool:
    """``with grad_pool(n_floats, device):`` around a backward pass: the backward-weights kernels (which accumulate their
    pixel splits with float atomics and therefore need zeroed outputs) take their outputs from ONE pre-zeroed buffer,
    256-byte aligned slices handed out in call order, instead of zeroing ~50 separate tensors.  The slices stay valid as
    long as the gradients that view them live; a pool that runs out falls back to per-layer buffers.
    ``reducer``: the parallel.GradAllReducer of the parameters this pass differentiates -- its bucket store is zeroed here, on the
    stream the pass starts on, and the weight gradients are then written straight into their bucket slots (only THIS reducer: the
    gradients another network's reducer still holds for its optimizer are not touched)."""

    def __init__(self, n_floats, device, reducer=None):
        self.n, self.device, self.reducer = int(n_floats), device, reducer

    def __enter__(self):
        global _POOL
        self.saved = _POOL
        _POOL = [torch.zeros(self.n, device=self.device, dtype=torch.float32), 0] if self.n > 0 else None
        if self.reducer is not None and hasattr(self.reducer, "begin_backward"):
            self.reducer.begin_backward()
        return self

    def __exit__(self, *a):
        global _POOL
        _POOL = self.saved


def _pow2_at_least(v, floor):
    v = max(int(v), floor)
    return 1 << (v - 1).bit_length()


# slice alignment of the gradient pool in floats (DSF_POOL_ALIGN: tuning aid, read once; rounded up to a power of two >= 4 -- it
# is used as a mask, and any other value would make slices overlap)
_POOL_ALIGN = _pow2_at_least(os.environ.get("DSF_POOL_ALIGN", "64") or 64, 4)


def _grad_out(n, device, param=None):
    """zeroed output (n floats) of a weight-gradient launch: the parameter's slot in its data-parallel bucket -- the gradient is
    then written where the all-reduce reads it, and autograd adopts that view as ``.grad`` --, else a slice of the step's gradient
    pool, else None (the caller allocates and the kernel overwrites)"""
    if param is not None:
        owner = param.__dict__.get("_dsf_grad_slot")
        if owner is not None and param.dim() == 4 and param.permute(2, 3, 1, 0).is_contiguous():    # memory order [KH][KW][.][.]
            v = owner.grad_slot(param)
            if v is not None and v.device == device and v.numel() == n:
                return v
    return _pool_take(n, device)


def _pool_take(n, device):
    if _POOL is None or _POOL[0].device != device:
        return None
    off = _POOL[1]
    a = _POOL_ALIGN - 1                                  # 256-byte slices: a 32-lane row of float atomics never straddles a cache line
    end = off + ((n + a) & ~a)
    if end > _POOL[0].numel():
        return None
    _POOL[1] = end
    return _POOL[0][off:off + n]


def weight_grad_floats(module):
    """pool size for one backward pass over ``module``: the weights of its dsf_amd convolution layers"""
    return sum(((m.weight.numel() + 63) & ~63) + (((m.bias.numel() + 63) & ~63) if m.bias is not None else 0)      # (+ bias gradients: _wrw)
               for m in module.modules() if isinstance(m, (Conv2d, ConvTranspose2d)))


def _bias_grad(gy):
    """sum over (B,H,W) of an NHWC tensor: one column-sum kernel (torch's strided reduce and a skinny hipBLASLt
    GEMM both need ~220-250 us for a 33 MB channels_last tensor)."""
    B, Co, H, W = gy.shape
    out = torch.empty(Co, device=gy.device, dtype=torch.float32)
    from ._lib import I64
    ws = torch.empty(256 * Co, device=gy.device, dtype=torch.float32)            # dsf_col_sum_workspace_bytes(Co)
    check(L.lib().dsf_col_sum(ptr_nhwc(gy), I64(B * H * W), I(Co), ptr(out), ptr(ws), stream_ptr()), "dsf_col_sum")
    return out


def ptr_nhwc(t):
    if not t.is_cuda:
        raise RuntimeError("dsf_amd convolution runs on the GPU only (got a %s tensor); there is no CPU path" % t.device)
    assert t.is_contiguous(memory_format=CL) or t.is_contiguous()
    import ctypes
    if t.numel() == 0:                                   # empty batch: a valid (never dereferenced) address instead of NULL
        return ctypes.c_void_p(L._dummy(t.device).data_ptr())
    return ctypes.c_void_p(t.data_ptr())


def kernel_layout_(param, perm):
    """Re-strides ``param`` in place so that its MEMORY order is ``param.permute(perm)`` (the kernels' GEMM operand)
    while the logical shape -- state-dict compatibility -- is unchanged.  With it the forward operand is a free view,
    the backward-weights output is already in the parameter's layout (autograd adopts it without a copy) and
    optimizer state created with preserve_format matches."""
    inv = [perm.index(i) for i in range(len(perm))]
    param.data = param.data.permute(*perm).contiguous().permute(*inv)


def _wt_ok(Ci, Co, Ho, Wo, dil):
    """shapes the transposed-weight kernels implement (dsf_conv_igemm_forward_wt)"""
    return Ci >= 32 and Ci % 4 == 0 and Co % 4 == 0 and (dil == 1 or (dil == 2 and Ho % 2 == 0 and Wo % 2 == 0))


def _fwd_wt(x, wt, bias, out_hw, Co, KH, KW, stride, dil, pad):
    """_fwd with the weight operand transposed and tap-flipped, wt [KH][KW][Co][Ci] (a layer's own parameter memory
    seen from its other direction): no re-laid copy of the weights."""
    B, Ci, Hi, Wi = x.shape
    Ho, Wo = out_hw
    if RECORD is not None:
        RECORD.append(("fwd_wt", B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, dil, pad[0], pad[1]))
    y = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=CL)
    check(L.lib().dsf_conv_igemm_forward_wt(ptr_nhwc(x), ptr(wt), ptr(bias), ptr_nhwc(y), I(B), I(Hi), I(Wi), I(Ci), I(Ho),
                                            I(Wo), I(Co), I(KH), I(KW), I(stride), I(dil), I(pad[0]), I(pad[1]),
                                            stream_ptr()), "dsf_conv_igemm_forward_wt")
    return y


def _bwd_data_s1(gy, wk, out_hw, Cin, KH, KW, pad):
    """stride-1 backward-data straight from the forward operand wk [KH][KW][Cin][Cout] (no flipped copy)."""
    B, Cout, Ho, Wo = gy.shape
    H, W = out_hw
    if RECORD is not None:
        RECORD.append(("bwd_s1", B, Ho, Wo, Cout, H, W, Cin, KH, KW, 1, 1, pad[0], pad[1]))
    gx = torch.empty((B, Cin, H, W), device=gy.device, dtype=torch.float32, memory_format=CL)
    check(L.lib().dsf_conv_igemm_bwd_data_s1(ptr_nhwc(gy), ptr(wk), ptr_nhwc(gx), I(B), I(H), I(W), I(Cout), I(Cin), I(KH), I(KW),
                                             I(pad[0]), I(pad[1]), stream_ptr()), "dsf_conv_igemm_bwd_data_s1")
    return gx


class Conv2dFunction(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding):
        x = _nhwc(x)
        Co, Ci, KH, KW = weight.shape
        B, _, Hi, Wi = x.shape
        Ho = (Hi + 2 * padding[0] - KH) // stride + 1
        Wo = (Wi + 2 * padding[1] - KW) // stride + 1
        wk = weight.detach().float().permute(2, 3, 1, 0).contiguous()          # a free view when the weight has kernel layout
        b = bias.detach().float().contiguous() if bias is not None else None
        if _c1_ok(Ci, Co, KH, KW, stride, padding):
            y = _fwd_c1(x, wk, b, (Ho, Wo), Co, KH, stride, padding[0])
        elif _co1_ok(Ci, Co, KH, KW, stride, padding):
            y = _fwd_co1(x, wk, b, (Ho, Wo), KH, padding[0])
        elif _x6_ok(Ci, x.numel()):
            y = _fwd_x6(x, _x6_image(weight, wk, 0), b, (Ho, Wo), Co, KH, KW, stride, padding)
        else:
            y = _fwd(x, wk, b, (Ho, Wo), Co, KH, KW, stride, 1, padding)
        ctx.save_for_backward(x, weight, wk)
        ctx.cfg = (stride, padding, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, wk = ctx.saved_tensors
        stride, padding, has_bias = ctx.cfg
        Co, Ci, KH, KW = weight.shape
        if torch.is_grad_enabled():
            # backward of the backward is wanted (create_graph=True: WGAN-GP's gradient penalty, reference
            # render_model/transfer.py:356-391): express both gradients through differentiable Functions
            gx = gw = gb = None
            if ctx.needs_input_grad[0]:
                op = (x.shape[2] - ((gy.shape[2] - 1) * stride - 2 * padding[0] + KH), x.shape[3] - ((gy.shape[3] - 1) * stride - 2 * padding[1] + KW))
                gx = ConvTranspose2dFunction.apply(gy, weight, None, stride, padding, op)     # (Co -> Ci): the conv's weight IS its (in, out, kh, kw)
            if ctx.needs_input_grad[1]:
                join_side_streams()                      # a main-stream contribution: see _wrw_dispatch
                if SIDE_API:
                    weight.__dict__["_dsf_pass"] = (torch._C._current_graph_task_id(), None)
                gw = _WeightGradFunction.apply(x, gy, (KH, KW), stride, padding)
            if has_bias and ctx.needs_input_grad[2]:
                gb = gy.sum((0, 2, 3))
            return gx, gw, gb, None, None
        with torch.no_grad():
            return Conv2dFunction._backward_fast(ctx, gy)

    @staticmethod
    def _backward_fast(ctx, gy):
        x, weight, wk = ctx.saved_tensors
        stride, padding, has_bias = ctx.cfg
        Co, Ci, KH, KW = weight.shape
        gy = _nhwc(gy)
        gx = gw = gb = None
        if ctx.needs_input_grad[0] and stride == 1 and _x6_ok(Co, gy.numel()):
            gx = _fwd_x6(gy, _x6_image(weight, wk, 1), None, (x.shape[2], x.shape[3]), Ci, KH, KW, 1,
                         (KH - 1 - padding[0], KW - 1 - padding[1]))
        elif ctx.needs_input_grad[0] and stride == 1 and Co >= 32 and Co % 4 == 0 and Ci % 4 == 0:
            gx = _bwd_data_s1(gy, wk, (x.shape[2], x.shape[3]), Ci, KH, KW, padding)
        elif ctx.needs_input_grad[0] and _x6_dil_ok(Co, x.shape[2], x.shape[3], stride, gy.numel()):
            # backward-data of a strided convolution = transposed-convolution gather over gy with the mode-1 image
            gx = _fwd_x6(gy, _x6_image(weight, wk, 1), None, (x.shape[2], x.shape[3]), Ci, KH, KW, 1,
                         (KH - 1 - padding[0], KW - 1 - padding[1]), dil=stride)
        elif ctx.needs_input_grad[0] and _wt_ok(Co, Ci, x.shape[2], x.shape[3], stride):
            # backward-data as a dilated convolution over gy; wk (the forward operand) is its transposed, flipped weight
            gx = _fwd_wt(gy, wk, None, (x.shape[2], x.shape[3]), Ci, KH, KW, 1, stride, (KH - 1 - padding[0], KW - 1 - padding[1]))
        elif ctx.needs_input_grad[0]:
            wd = weight.detach().float().flip(2, 3).permute(2, 3, 0, 1).contiguous()          # [kh'][kw'][Co][Ci]
            gx = _fwd(gy, wd, None, (x.shape[2], x.shape[3]), Ci, KH, KW, 1, stride, (KH - 1 - padding[0], KW - 1 - padding[1]))
        if ctx.needs_input_grad[1]:
            if _c1_ok(Ci, Co, KH, KW, stride, padding):  # (the stem kernel overwrites its output: never a shared dW)
                dw = _wrw_dispatch(weight, lambda: _wrw_c1(x, gy, KH, stride, padding[0]), None, (x, gy))
            else:
                # the bias gradient rides in the weight-gradient launch when that launch runs on THIS stream (small layers and
                # everything under DSF_WRW_STREAM=0): on the side stream nothing orders the bias's AccumulateGrad behind it
                on = BIAS_IN_WRW[0] if BIAS_IN_WRW[0] is not None else os.environ.get("DSF_BIAS_IN_WRW", "1") == "1"
                want_db = has_bias and ctx.needs_input_grad[2] and on
                main = stream_ptr().value if want_db else None      # (raw handles: a torch Stream object costs ~9 us to build)
                cell = [None]
                dw = _wrw_dispatch(weight, lambda: _wrw(x, gy, KH, KW, stride, padding,
                                                        dbias=cell if (want_db and stream_ptr().value == main) else None, param=weight),
                                   lambda acc: _wrw(x, gy, KH, KW, stride, padding, out=acc), (x, gy),
                                   work=2.0 * gy.numel() * Ci * KH * KW)
                gb = cell[0]
            gw = None if dw is False else dw.permute(3, 2, 0, 1)
        if has_bias and ctx.needs_input_grad[2] and gb is None:
            gb = _bias_grad(gy)
        return gx, gw, gb, None, None


# ------------------------------------------------------------------------------------------------
# Backward-weights beside the backward-data / BatchNorm chain.  dW of a layer feeds nothing in the backward pass: it is
# launched on a second stream (forked from the current one, so its inputs are ready) while the chain continues, and the
# streams meet when the backward pass ends (an autograd-engine callback, as DistributedDataParallel uses) -- before an
# optimizer, a test or anything else can look at a gradient.  19.9 vs 20.3 ms per step.
# Only where nothing can read the gradient earlier (_side_ok): the weight is a leaf parameter (not, e.g., the merged head
# weight, whose gradient autograd splits right away) in the kernels' memory layout (so that AccumulateGrad adopts dW
# instead of copying it on the main stream), it has no gradient yet (else AccumulateGrad adds into it), and no tensor /
# post-accumulate hooks other than GradAllReducer's, which joins the streams itself before it packs a bucket.
# A weight that receives SEVERAL contributions in one backward pass (a network applied to two batches, a discriminator
# on real and fake images, a retained graph walked together with a new one, the double-backward graph of a gradient
# penalty) would have them ADDED by the autograd engine on the main stream as they arrive.  That is decided per backward
# pass, not from a count taken in the forward (which a retained graph can falsify): the FIRST contribution of a pass is
# recorded on the weight (graph task id, dW); every later one of the same pass adds into that same buffer on the side
# stream and hands autograd nothing, or -- deterministic mode, the stem kernel, a first contribution that ran on the main
# stream -- joins the streams and runs on the main stream, after which the rest of the pass stays there.
# Limitation: a weight that ALSO feeds a non-convolution op of the same graph (a weight penalty written as w.pow(2).sum())
# gets that op's gradient added by the engine, which this module cannot see: mark such weights with
# ``no_side_stream(params)`` or set DSF_WRW_STREAM=0.
# DSF_WRW_STREAM=0 keeps everything on one stream.  In a multi-rank process group only weights managed by GradAllReducer take
# the side stream (torch's own DistributedDataParallel hooks the gradient accumulator nodes, which a tensor cannot report).
# The feature rests on private torch interfaces (the graph task id, the engine's end-of-pass callback, the tensor's hook
# dictionaries): _side_api_ok() probes them once and everything stays on one stream when one is missing.
# ------------------------------------------------------------------------------------------------
def _side_api_ok():
    try:
        t = torch.zeros(1)
        return (callable(getattr(torch._C, "_current_graph_task_id", None)) and torch._C._current_graph_task_id() == -1 and
                callable(getattr(torch.autograd.Variable._execution_engine, "queue_callback", None)) and
                hasattr(t, "_backward_hooks") and hasattr(t, "_post_accumulate_grad_hooks"))
    except Exception:
        return False


SIDE_API = _side_api_ok()
WRW_STREAM = [os.environ.get("DSF_WRW_STREAM", "1") == "1" and SIDE_API]
WRW_MIN_WORK = [float(os.environ.get("DSF_WRW_MIN_GFLOP", "8")) * 1e9]   # below: the layer's dW stays on the chain's stream
BIAS_IN_WRW = [None]     # bias gradient from the weight-gradient launch (dsf_conv_x6_wrw_bias): None = DSF_BIAS_IN_WRW (default on), read per call
WRW_PRIORITY = int(os.environ.get("DSF_WRW_PRIORITY", "0"))          # priority of the side stream (lower number = higher priority)
_SIDE = {}
_JOIN_QUEUED = [-1]
_PENDING = [False]
_HELD = []


def no_side_stream(params, flag=True):
    """Keeps the weight gradients of ``params`` on the main stream (see the limitation above)."""
    for p in params:
        p.__dict__["_dsf_no_side"] = bool(flag)


def _wrw_dispatch(weight, fn_new, fn_add, held, work=None):
    """One node's weight-gradient launch -> dW (handed to autograd) or False (added into the dW an earlier node of this
    backward pass handed over; autograd gets None).  ``fn_new()`` -> dW;  ``fn_add(acc)`` adds into acc (None: this kernel cannot)."""
    task = torch._C._current_graph_task_id() if SIDE_API else -1
    rec = weight.__dict__.get("_dsf_pass")
    first = task < 0 or rec is None or rec[0] != task
    if first and task >= 0 and work is not None and work < WRW_MIN_WORK[0]:
        # a small layer (< DSF_WRW_MIN_GFLOP, default 8): the fork and the join cost more than its launch hides.  Config 3 under
        # the HIP-graph replay, same box: 17.9 ms per step with all 96 layers on the side stream, 17.0 with none, 16.3-17.3 with
        # the threshold at 8, 16.7-16.8 at 32; config 2: 18.8 ms at 0, 2 and 8, 19.6 at 32.  First contribution of the pass, so
        # nothing of this weight is pending on the side stream: no join either
        weight.__dict__["_dsf_pass"] = (task, None)
        return fn_new()
    if _side_ok(weight) and task >= 0:
        if first:
            dw = _on_side_stream(fn_new, held)
            weight.__dict__["_dsf_pass"] = (task, dw)
            return dw
        if rec[1] is not None and fn_add is not None and not L.deterministic() and MATH == "x6":
            acc = rec[1]
            _on_side_stream(lambda: fn_add(acc), held)
            return False
    # main stream: whatever reads this gradient next (the engine's add of two contributions, AccumulateGrad's add into an
    # existing .grad) may also read a pending one of the side stream
    join_side_streams()
    if task >= 0:
        weight.__dict__["_dsf_pass"] = (task, None)    # the rest of this pass stays on the main stream for this weight
    return fn_new()


def main_stream_weight_grad(weight):
    """Bookkeeping of a contribution to ``weight``'s gradient that the CALLER launches on the current stream (nn_norm._StemFunction):
    _wrw_dispatch's main-stream branch without the launch -- a later contribution of this backward pass stays on the main stream too,
    and one that is still running on the side stream is joined first."""
    task = torch._C._current_graph_task_id() if SIDE_API else -1
    rec = weight.__dict__.get("_dsf_pass")
    if task >= 0:
        if rec is not None and rec[0] == task:
            join_side_streams()
        weight.__dict__["_dsf_pass"] = (task, None)


def _side_ok(weight):
    if not (WRW_STREAM[0] and weight.is_leaf and weight.grad is None and not weight._backward_hooks):
        return False
    if weight.__dict__.get("_dsf_no_side") or weight.dtype != torch.float32 or weight.dim() != 4:
        return False
    # AccumulateGrad adopts dW only when it has the parameter's strides (else it clones it -- on the main stream, while the
    # side stream still writes): dW comes in the kernels' memory order [KH][KW][.][.]
    if not weight.permute(2, 3, 1, 0).is_contiguous():
        return False
    ours = bool(weight.__dict__.get("_dsf_hooks_join"))
    if getattr(weight, "_post_accumulate_grad_hooks", None) and not ours:
        return False
    # a multi-rank process group without GradAllReducer on this weight: some other data-parallel wrapper (torch's
    # DistributedDataParallel hooks the gradient-accumulator nodes, which a tensor cannot report) may read the gradient as
    # soon as it is accumulated -- stay on one stream
    if not ours and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        return False
    return True


def _on_side_stream(fn, inputs):
    # (ONE weight-gradient stream: taking turns over two or three of them, so that one launch's tail overlaps the next one's
    #  start, measured 20.17 / 20.02 ms per step against 19.83-19.91 with one -- the launches then compete with each other)
    cur = torch.cuda.current_stream()
    side = _SIDE.get(cur.device)
    if side is None:
        side = _SIDE[cur.device] = torch.cuda.Stream(device=cur.device, priority=WRW_PRIORITY)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        out = fn()
    # the side stream still reads `inputs`: they are kept alive until its launch has finished (polled, oldest first) or the streams
    # have met again.  Not Tensor.record_stream: its event-deferred frees made the caching allocator fall back to hipMalloc on
    # the big configurations (config 4: 193 -> 276 ms per step)
    if torch.cuda.is_current_stream_capturing():         # (no event queries inside a graph capture: held until the join)
        _HELD.append((None, inputs, out))
    else:
        ev = torch.cuda.Event()
        ev.record(side)
        _HELD.append((ev, inputs, out))
        while len(_HELD) > 1 and _HELD[0][0] is not None and _HELD[0][0].query():    # launches the side stream has finished:
            _HELD.pop(0)                                                             # nothing reads their inputs any more
    _PENDING[0] = True
    task = torch._C._current_graph_task_id()             # one join callback per backward pass (ids are never reused, so a pass
    if _JOIN_QUEUED[0] != task:                           # that died with an exception cannot leave a stale "already queued")
        _JOIN_QUEUED[0] = task
        torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
    return out


def join_side_streams():
    """Orders the current stream(s) behind every backward-weights launch still running on the side stream."""
    if _PENDING[0]:
        _PENDING[0] = False
        for dev, side in _SIDE.items():
            torch.cuda.current_stream(dev).wait_stream(side)
        del _HELD[:]


class _WeightGradFunction(Function):
    """dW (Co,Ci,KH,KW) of a convolution as a differentiable function of (x, gy): used only under create_graph=True.
    Its own backward is two ordinary convolutions:  d/dx = conv_transpose(gy, ggW),  d/dgy = conv(x, ggW)."""

    @staticmethod
    def forward(ctx, x, gy, ksize, stride, padding):
        x, gy = _nhwc(x), _nhwc(gy)
        KH, KW = ksize
        Ci = x.shape[1]
        Co = gy.shape[1]
        if _c1_ok(Ci, Co, KH, KW, stride, padding):
            gw = _wrw_c1(x, gy, KH, stride, padding[0]).permute(3, 2, 0, 1)
        else:
            gw = _wrw(x, gy, KH, KW, stride, padding).permute(3, 2, 0, 1)
        ctx.save_for_backward(x, gy)
        ctx.cfg = (ksize, stride, padding)
        return gw

    @staticmethod
    @once_differentiable
    def backward(ctx, ggw):
        x, gy = ctx.saved_tensors
        (KH, KW), stride, padding = ctx.cfg
        gx = ggy = None
        ggw = ggw.contiguous()
        if ctx.needs_input_grad[0]:
            op = (x.shape[2] - ((gy.shape[2] - 1) * stride - 2 * padding[0] + KH), x.shape[3] - ((gy.shape[3] - 1) * stride - 2 * padding[1] + KW))
            gx = ConvTranspose2dFunction.apply(gy, ggw, None, stride, padding, op)
        if ctx.needs_input_grad[1]:
            ggy = Conv2dFunction.apply(x, ggw, None, stride, padding)
        return gx, ggy, None, None, None


class ConvTranspose2dFunction(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, output_padding):
        x = _nhwc(x)
        Cin, Cout, KH, KW = weight.shape
        B, _, Hi, Wi = x.shape
        Ho = (Hi - 1) * stride - 2 * padding[0] + KH + output_padding[0]
        Wo = (Wi - 1) * stride - 2 * padding[1] + KW + output_padding[1]
        b = bias.detach().float().contiguous() if bias is not None else None
        if _x6_dil_ok(Cin, Ho, Wo, stride, x.numel()):
            wt = weight.detach().float().permute(2, 3, 1, 0).contiguous()                     # [kh][kw][Cout][Cin]: a free view in kernel layout
            y = _fwd_x6(x, _x6_image(weight, wt, 1), b, (Ho, Wo), Cout, KH, KW, 1, (KH - 1 - padding[0], KW - 1 - padding[1]),
                        dil=stride)
        elif _wt_ok(Cin, Cout, Ho, Wo, stride):
            wt = weight.detach().float().permute(2, 3, 1, 0).contiguous()                     # [kh][kw][Cout][Cin]: a free view in kernel layout
            y = _fwd_wt(x, wt, b, (Ho, Wo), Cout, KH, KW, 1, stride, (KH - 1 - padding[0], KW - 1 - padding[1]))
        else:
            wk = weight.detach().float().flip(2, 3).permute(2, 3, 0, 1).contiguous()         # [kh'][kw'][Cin][Cout]
            y = _fwd(x, wk, b, (Ho, Wo), Cout, KH, KW, 1, stride, (KH - 1 - padding[0], KW - 1 - padding[1]))
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, padding, bias is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, padding, has_bias = ctx.cfg
        Cin, Cout, KH, KW = weight.shape
        gy = _nhwc(gy)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wd = weight.detach().float().permute(2, 3, 1, 0).contiguous()                     # [kh][kw][Cout][Cin]
            if _x6_ok(Cout, gy.numel()):
                gx = _fwd_x6(gy, _x6_image(weight, wd, 0), None, (x.shape[2], x.shape[3]), Cin, KH, KW, stride, padding)
            else:
                gx = _fwd(gy, wd, None, (x.shape[2], x.shape[3]), Cin, KH, KW, stride, 1, padding)
        if ctx.needs_input_grad[1]:
            dw = _wrw_dispatch(weight, lambda: _wrw(gy, x, KH, KW, stride, padding, param=weight),
                               lambda acc: _wrw(gy, x, KH, KW, stride, padding, out=acc), (x, gy),
                               work=2.0 * x.numel() * Cout * KH * KW)
            gw = None if dw is False else dw.permute(3, 2, 0, 1)                               # [kh][kw][Cout][Cin] -> (Cin,Cout,kh,kw)
        if has_bias and ctx.needs_input_grad[2]:
            gb = _bias_grad(gy)
        return gx, gw, gb, None, None, None


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose forward runs dsf_conv_igemm_* (groups=1, dilation=1, zero padding, square stride)."""

    def forward(self, x):
        assert self.groups == 1 and _pair(self.dilation) == (1, 1) and self.padding_mode == 'zeros'
        s = _pair(self.stride)
        assert s[0] == s[1]
        if x.is_cuda and not self.weight.permute(2, 3, 1, 0).is_contiguous():
            kernel_layout_(self.weight, (2, 3, 1, 0))             # once: memory order [KH][KW][Ci][Co]
        return Conv2dFunction.apply(x, self.weight, self.bias, s[0], _pair(self.padding))


class ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x, output_size=None):
        assert self.groups == 1 and _pair(self.dilation) == (1, 1) and output_size is None
        s = _pair(self.stride)
        assert s[0] == s[1]
        if x.is_cuda and not self.weight.permute(2, 3, 1, 0).is_contiguous():
            kernel_layout_(self.weight, (2, 3, 1, 0))             # once: memory order [KH][KW][Cout][Cin] (bwd-data / wrw operand)
        return ConvTranspose2dFunction.apply(x, self.weight, self.bias, s[0], _pair(self.padding), _pair(self.output_padding))


# Layer registry the model builders construct from (model/backbone.py, model/hourglass.py, render_model/transfer.py).
LAYERS = {"Conv2d": None, "ConvTranspose2d": None, "fused_bn": True, "MaxPool2d": None}


def fused_heads(x, heads):
    """cat([h(x) for h in heads], dim=1) for parallel Conv2d heads of one geometry as ONE convolution over the
    concatenated output channels (the 63 + 21 channel 1x1 heads of the reference's ``finals`` become one 84-channel
    launch on the vectorised path instead of two odd-width ones plus a concatenation).  Parameters stay separate
    modules; the merged operand is rebuilt per call (a few KB) and autograd splits its gradient."""
    h0 = heads[0]
    same = all(isinstance(h, Conv2d) and h.kernel_size == h0.kernel_size and h.stride == h0.stride and
               h.padding == h0.padding and h.in_channels == h0.in_channels and (h.bias is None) == (h0.bias is None)
               for h in heads)
    if not (same and x.is_cuda):
        return torch.cat([h(x) for h in heads], dim=1)
    w = torch.cat([h.weight for h in heads], dim=0)
    b = torch.cat([h.bias for h in heads], dim=0) if h0.bias is not None else None
    return Conv2dFunction.apply(x, w, b, _pair(h0.stride)[0], _pair(h0.padding))


LAYERS["Conv2d"], LAYERS["ConvTranspose2d"] = Conv2d, ConvTranspose2d
from .nn_pool import MaxPool2d as _HipMaxPool2d          # noqa: E402  (registry entry; oracle.nets swaps in nn.MaxPool2d)
LAYERS["MaxPool2d"] = _HipMaxPool2d


def replay(rec, iters=3):
    """Re-issues one recorded igemm launch on fresh buffers and returns (avg microseconds, flops, bytes).
    flops = 2*M*N*K of the implicit GEMM, counting only taps that can hit data when dil > 1;
    bytes = the algorithmic traffic of the launch: each operand read once, the result written once."""
    kind, B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, dil, ph, pw = rec
    dev = torch.device("cuda")
    x = torch.randn(B, Ci, Hi, Wi, device=dev).contiguous(memory_format=CL)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if kind == "fwd":
        wk = torch.randn(KH, KW, Ci, Co, device=dev)
        run = lambda: _fwd(x, wk, None, (Ho, Wo), Co, KH, KW, stride, dil, (ph, pw))
    elif kind == "c1_fwd":
        wk = torch.randn(KH, KW, 1, Co, device=dev)
        run = lambda: _fwd_c1(x, wk, None, (Ho, Wo), Co, KH, stride, ph)
    elif kind == "co1_fwd":
        wk = torch.randn(KH, KW, Ci, 1, device=dev)
        run = lambda: _fwd_co1(x, wk, None, (Ho, Wo), KH, ph)
    elif kind == "c1_wrw":
        gy = torch.randn(B, Co, Ho, Wo, device=dev).contiguous(memory_format=CL)
        run = lambda: _wrw_c1(x, gy, KH, stride, ph)
    elif kind == "c1_fwd_bn":                                # nn_norm._StemFunction's forward convolution (statistics in the epilogue)
        wk = torch.randn(KH, KW, 1, Co, device=dev)
        y = torch.empty((B, Co, Ho, Wo), device=dev, dtype=torch.float32, memory_format=CL)
        rows = int(L.lib().dsf_bn_acc_rows())
        acc = torch.zeros(rows * 2 * Co, device=dev, dtype=torch.float64)
        run = lambda: check(L.lib().dsf_conv_c1_forward_bn_acc(ptr_nhwc(x), ptr(wk), None, ptr_nhwc(y), I(B), I(Hi), I(Wi), I(Ho), I(Wo), I(Co), I(KH),
                                                               I(stride), I(ph), ptr(acc), I(rows), stream_ptr()), "dsf_conv_c1_forward_bn_acc")
    elif kind.startswith("c1_wrw_bn"):                       # ... and its backward launch (BatchNorm apply arithmetic + dW), pooling mode in the name
        k, s, p = {"0": (0, 0, 0), "1": (3, 2, 1), "2": (2, 2, 0)}[kind[-1]]
        y = torch.randn(B, Co, Ho, Wo, device=dev).contiguous(memory_format=CL)
        Po, Qo = ((Ho + 2 * p - k) // s + 1, (Wo + 2 * p - k) // s + 1) if k else (Ho, Wo)
        g = torch.randn(B, Co, Po, Qo, device=dev).contiguous(memory_format=CL)
        arg = torch.randint(0, max(k * k, 1), (B, Po, Qo, Co), device=dev, dtype=torch.uint8)
        vec = [torch.rand(Co, device=dev) + 0.5 for _ in range(4)]
        rows = int(L.lib().dsf_bn_acc_rows())
        acc = torch.randn(rows * 2 * Co, device=dev, dtype=torch.float64)
        dw, gg, gb = torch.empty((KH * KW + 1) * Co, device=dev), torch.empty(Co, device=dev), torch.empty(Co, device=dev)
        ws = torch.empty(L.lib().dsf_conv_c1_workspace_bytes(I(KH), I(KW)) // 4, device=dev, dtype=torch.float32)
        run = lambda: check(L.lib().dsf_conv_c1_wrw_bn(ptr_nhwc(x), ptr_nhwc(y), ptr_nhwc(g), ptr(arg) if k else None, ptr(vec[0]), ptr(vec[1]),
                                                       ptr(vec[2]), ptr(vec[3]), ptr(acc), I(rows), I(1), I(k), I(s), I(p), ptr(dw), ptr(gg), ptr(gb),
                                                       I(0), ptr(ws), I(B), I(Hi), I(Wi), I(Ho), I(Wo), I(Co), I(KH), I(stride), I(ph), stream_ptr()),
                            "dsf_conv_c1_wrw_bn")
    elif kind == "x6":
        wk = torch.randn(KH, KW, Ci, Co, device=dev)
        img = _x6_image(wk, wk, 0)
        run = lambda: _fwd_x6(x, img, None, (Ho, Wo), Co, KH, KW, stride, (ph, pw), dil=dil)
    elif kind == "fwd_wt":
        wt = torch.randn(KH, KW, Co, Ci, device=dev)
        run = lambda: _fwd_wt(x, wt, None, (Ho, Wo), Co, KH, KW, stride, dil, (ph, pw))
    elif kind == "bwd_s1":                                   # x plays grad_out (B,Cout=Ci,..), result is grad_in (B,Cin=Co,..)
        wk = torch.randn(KH, KW, Co, Ci, device=dev)
        run = lambda: _bwd_data_s1(x, wk, (Ho, Wo), Co, KH, KW, (ph, pw))
    else:
        gy = torch.randn(B, Co, Ho, Wo, device=dev).contiguous(memory_format=CL)
        run = lambda: _wrw(x, gy, KH, KW, stride, (ph, pw))
    global RECORD
    saved, RECORD = RECORD, None
    try:
        run()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
    finally:
        RECORD = saved
    taps = KH * KW / float(dil * dil)
    nbytes = 4.0 * (B * Hi * Wi * Ci + KH * KW * Ci * Co + B * Ho * Wo * Co)
    return e0.elapsed_time(e1) * 1e3 / iters, 2.0 * B * Ho * Wo * Co * Ci * taps, nbytes


def kernel_name(rec):
    """Name of the kernel conv.hip launches for a recorded call, as rocprofv3 prints it (mirrors the host heuristics of
    conv_forward_impl / dsf_conv_igemm_wrw: tile rows BMT, stage depth BKT)."""
    kind, B, Hi, Wi, Ci, Ho, Wo, Co = rec[:8]
    dil = rec[11]
    bn = 128 if Co > 64 else 64
    M = B * Ho * Wo
    n_tiles = (Co + bn - 1) // bn
    vec = Ci % 4 == 0 and Co % 4 == 0
    if kind in ("c1_fwd", "c1_wrw"):
        return "conv_c1_%s_kernel<%d, %d>" % (kind[3:], rec[8], rec[10])
    if kind == "c1_fwd_bn":
        return "conv_c1_fwd_stats_kernel<%d, %d>" % (rec[8], rec[10])
    if kind.startswith("c1_wrw_bn"):
        return "conv_c1_wrw_bn_kernel<%d, %d, %s>" % (rec[8], rec[10], kind[-1])
    if kind == "co1_fwd":
        return "conv_co1_fwd_kernel<%d>" % rec[8]
    if kind == "x6":
        bn6 = 128 if Co > 64 else 64
        n6 = (Co + bn6 - 1) // bn6
        bmt = 64 if (bn6 == 128 and ((B * Ho * Wo + 127) // 128) * n6 < 384) else 128
        direct = os.environ.get("DSF_X6_BDIRECT", "1") != "0"
        if direct and bn6 == 64:
            bmt = 256
        import ctypes
        variant = ctypes.c_int(-1)
        L.lib().dsf_conv_x6_forward_plan(I(B), I(Hi), I(Wi), I(Ci), I(Ho), I(Wo), I(Co), I(rec[8]), I(rec[9]), I(rec[10]), I(dil), I(rec[12]),
                                         I(rec[13]), ctypes.byref(variant), None)
        if variant.value == 2:                                 # the patch-staged kernel: <BN, BMT, W, waves along n, B sets, taps, input parity>
            ip = dil == 1 and rec[8] == 4 and rec[10] == 2     # a 4 x 4 stride-2 convolution by input parity classes (round 6)
            nt = 4 if (dil == 2 or ip) else 9                   # 4: a 4 x 4 stride-2 transposed convolution, class by class
            pw = Wi if dil == 2 else Wo                         # width of the image the tile rows index
            if bn6 == 128 and pw == 8:
                bmt = 64
            return "igemm_x6p_kernel<%d, %d, %d, %d, %d, %d, %s>" % (bn6, bmt, pw, 4 if bmt == 64 else 2, 3 if (bmt == 64 and nt == 9) else 2, nt,
                                                                     "true" if ip else "false")
        if variant.value == 3:                                 # a 1 x 1 filter under dilation 2: the live parity class only (a quarter of the rows)
            bmt = 256 if bn6 == 64 else (64 if ((B * Ho * Wo // 4 + 127) // 128) * n6 < 384 else 128)
            return "igemm_x6b_kernel<%d, true, %d>" % (bn6, bmt)
        if direct and not (bmt == 64 and n6 >= 2):            # weight operand straight into the MFMA fragments (conv_x6.hip)
            return "igemm_x6b_kernel<%d, %s, %d>" % (bn6, "true" if dil == 2 else "false", bmt)
        return "igemm_x6_kernel<%d, %s, %d>" % (bn6, "true" if dil == 2 else "false", bmt)
    if kind in ("fwd", "fwd_wt", "bwd_s1"):
        wt = "false" if kind == "fwd" else "true"
        if dil == 1 and Ci >= 32 and vec:
            bmt = 64 if (bn == 128 and ((M + 127) // 128) * n_tiles < 512) else 128
            tiles = ((M + bmt - 1) // bmt) * n_tiles
            return "igemm_fwd_fast_kernel<%d, %s, %d, %d>" % (bn, wt, 16 if tiles >= 1024 else 32, bmt)
        tiles = ((M + 127) // 128) * n_tiles
        if dil == 2 and Ci >= 32 and vec and Ho % 2 == 0 and Wo % 2 == 0:
            return "igemm_fwd_dil2_kernel<%d, %d, %s>" % (bn, 16 if (tiles >= 1024 and bn == 128) else 32, wt)
        return "igemm_fwd_kernel<%d, %s>" % (bn, "true" if (Ci < 32 and dil == 1) else "false")
    if _wrw_x6_ok(Ci, Co, B * Hi * Wi * Ci, M * Co):
        # the row-staged 3 x 3 kernel (conv_x6.hip: x6_wrw_patch_applies / x6_wrw_patch_plan)
        level = int(os.environ.get("DSF_X6_WRW_PATCH", "1"))
        if (level > 0 and rec[8] == 3 and rec[9] == 3 and rec[10] == 1 and rec[12] == 1 and rec[13] == 1
                and Ho == Hi and Wo == Wi and Wi in (64, 32, 16)):
            tiles = ((Ci + 31) // 32) * ((Co + 127) // 128)
            splits = max(1, (int(os.environ.get("DSF_X6_WRWP_WGS", "0")) or 256) // tiles)
            rows = max(4, (B * Hi + splits - 1) // splits)
            want = 1024 // Wi
            if Wi == 64 and rows < want and tiles * ((B * Hi + want - 1) // want) >= 256:
                rows = want
            rows += rows & 1 if Wi == 16 else 0
            if level >= 2 or rows * Wi >= 1024:
                return "igemm_wrw_x6p_kernel<%d, %d>" % (Wi, 128 if Co > 64 else 64)
        return "igemm_wrw_x6_kernel<%d, false>" % (128 if Co > 64 else 64)      # (<.., true>: with the bias sums, dsf_conv_x6_wrw_bias)
    if vec:
        return "igemm_wrw_fast_kernel<%d, %d>" % (bn, 16 if M >= 32768 else 32)
    return "igemm_wrw_kernel<%d>" % bn
