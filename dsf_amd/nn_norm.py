"""BatchNorm2d fused with the residual add and ReLU that follow it, on the HIP kernels of
dsf_amd/csrc/norm.hip (NHWC).  Drop-in subclass of nn.BatchNorm2d: same parameters / buffers /
state-dict keys; ``forward(x, residual=None, relu=False)``.  GPU only."""
import ctypes

import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L
from ._lib import I, I64, F as CF, check, stream_ptr

CL = torch.channels_last


def supported(C):
    """channel counts of the HIP kernels: a power of two from 4 to 1024, or a multiple of 1024 (ResNet-50's 2048)"""
    c4 = C >> 2
    return C >= 4 and C % 4 == 0 and (256 % c4 == 0 if c4 <= 256 else c4 % 256 == 0)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


_WS = {}


def _workspace(device, C):
    """Zero-initialised scratch (partials + workgroup ticket) per (device, stream, C), shared by every BN call with
    that channel count issued on that stream: calls are stream-ordered and each leaves the ticket at zero."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, C)
    ws = _WS.get(key)
    if ws is None:
        fn = L.lib().dsf_bn_workspace_bytes
        fn.restype = ctypes.c_int64
        ws = torch.zeros((int(fn(I(C))) + 7) // 8, device=device, dtype=torch.float64)
        _WS[key] = ws
    return ws


# ---- accumulation rows of the finalise-free BatchNorm passes (dsf_bn_forward_acc / dsf_bn_backward_acc) ----------------
# Each pass needs a ZEROED block of dsf_bn_acc_rows() x 2C doubles.  ``with stat_pool(n_floats, device):`` around one forward +
# backward of a step provides them from ONE zero fill (as nn_conv.grad_pool does for the weight gradients); without an open
# pool, when it runs out, or in deterministic mode (float atomics build the rows) the layers take the ordered-partials path
# with its finalise launches.  DSF_BN_ACC=0 switches the pool off.
ACC = [os.environ.get("DSF_BN_ACC", "1") == "1"]
_ACC_POOL = None         # [flat zeroed tensor, next offset]
_ACC_ROWS = [0]


class stat_pool:
    def __init__(self, n_floats, device):
        self.n, self.device = int(n_floats), device

    def __enter__(self):
        global _ACC_POOL
        self.saved = _ACC_POOL
        if _ACC_POOL is None and self.n > 0 and ACC[0] and torch.device(self.device).type == "cuda" and not L.deterministic():
            _ACC_POOL = [torch.zeros(self.n, device=self.device, dtype=torch.float64), 0]
        return self

    def __exit__(self, *a):
        global _ACC_POOL
        _ACC_POOL = self.saved


def acc_rows():
    if not _ACC_ROWS[0]:
        _ACC_ROWS[0] = int(L.lib().dsf_bn_acc_rows())
    return _ACC_ROWS[0]


def _acc_take(C, device):
    """a zeroed [rows][2][C] block of the open pool, or None (no pool / exhausted / deterministic mode / unsupported C)"""
    if _ACC_POOL is None or _ACC_POOL[0].device != device or not supported(C) or L.deterministic():
        return None
    n = acc_rows() * 2 * C
    off = _ACC_POOL[1]
    if off + n > _ACC_POOL[0].numel():
        return None
    _ACC_POOL[1] = off + n
    return _ACC_POOL[0][off:off + n]


def stat_floats(module, applications=1):
    """pool size (in doubles) for ``applications`` forward + backward passes over ``module``: two blocks per fused BatchNorm and pass"""
    return applications * sum(2 * acc_rows() * 2 * m.num_features for m in module.modules()
                              if isinstance(m, FusedBatchNorm2d) and supported(m.num_features))


# ---- twin outputs: the fan-in sum of a block output's two gradients inside the BatchNorm backward --------------------------------
# The output of a residual block feeds the next block's first convolution AND its identity / downsample path (reference
# model/resnet.py:39-55, 78-98); autograd sums the two gradients with an elementwise pass (2R + 1W of the activation) before
# the BatchNorm backward that produced the output can run.  With ``twin=True`` the fused BatchNorm hands out its output TWICE (two
# tensors on one storage: ``y`` and ``y._dsf_twin``); a consumer that reads the output twice takes one each (``take_twin``),
# and the backward receives the two gradients separately: its sums pass adds them on the fly (dsf_bn_backward_acc_pair).
# DSF_BN_TWIN=0: one output, autograd adds.
TWIN = [os.environ.get("DSF_BN_TWIN", "1") == "1"]
# DSF_BN_POOL=0: the stem's BatchNorm + ReLU and MaxPool2d run as separate layers (see FusedBatchNorm2d.forward_pooled)
POOL_FUSED = [os.environ.get("DSF_BN_POOL", "1") == "1"]
# second application of a layer in one backward pass adds its dgamma / dbeta into the first one's buffers (DSF_BN_AFFINE_ACC=0: off)
AFFINE_ACCUMULATE = [os.environ.get("DSF_BN_AFFINE_ACC", "1") == "1"]
_TASK_ID = getattr(torch._C, "_current_graph_task_id", None)         # (private: the id of the running backward pass, -1 outside one)


def _alias(y):
    """a second tensor on y's storage (not a view: no ``_base`` reference back to y, so hanging it on y makes no cycle)"""
    return torch.empty(0, device=y.device, dtype=y.dtype).set_(y.untyped_storage(), y.storage_offset(), y.shape, y.stride())


def take_twin(x):
    """-> (x, x2): the two handles of a twin output (x2 is x itself when x has none).  The twin is handed out once."""
    t = x.__dict__.pop("_dsf_twin", None) if torch.is_tensor(x) else None
    return (x, t) if t is not None else (x, x)


class _BNFunction(Function):
    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, eps, momentum, relu, part=None, rows=0, acc=None, acc_filled=0,
                twin=False, pool=None):
        ctx.set_materialize_grads(False)
        x = x.contiguous(memory_format=CL)
        ctx.pool = pool
        if pool is not None:
            # (k, stride, pad) of a MaxPool2d behind BatchNorm + ReLU (the backbone stem, reference model/backbone.py:200-204): the
            # apply pass writes the POOLED output and the argmax bytes only (forward_pooled checked the preconditions)
            k, s, p = pool
            B, C, H, W = x.shape
            Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
            y = torch.empty((B, C, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=CL)
            arg = torch.empty((B, Ho, Wo, C), device=x.device, dtype=torch.uint8)
            mean = torch.empty(C, device=x.device, dtype=torch.float32)
            invstd = torch.empty(C, device=x.device, dtype=torch.float32)
            check(L.lib().dsf_bn_relu_pool_forward(_p(x), _p(gamma), _p(beta), I(B), I(H), I(W), I(C), I(k), I(s), I(p), CF(eps), CF(momentum),
                                                   _p(running_mean), _p(running_var), _p(y), _p(arg), _p(mean), _p(invstd), _p(acc),
                                                   I(int(acc_filled)), stream_ptr()), "dsf_bn_relu_pool_forward")
            ctx.save_for_backward(x, None, gamma, beta, mean, invstd, arg)
            ctx.cfg = (True, False, gamma is not None, beta is not None)
            return y
        if residual is not None:
            residual = residual.contiguous(memory_format=CL)
        B, C, H, W = x.shape
        M = B * H * W
        y = torch.empty_like(x, memory_format=CL)
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(C, device=x.device, dtype=torch.float32)
        if acc is not None:
            # statistics rows accumulated by the producing convolution's epilogue (acc_filled) or by this call's own reduction;
            # the apply kernel folds them in its prologue: no finalise launch
            check(L.lib().dsf_bn_forward_acc(_p(x), _p(residual), _p(gamma), _p(beta), I64(M), I(C), CF(eps), CF(momentum),
                                             I(int(relu)), _p(running_mean), _p(running_var), _p(y), _p(mean), _p(invstd), _p(acc),
                                             I(int(acc_filled)), stream_ptr()), "dsf_bn_forward_acc")
        elif part is not None and rows > 0:
            # the producing convolution's epilogue already reduced the tile sums (dsf_conv_x6_forward_bn): finalise + apply only
            check(L.lib().dsf_bn_forward_from_stats(_p(x), _p(residual), _p(gamma), _p(beta), I64(M), I(C), CF(eps), CF(momentum),
                                                    I(int(relu)), _p(running_mean), _p(running_var), _p(y), _p(mean), _p(invstd),
                                                    _p(part), I(int(rows)), stream_ptr()), "dsf_bn_forward_from_stats")
        else:
            ws = _workspace(x.device, C)
            check(L.lib().dsf_bn_forward(_p(x), _p(residual), _p(gamma), _p(beta), I64(M), I(C), CF(eps), CF(momentum),
                                         I(int(relu)), _p(running_mean), _p(running_var), _p(y), _p(mean), _p(invstd), _p(ws),
                                         stream_ptr()), "dsf_bn_forward")
        # ReLU mask in the backward: recomputed from x when no residual was added (y is then not kept alive for it)
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, gamma, beta, mean, invstd, None)
        ctx.cfg = (relu, residual is not None, gamma is not None, beta is not None)
        if twin:
            return y, _alias(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gy2=None):
        n_in = 15
        if gy is None:
            gy, gy2 = gy2, None
        if gy is None:                                        # neither handle of the output reached the loss
            return (None,) * n_in
        x, y, gamma, beta, mean, invstd, arg = ctx.saved_tensors
        relu, has_res, has_g, has_b = ctx.cfg
        relu_mode = 0 if not relu else (1 if has_res else 2)
        gy = gy.contiguous(memory_format=CL)
        if gy2 is not None:
            gy2 = gy2.contiguous(memory_format=CL)
        B, C, H, W = x.shape
        M = B * H * W
        gx = torch.empty_like(x, memory_format=CL)
        # the residual's gradient is the (summed, masked) incoming gradient; unmasked and alone it IS gy: nothing to write
        gres = None if not has_res else (gy if (relu_mode == 0 and gy2 is None) else torch.empty_like(x, memory_format=CL))
        gres_out = gres if gres is not gy else None
        gg = torch.empty(C, device=x.device, dtype=torch.float32) if has_g else None
        gb = torch.empty(C, device=x.device, dtype=torch.float32) if has_b else None
        # A layer applied TWICE before one backward() (the network on the synthetic and on the real batch, train_render.py:628-703):
        # the pass's first contribution to dgamma / dbeta is recorded on the parameter, the second ADDS into those buffers
        # (accumulate_affine) and hands autograd nothing -- else the engine sums the two with a launch per parameter (~90 per
        # config-5 step).  Same stream only (the engine holds the first buffer until every contribution has arrived).
        accumulate, rec_key = 0, None
        if has_g and has_b and _TASK_ID is not None and gamma.is_leaf and beta.is_leaf and AFFINE_ACCUMULATE[0]:
            task, st_now = _TASK_ID(), stream_ptr().value
            rec = gamma.__dict__.get("_dsf_bnpass")
            if task >= 0 and rec is not None and rec[0] == task and rec[1] == st_now and rec[2].shape == gg.shape:
                accumulate, gg_k, gb_k = 1, rec[2], rec[3]
            elif task >= 0:
                rec_key = (task, st_now)
        acc = _acc_take(C, x.device)
        gg_w, gb_w = (gg_k, gb_k) if accumulate else (gg, gb)
        pooled = ctx.pool is not None and acc is not None
        if ctx.pool is not None:
            k, s, p = ctx.pool
            if pooled:                                        # both passes gather the pooling's backward from the pooled gradient
                check(L.lib().dsf_bn_relu_pool_backward(_p(x), _p(gy), _p(arg), _p(gamma), _p(beta), _p(mean), _p(invstd), I(B), I(H), I(W),
                                                        I(C), I(k), I(s), I(p), _p(gx), _p(gg_w), _p(gb_w), I(accumulate), _p(acc),
                                                        stream_ptr()), "dsf_bn_relu_pool_backward")
            else:                                             # (the statistics pool ran out between the passes): the two layers' own kernels
                g_full = torch.empty_like(x, memory_format=CL)
                check(L.lib().dsf_maxpool_backward(_p(gy), _p(arg), _p(g_full), I(B), I(H), I(W), I(C), I(gy.shape[2]), I(gy.shape[3]), I(k),
                                                   I(s), I(p), stream_ptr()), "dsf_maxpool_backward")
                gy = g_full
        if not pooled and acc is not None:
            check(L.lib().dsf_bn_backward_acc_pair(_p(x), _p(gy), _p(gy2), _p(y), _p(gamma), _p(beta), _p(mean), _p(invstd), I64(M), I(C),
                                                   I(relu_mode), _p(gx), _p(gres_out), _p(gg_w), _p(gb_w), I(accumulate), _p(acc), stream_ptr()),
                  "dsf_bn_backward_acc_pair")
        elif not pooled:
            ws = _workspace(x.device, C)
            check(L.lib().dsf_bn_backward_pair(_p(x), _p(gy), _p(gy2), _p(y), _p(gamma), _p(beta), _p(mean), _p(invstd), I64(M), I(C),
                                               I(relu_mode), _p(gx), _p(gres_out), _p(gg_w), _p(gb_w), I(accumulate), _p(ws), stream_ptr()),
                  "dsf_bn_backward_pair")
        if accumulate:
            gg = gb = None                                    # added into the buffers the first contribution handed to autograd
        elif rec_key is not None:
            gamma.__dict__["_dsf_bnpass"] = rec_key + (gg, gb)
            # (autograd adopts a gradient as ``.grad`` without a copy only when nobody else holds the tensor OBJECT: hand it views)
            gg, gb = gg.view(C), gb.view(C)
        return (gx, gres, gg, gb) + (None,) * (n_in - 4)


class FusedBatchNorm2d(nn.BatchNorm2d):
    """``fuse_relu=True`` makes plain ``bn(x)`` apply the ReLU too (used inside nn.Sequential stacks,
    where the following nn.ReLU slot is replaced by nn.Identity so module indices / keys stay put)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, fuse_relu=False):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine, track_running_stats=track_running_stats)
        self.fuse_relu = fuse_relu
        self._pending_batches = 0      # training steps not yet added to num_batches_tracked (flushed when it is read out)

    def flush_batch_counter(self):
        """num_batches_tracked is only an output when momentum is set (the running statistics do not depend on it), so
        the per-step `add_(1)` launch is deferred and folded in when the buffer is read (attribute access, state dict)."""
        n, self._pending_batches = self._pending_batches, 0
        if n:
            nbt = self._buffers.get("num_batches_tracked")
            if nbt is not None:
                nbt.add_(n)

    def folded_affine(self):
        """(scale, shift) of the frozen-statistics transform y = x * scale + shift, cached until the statistics or the affine
        parameters change (torch's version counters for ordinary writes, ``_stats_epoch`` for the training kernels' raw-pointer
        updates of the running statistics, the global write epoch for FusedAdamW)."""
        w, b = self.weight, self.bias
        key = (self.running_mean._version, self.running_var._version, None if w is None else w._version,
               None if b is None else b._version, self.__dict__.get("_stats_epoch", 0), L.WRITE_EPOCH[0], self.running_mean.data_ptr())
        hit = self.__dict__.get("_folded")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                scale = torch.rsqrt(self.running_var + self.eps)
                if w is not None:
                    scale = scale * w
                shift = -self.running_mean * scale
                if b is not None:
                    shift = shift + b
            hit = self.__dict__["_folded"] = (key, scale.contiguous(), shift.contiguous())
        return hit[1], hit[2]

    @staticmethod
    def _out(res):
        if isinstance(res, tuple):                           # (y, second handle): see take_twin
            res[0].__dict__["_dsf_twin"] = res[1]
            return res[0]
        return res

    def __getattr__(self, name):
        if name == "num_batches_tracked" and self.__dict__.get("_pending_batches"):
            self.flush_batch_counter()
        return super().__getattr__(name)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush_batch_counter()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._pending_batches = 0
        super()._load_from_state_dict(*args, **kwargs)

    def _count_training_batch(self):
        """bookkeeping of a training-mode pass whose kernels rewrite the running statistics through raw pointers (see forward)"""
        if self._buffers.get("num_batches_tracked") is not None:
            self._pending_batches += 1
        self.__dict__["_stats_epoch"] = self.__dict__.get("_stats_epoch", 0) + 1

    def forward_pooled(self, x, k, stride, pad, stats=None):
        """MaxPool2d(k, stride, pad)(relu(bn(x))) of a training-mode layer as ONE apply pass that writes the pooled map only, with the
        pooling's backward inside the BatchNorm backward (dsf_bn_relu_pool_forward / _backward; the backbone stem, reference
        model/backbone.py:200-204).  None when this call cannot take that path -- the caller then runs the separate layers."""
        if not (POOL_FUSED[0] and type(self) is FusedBatchNorm2d and self.training and self.track_running_stats and self.momentum is not None
                and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.numel() and supported(x.shape[1])
                and k in (2, 3) and k <= 2 * stride and 0 <= 2 * pad <= k and x.numel() // 4 < 2 ** 31
                and min(x.shape[2], x.shape[3]) + 2 * pad >= k):
            return None
        # stats = ("acc", rows block, filled): the block a producing convolution was handed for its epilogue (conv_bn_act's protocol)
        acc, filled = (stats[1], stats[2]) if stats is not None else (_acc_take(x.shape[1], x.device), 0)
        if acc is None:                                      # no open statistics pool / deterministic mode
            return None
        self._count_training_batch()
        return _BNFunction.apply(x, None, self.weight, self.bias, self.running_mean, self.running_var, self.eps, self.momentum, True, None, 0,
                                 acc, filled, False, (k, stride, pad))

    def forward(self, x, residual=None, relu=None, stats=None, twin=False):
        """``twin=True`` (training path only): the output also carries a second handle ``y._dsf_twin`` for a consumer that reads it
        twice (take_twin); every other path ignores the flag."""
        relu = self.fuse_relu if relu is None else relu
        if not x.is_cuda:
            raise RuntimeError("dsf_amd FusedBatchNorm2d runs on the GPU only (got %s)" % x.device)
        C = x.shape[1]
        if x.numel() == 0:                                   # empty batch: nothing to normalise, no statistics to update
            return x.contiguous(memory_format=CL)
        use_batch_stats = self.training or not self.track_running_stats
        if supported(C) and x.dtype == torch.float32:
            if use_batch_stats:
                if self.momentum is None:
                    raise NotImplementedError("FusedBatchNorm2d: cumulative moving average (momentum=None) is not used by "
                                              "the reference and not implemented")
                if self.training and self.track_running_stats and self._buffers.get("num_batches_tracked") is not None:   # (not the attribute: reading it flushes)
                    self._pending_batches += 1
                if self.training and self.track_running_stats:
                    self.__dict__["_stats_epoch"] = self.__dict__.get("_stats_epoch", 0) + 1   # the kernel rewrites the running statistics
                mom = self.momentum
                rm = self.running_mean if (self.training and self.track_running_stats) else None
                rv = self.running_var if (self.training and self.track_running_stats) else None
                twin = bool(twin and TWIN[0] and torch.is_grad_enabled())
                if stats is not None and isinstance(stats[0], str):            # ("acc", rows block, filled)
                    return self._out(_BNFunction.apply(x, residual, self.weight, self.bias, rm, rv, self.eps, mom, relu, None, 0, stats[1], stats[2], twin))
                part, rows = stats if stats is not None else (None, 0)
                if part is None:
                    acc = _acc_take(C, x.device)
                    if acc is not None:
                        return self._out(_BNFunction.apply(x, residual, self.weight, self.bias, rm, rv, self.eps, mom, relu, None, 0, acc, 0, twin))
                return self._out(_BNFunction.apply(x, residual, self.weight, self.bias, rm, rv, self.eps, mom, relu, part, rows, None, 0, twin))
            if torch.is_grad_enabled() and (x.requires_grad or (residual is not None and residual.requires_grad) or
                                            (self.weight is not None and self.weight.requires_grad) or
                                            (self.bias is not None and self.bias.requires_grad)):
                y = super().forward(x)                      # frozen-statistics BN inside a differentiated graph: torch's kernels
                if residual is not None:
                    y = y + residual
                return F.relu(y) if relu else y
            # eval: frozen statistics, one streaming pass
            xc = x.contiguous(memory_format=CL)
            res = residual.contiguous(memory_format=CL) if residual is not None else None
            y = torch.empty_like(xc, memory_format=CL)
            invstd = torch.rsqrt(self.running_var + self.eps)
            B, _, H, W = xc.shape
            check(L.lib().dsf_bn_apply(_p(xc), _p(res), _p(self.weight), _p(self.bias), _p(self.running_mean), _p(invstd),
                                       I64(B * H * W), I(C), I(int(relu)), _p(y), stream_ptr()), "dsf_bn_apply")
            return y
        # channel counts the kernels do not cover (e.g. 2048 in ResNet-50 layer4): torch's own kernels
        y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y


class _SyncBNFunction(Function):
    """Training-mode BatchNorm (+ residual)(+ ReLU) over the batch of EVERY rank of ``group``: one all-reduce of 2C + 1
    doubles in the forward pass (per-channel sum, sum of squares, element count) and one of 2C doubles in the backward
    pass (sum g, sum g xhat) -- torch.nn.SyncBatchNorm's arithmetic (global mean / biased variance for the normalisation,
    unbiased global variance into the running buffers, local dgamma / dbeta) on the fused HIP passes."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, eps, momentum, relu, group, part=None, rows=0):
        import torch.distributed as dist
        x = x.contiguous(memory_format=CL)
        if residual is not None:
            residual = residual.contiguous(memory_format=CL)
        B, C, H, W = x.shape
        M = B * H * W
        sums = torch.empty(2 * C + 1, device=x.device, dtype=torch.float64)
        sums[2 * C:].fill_(float(M))
        ws = _workspace(x.device, C)
        check(L.lib().dsf_bn_local_sums(_p(x), I64(M), I(C), _p(part) if (part is not None and rows > 0) else None,
                                        I(int(rows) if part is not None else 0), _p(sums), _p(ws), stream_ptr()), "dsf_bn_local_sums")
        dist.all_reduce(sums, group=group)
        y = torch.empty_like(x, memory_format=CL)
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(C, device=x.device, dtype=torch.float32)
        check(L.lib().dsf_bn_forward_from_sums(_p(x), _p(residual), _p(gamma), _p(beta), I64(M), I(C), CF(eps), CF(momentum),
                                               I(int(relu)), _p(running_mean), _p(running_var), _p(y), _p(mean), _p(invstd),
                                               _p(sums), stream_ptr()), "dsf_bn_forward_from_sums")
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, gamma, beta, mean, invstd, sums)
        ctx.cfg = (relu, residual is not None, gamma is not None, beta is not None, group)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        import torch.distributed as dist
        x, y, gamma, beta, mean, invstd, sums = ctx.saved_tensors
        relu, has_res, has_g, has_b, group = ctx.cfg
        relu_mode = 0 if not relu else (1 if has_res else 2)
        gy = gy.contiguous(memory_format=CL)
        B, C, H, W = x.shape
        M = B * H * W
        gx = torch.empty_like(x, memory_format=CL)
        gres = torch.empty_like(x, memory_format=CL) if has_res else None
        gg = torch.empty(C, device=x.device, dtype=torch.float32) if has_g else None
        gb = torch.empty(C, device=x.device, dtype=torch.float32) if has_b else None
        gs = torch.empty(2 * C, device=x.device, dtype=torch.float64)
        ws = _workspace(x.device, C)
        check(L.lib().dsf_bn_backward_sums(_p(x), _p(gy), _p(y), _p(gamma), _p(beta), _p(mean), _p(invstd), I64(M), I(C), I(relu_mode),
                                           _p(gs), _p(gg), _p(gb), _p(ws), stream_ptr()), "dsf_bn_backward_sums")
        dist.all_reduce(gs, group=group)
        count = sums[2 * C:]                                   # global element count, on the device
        check(L.lib().dsf_bn_backward_apply(_p(x), _p(gy), _p(y), _p(gamma), _p(beta), _p(mean), _p(invstd), _p(gs), _p(count), I64(M),
                                            I(C), I(relu_mode), _p(gx), _p(gres), stream_ptr()), "dsf_bn_backward_apply")
        return gx, gres, gg, gb, None, None, None, None, None, None, None, None


class FusedSyncBatchNorm2d(FusedBatchNorm2d):
    """FusedBatchNorm2d whose TRAINING statistics are those of the global batch (all ranks of ``process_group``): the
    drop-in for ``torch.nn.SyncBatchNorm`` on the fused HIP passes -- same parameters, buffers and state-dict keys, same
    ``forward(x, residual=None, relu=None, stats=None)`` as the module it replaces (parallel.convert_sync_batchnorm swaps the
    class in place).  Single-rank groups and evaluation mode fall through to FusedBatchNorm2d."""
    process_group = None
    force_sync = False            # True: the cross-replica passes (and their collectives) also run in a ONE-rank group (tests/test_gpu_rccl.py)

    def forward(self, x, residual=None, relu=None, stats=None, twin=False):
        import torch.distributed as dist
        world = dist.get_world_size(self.process_group) if (dist.is_available() and dist.is_initialized()) else 1
        if self.force_sync and dist.is_available() and dist.is_initialized():
            world = max(world, 2)
        if not (self.training and self.track_running_stats and world > 1 and x.is_cuda and x.dtype == torch.float32 and
                supported(x.shape[1]) and x.numel() > 0 and self.momentum is not None):
            return super().forward(x, residual, relu, stats, twin)
        # (the cross-replica passes hand out one output: autograd sums a two-consumer gradient itself)
        relu = self.fuse_relu if relu is None else relu
        if self._buffers.get("num_batches_tracked") is not None:
            self._pending_batches += 1
        self.__dict__["_stats_epoch"] = self.__dict__.get("_stats_epoch", 0) + 1
        part, rows = (stats if (stats is not None and not isinstance(stats[0], str)) else (None, 0))
        return _SyncBNFunction.apply(x, residual, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                     self.momentum, relu, self.process_group, part, rows)


def instance_norm_act(x, residual=None, relu=False, eps=1e-5, acc=None):
    """``nn.InstanceNorm2d(affine=False, track_running_stats=False)(x)`` (+ residual) (+ ReLU) on the HIP passes, inference only
    (no autograd): the frozen transfer generator's normalisation.  x (B,C,H,W) channels_last; ``acc``: B * 2C zeroed doubles of
    scratch (a slice of one buffer zeroed for the whole network pass), allocated here when None."""
    x = x.contiguous(memory_format=CL)
    B, C, H, W = x.shape
    res = residual.contiguous(memory_format=CL) if residual is not None else None
    y = torch.empty_like(x, memory_format=CL)
    if acc is None:
        acc = torch.zeros(B * 2 * C, device=x.device, dtype=torch.float64)
    check(L.lib().dsf_instnorm_forward(_p(x), _p(res), I(B), I64(H * W), I(C), CF(eps), I(int(relu)), _p(y), _p(acc), stream_ptr()),
          "dsf_instnorm_forward")
    return y


def reflect_pad(x, pad):
    """``nn.ReflectionPad2d(pad)`` on a channels_last tensor, staying channels_last (torch's kernel is NCHW: two layout copies)"""
    x = x.contiguous(memory_format=CL)
    B, C, H, W = x.shape
    y = torch.empty((B, C, H + 2 * pad, W + 2 * pad), device=x.device, dtype=torch.float32, memory_format=CL)
    check(L.lib().dsf_reflect_pad_nhwc(_p(x), _p(y), I(B), I(H), I(W), I(C), I(pad), stream_ptr()), "dsf_reflect_pad_nhwc")
    return y


def bn_act(bn, x, residual=None, relu=False, twin=False):
    """bn(x) (+ residual) (relu) for either the fused module or a plain nn.BatchNorm2d (CPU twin)."""
    if isinstance(bn, FusedBatchNorm2d):
        return bn(x, residual, relu, twin=twin)
    y = bn(x)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


# BatchNorm statistics from the producing convolution's epilogue (DSF_BN_EPILOGUE=0: the separate reduction pass)
EPILOGUE_STATS = [os.environ.get("DSF_BN_EPILOGUE", "1") == "1"]
EPILOGUE_AFFINE = [os.environ.get("DSF_BN_EPILOGUE", "1") == "1"]


# DSF_C1_BN=0: the stem convolution and its BatchNorm stay two autograd nodes (see _StemFunction)
C1_BN = [os.environ.get("DSF_C1_BN", "1") == "1"]


class _StemFunction(Function):
    """[1-channel convolution -> BatchNorm (-> ReLU) (-> MaxPool2d)] of a network stem as ONE autograd node (reference
    model/backbone.py:196-204).  Forward: the convolution's epilogue leaves the batch statistics (dsf_conv_c1_forward_bn_acc), the
    apply pass normalises (and pools).  Backward: nobody needs the gradient of the convolution OUTPUT except the dW sum, so after the
    sums pass ONE launch does the BatchNorm backward's apply arithmetic and the dW accumulation (dsf_conv_c1_wrw_bn): the 134 MB
    gradient is neither written nor read back.  The input must not require a gradient (conv_bn_act checks)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, eps, momentum, relu, stride, pad, pool, acc):
        from . import nn_conv
        ctx.set_materialize_grads(False)
        x = nn_conv._nhwc(x)
        Co, _, K, _ = weight.shape
        B, _, Hi, Wi = x.shape
        Ho, Wo = (Hi + 2 * pad - K) // stride + 1, (Wi + 2 * pad - K) // stride + 1
        wk = weight.detach().permute(2, 3, 1, 0).contiguous()                  # (a free view: the weight has kernel layout)
        if nn_conv.RECORD is not None:
            nn_conv.RECORD.append(("c1_fwd_bn", B, Hi, Wi, 1, Ho, Wo, Co, K, K, stride, 1, pad, pad))
        y = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=CL)
        lib = L.lib()
        check(lib.dsf_conv_c1_forward_bn_acc(_p(x), _p(wk), _p(bias.detach() if bias is not None else None), _p(y), I(B), I(Hi), I(Wi), I(Ho), I(Wo), I(Co), I(K), I(stride), I(pad), _p(acc),
                                             I(int(lib.dsf_bn_acc_rows())), stream_ptr()), "dsf_conv_c1_forward_bn_acc")
        mean = torch.empty(Co, device=x.device, dtype=torch.float32)
        invstd = torch.empty(Co, device=x.device, dtype=torch.float32)
        arg = None
        if pool is not None:
            k, s, p = pool
            Po, Qo = (Ho + 2 * p - k) // s + 1, (Wo + 2 * p - k) // s + 1
            out = torch.empty((B, Co, Po, Qo), device=x.device, dtype=torch.float32, memory_format=CL)
            arg = torch.empty((B, Po, Qo, Co), device=x.device, dtype=torch.uint8)
            check(lib.dsf_bn_relu_pool_forward(_p(y), _p(gamma), _p(beta), I(B), I(Ho), I(Wo), I(Co), I(k), I(s), I(p), CF(eps), CF(momentum),
                                               _p(running_mean), _p(running_var), _p(out), _p(arg), _p(mean), _p(invstd), _p(acc), I(1),
                                               stream_ptr()), "dsf_bn_relu_pool_forward")
        else:
            out = torch.empty_like(y, memory_format=CL)
            check(lib.dsf_bn_forward_acc(_p(y), _p(None), _p(gamma), _p(beta), I64(B * Ho * Wo), I(Co), CF(eps), CF(momentum), I(int(relu)),
                                         _p(running_mean), _p(running_var), _p(out), _p(mean), _p(invstd), _p(acc), I(1), stream_ptr()),
                  "dsf_bn_forward_acc")
        ctx.save_for_backward(x, y, weight, gamma, beta, mean, invstd, arg)
        ctx.cfg = (bool(relu), stride, pad, pool, bias is not None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from . import nn_conv
        n_in = 14
        if g is None:
            return (None,) * n_in
        x, y, weight, gamma, beta, mean, invstd, arg = ctx.saved_tensors
        relu, stride, pad, pool, has_bias = ctx.cfg
        g = g.contiguous(memory_format=CL)
        B, Co, Ho, Wo = y.shape
        _, _, Hi, Wi = x.shape
        K = weight.shape[2]
        lib = L.lib()
        acc = _acc_take(Co, x.device)
        if acc is None:                                       # (the statistics pool ran out between the passes)
            acc = torch.zeros(acc_rows() * 2 * Co, device=x.device, dtype=torch.float64)
        has_g, has_b = gamma is not None, beta is not None
        gg = torch.empty(Co, device=x.device, dtype=torch.float32) if has_g else None
        gb = torch.empty(Co, device=x.device, dtype=torch.float32) if has_b else None
        accumulate, rec_key = 0, None                         # (a second application in one backward pass: see _BNFunction.backward)
        if has_g and has_b and _TASK_ID is not None and gamma.is_leaf and beta.is_leaf and AFFINE_ACCUMULATE[0]:
            task, st_now = _TASK_ID(), stream_ptr().value
            rec = gamma.__dict__.get("_dsf_bnpass")
            if task >= 0 and rec is not None and rec[0] == task and rec[1] == st_now and rec[2].shape == gg.shape:
                accumulate, gg_k, gb_k = 1, rec[2], rec[3]
            elif task >= 0:
                rec_key = (task, st_now)
        gg_w, gb_w = (gg_k, gb_k) if accumulate else (gg, gb)
        # sums pass (no gradient written), then apply + dW in one launch
        if pool is not None:
            k, s, p = pool
            check(lib.dsf_bn_relu_pool_backward(_p(y), _p(g), _p(arg), _p(gamma), _p(beta), _p(mean), _p(invstd), I(B), I(Ho), I(Wo), I(Co),
                                                I(k), I(s), I(p), _p(None), _p(None), _p(None), I(0), _p(acc), stream_ptr()),
                  "dsf_bn_relu_pool_backward")
        else:
            k = s = p = 0
            check(lib.dsf_bn_backward_acc(_p(y), _p(g), _p(None), _p(gamma), _p(beta), _p(mean), _p(invstd), I64(B * Ho * Wo), I(Co),
                                          I(2 if relu else 0), _p(None), _p(None), _p(None), _p(None), _p(acc), stream_ptr()),
                  "dsf_bn_backward_acc")
        if nn_conv.RECORD is not None:
            nn_conv.RECORD.append(("c1_wrw_bn%d" % {0: 0, 3: 1, 2: 2}[k], B, Hi, Wi, 1, Ho, Wo, Co, K, K, stride, 1, pad, pad))
        nn_conv.main_stream_weight_grad(weight)
        dw = nn_conv._pool_take((K * K + 1) * Co, x.device)                   # rows 0 .. K*K - 1: dW; row K*K: the bias gradient
        if dw is None:
            dw = torch.empty((K * K + 1) * Co, device=x.device, dtype=torch.float32)
        ws = torch.empty(lib.dsf_conv_c1_workspace_bytes(I(K), I(K)) // 4, device=x.device, dtype=torch.float32)
        check(lib.dsf_conv_c1_wrw_bn(_p(x), _p(y), _p(g), _p(arg), _p(gamma), _p(beta), _p(mean), _p(invstd), _p(acc), I(acc_rows()),
                                     I(int(relu)), I(k), I(s), I(p), _p(dw), _p(gg_w), _p(gb_w), I(accumulate), _p(ws), I(B), I(Hi), I(Wi),
                                     I(Ho), I(Wo), I(Co), I(K), I(stride), I(pad), stream_ptr()), "dsf_conv_c1_wrw_bn")
        if accumulate:
            gg = gb = None
        elif rec_key is not None:
            gamma.__dict__["_dsf_bnpass"] = rec_key + (gg, gb)
            gg, gb = gg.view(Co), gb.view(Co)
        gw = dw[:K * K * Co].view(K, K, 1, Co).permute(3, 2, 0, 1) if ctx.needs_input_grad[1] else None
        gbias = dw[K * K * Co:] if (has_bias and ctx.needs_input_grad[2]) else None
        return (None, gw, gbias, gg, gb) + (None,) * (n_in - 5)


def _stem_node(conv, bn, x, relu, pool):
    """conv_bn_act's route into _StemFunction, or None: a bias-free 1-channel convolution of this package whose input needs no gradient, a
    training-mode FusedBatchNorm2d, more than 1024 output pixels, an open statistics pool; pooling (3, 2, 1) or (2, 2, 0) behind a ReLU"""
    from . import nn_conv
    if not (C1_BN[0] and nn_conv.C1_STATS[0] and EPILOGUE_STATS[0] and type(bn) is FusedBatchNorm2d and type(conv) is nn_conv.Conv2d and
            bn.training and bn.track_running_stats and x.is_cuda and supported(bn.num_features) and nn_conv.STATS is None and
            (conv.bias is None or conv.bias.dtype == torch.float32) and not x.requires_grad and x.dim() == 4 and x.dtype == torch.float32 and x.shape[0] > 0 and bn.momentum is not None and
            conv.weight.dtype == torch.float32 and conv.groups == 1 and tuple(conv.dilation) == (1, 1) and conv.padding_mode == "zeros"):
        return None
    s, pd = tuple(conv.stride), tuple(conv.padding)
    Co, Ci, KH, KW = conv.weight.shape
    if s[0] != s[1] or pd[0] != pd[1] or Co != bn.num_features or not nn_conv._c1_ok(Ci, Co, KH, KW, s[0], pd):
        return None
    Ho, Wo = (x.shape[2] + 2 * pd[0] - KH) // s[0] + 1, (x.shape[3] + 2 * pd[0] - KW) // s[0] + 1
    if x.shape[0] * Ho * Wo <= 1024 or x.shape[0] * Ho * Wo * (Co // 4) >= 2 ** 31:
        return None
    if pool is not None and not (relu and POOL_FUSED[0] and tuple(pool[:3]) in ((3, 2, 1), (2, 2, 0)) and min(Ho, Wo) + 2 * pool[2] >= pool[0]):
        return None
    if not conv.weight.permute(2, 3, 1, 0).is_contiguous():
        nn_conv.kernel_layout_(conv.weight, (2, 3, 1, 0))
    acc = _acc_take(Co, x.device)
    if acc is None:
        return None
    bn._count_training_batch()
    return _StemFunction.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, bool(relu), s[0], pd[0],
                               tuple(pool[:3]) if pool is not None else None, acc)


def _pooled(bn, y, pool, stats=None):
    """MaxPool2d(relu(bn(y))) for conv_bn_act's ``pool`` = (k, stride, pad, the MaxPool2d module): the pooled apply pass when the
    layer takes it (forward_pooled), the BatchNorm and the pooling module one after the other otherwise"""
    out = bn.forward_pooled(y, pool[0], pool[1], pool[2], stats=stats) if isinstance(bn, FusedBatchNorm2d) and bn.fuse_relu else None
    return out if out is not None else pool[3](bn_act(bn, y, None, getattr(bn, "fuse_relu", False), False) if stats is None
                                               else bn(y, None, None, stats=stats))


def conv_bn_act(conv, bn, x, residual=None, relu=None, twin=False, pool=None):
    """``bn(conv(x))`` (+ residual) (relu).  Both this package's HIP layers, training mode: the BatchNorm batch statistics
    come from the convolution's epilogue (one pass over the convolution output less, two launches instead of three).
    Evaluation mode without autograd: the whole frozen-statistics BatchNorm (+ residual)(+ ReLU) rides in the convolution's
    output epilogue -- one launch, no second pass.  Any other combination is the plain composition."""
    from . import nn_conv
    ours = isinstance(bn, FusedBatchNorm2d) and isinstance(conv, (nn_conv.Conv2d, nn_conv.ConvTranspose2d)) and x.is_cuda
    if (ours and EPILOGUE_AFFINE[0] and not bn.training and bn.track_running_stats and supported(bn.num_features) and
            x.dtype == torch.float32 and nn_conv.STATS is None and not torch.is_grad_enabled() and pool is None):
        r = bn.fuse_relu if relu is None else relu
        scale, shift = bn.folded_affine()
        res = residual.contiguous(memory_format=CL) if residual is not None else None
        req = nn_conv.AffineRequest(scale, shift, res, r)
        nn_conv.STATS = req
        try:
            y = conv(x)
        finally:
            nn_conv.STATS = None
        return y if req.applied else bn(y, residual, r)
    fusable = (EPILOGUE_STATS[0] and isinstance(bn, FusedBatchNorm2d) and isinstance(conv, (nn_conv.Conv2d, nn_conv.ConvTranspose2d)) and bn.training and
               bn.track_running_stats and conv.bias is None and x.is_cuda and supported(bn.num_features) and nn_conv.STATS is None)
    if residual is None and not twin and getattr(conv, "in_channels", 0) == 1 and isinstance(bn, FusedBatchNorm2d):     # (a network's stem)
        out = _stem_node(conv, bn, x, bn.fuse_relu if relu is None else relu, pool)
        if out is not None:
            return out
    if not fusable:
        if pool is not None:
            return _pooled(bn, conv(x), pool)
        return bn_act(bn, conv(x), residual, relu if relu is not None else getattr(bn, "fuse_relu", False), twin)
    req = nn_conv.StatsRequest()
    # finalise-free path: the epilogue adds into these zeroed rows (a cross-replica BatchNorm exchanges ordered partial rows instead)
    req.acc = None if isinstance(bn, FusedSyncBatchNorm2d) else _acc_take(bn.num_features, x.device)
    nn_conv.STATS = req
    try:
        y = conv(x)
    finally:
        nn_conv.STATS = None
    if pool is not None:                                     # (the stem: no residual, the layer's own ReLU; see model/backbone.py _stem)
        return _pooled(bn, y, pool, ("acc", req.acc, req.filled) if req.acc is not None else None)
    if req.acc is not None:
        return bn(y, residual, relu, stats=("acc", req.acc, req.filled), twin=twin)
    return bn(y, residual, relu, stats=(req.part, req.rows) if req.rows else None, twin=twin)


class ConvBN(nn.Sequential):
    """nn.Sequential(conv, bn[, nn.Identity]) -- same indices and state-dict keys -- evaluated through ``conv_bn_act``."""

    def forward(self, x):
        return conv_bn_act(self[0], self[1], x)
