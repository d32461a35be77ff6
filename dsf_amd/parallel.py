"""Data parallelism for the hot path: one process per GPU, per-image sharding, gradient
all-reduce over RCCL/xGMI (``torch.distributed`` backend ``nccl`` == RCCL on ROCm; ``gloo`` on CPU
for tests).  The reference has no distributed code at all (train_render.py:86 pins device 0);
every op on the path is per-sample independent, so the only exchange step is the
gradient average of the trainable backbone parameters.

Design for xGMI (8 GPUs fully connected, 7 links x ~153 GB/s each, no switch): gradients live in
a few large flat buckets (default 32 MiB, ~4 buckets for ResNet-18 2-stage's 128 MB) so that each
collective is bandwidth- not latency-bound and RCCL can spread it over all links; a bucket's
all-reduce is launched asynchronously from the autograd hook of its last-arriving parameter, i.e.
it overlaps with the rest of backward; ``finish()`` waits before the optimizer step.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, force=False):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env (torch.distributed.run).  ``force``: create the process
    group even at world size 1 (a one-rank RCCL communicator: tests/test_gpu_rccl.py drives the collective path through it)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # RCCL on GPUs; DSF_DIST_BACKEND=gloo lets several ranks share one GPU (flow tests on a 1-GPU box)
            backend = os.environ.get("DSF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def _flat(t):
    """1-D view of a dense tensor in MEMORY order (conv weights keep a kernel layout: logical (Co,Ci,KH,KW),
    memory [KH][KW][Ci][Co]); falls back to a logical-order copy for non-dense tensors."""
    if t.is_contiguous():
        return t.view(-1)
    order = sorted(range(t.dim()), key=lambda d: -t.stride(d))
    if t.permute(*order).is_contiguous():
        return t.as_strided((t.numel(),), (1,), t.storage_offset())
    return t.reshape(-1)


def shard_batch(tensors, rank, world):
    """Per-image sharding: rank r takes rows [r*B/world, (r+1)*B/world)."""
    out = []
    for t in tensors:
        B = t.shape[0]
        assert B % world == 0, "global batch must divide evenly (equal shards keep mean losses exact)"
        per = B // world
        out.append(t[rank * per:(rank + 1) * per].contiguous())
    return out


class GradAllReducer:
    """Bucketed, backward-overlapped gradient averaging, IN PLACE.

    Parameters are grouped (reverse registration order ~ the order backward produces their grads) into ~32 MiB buckets, and
    every bucket is a slice of ONE persistent flat buffer in which each parameter owns a 256-byte aligned slot (its gradient
    in the parameter's own memory order) followed by the bucket's presence row.  The buffer is zeroed once per backward pass
    (``begin_backward``: one fill launch, issued by ``nn_conv.grad_pool(..., reducer=self)`` when the pass starts, else by the first hook); the backward-weights kernels of
    this package write a convolution's dW straight into the weight's slot (``grad_slot``: the slot takes the place of the
    per-step gradient pool), autograd adopts that view as ``p.grad``, and the few gradients produced elsewhere (BatchNorm
    affine parameters, linear layers, biases) are copied into their slots by one multi-tensor copy per bucket.  The
    ``post_accumulate_grad`` hook of a bucket's last-arriving gradient launches the asynchronous all-reduce of the bucket --
    no ``torch.cat`` pack and no scaling pass: RCCL averages (``ReduceOp.AVG``; gloo, the CPU test backend, sums and
    ``finish()`` divides) -- overlapping with the rest of backward.  ``finish()`` waits; ``p.grad`` already is the view of
    the reduced slot, so steps run with ``zero_grad(set_to_none=True)`` and the optimizer's pointer table never changes.
    (Rounds 1-5 packed every bucket with ``torch.cat`` + ``div_``: two extra passes over the 128 MB of gradients per step.)
    """

    ALIGN = 64                       # floats: 256-byte slots (rows of float atomics of the backward-weights kernels stay inside their cache lines)

    def __init__(self, params, bucket_bytes=32 << 20, group=None, tail_bucket_bytes=2 << 20, force=False):
        """``force``: hooks, packing and the collective also run at world size 1 (they are skipped there otherwise: a
        one-rank average is the identity) -- the way to exercise RCCL's asynchronous semantics on a one-GPU box."""
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        self.group = group
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []            # lists of params
        self._bucket_of = {}
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
                self._seal(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            # the last bucket holds the earliest layers, whose gradients arrive at the very end of backward: nothing is
            # left to overlap its all-reduce with, so it is cut down to a small tail (the rest goes out one bucket earlier)
            tail, tail_bytes = [], 0
            while cur and tail_bytes + cur[-1].numel() * cur[-1].element_size() <= tail_bucket_bytes:
                tail_bytes += cur[-1].numel() * cur[-1].element_size()
                tail.insert(0, cur.pop())
            if cur:
                rest = sum(p.numel() * p.element_size() for p in cur)
                if self.buckets and rest < bucket_bytes // 4 and self.buckets[-1][0].dtype == cur[0].dtype and \
                        self.buckets[-1][0].device == cur[0].device:
                    for p in cur:                                   # a small remainder rides with the previous bucket
                        self._bucket_of[p] = len(self.buckets) - 1
                    self.buckets[-1].extend(cur)
                else:
                    self._seal(cur)
            if tail:
                self._seal(tail)
        # ---- the persistent store: per bucket [slot of param 0 | slot of param 1 | ... | presence row], slots ALIGN-float aligned
        self._slot = {}              # param -> (bucket, offset inside the bucket, in elements)
        self._len = []               # elements per bucket (slots + presence row)
        for b, plist in enumerate(self.buckets):
            off = 0
            for p in plist:
                self._slot[p] = (b, off)
                off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            self._len.append(off + (len(plist) + self.ALIGN - 1) // self.ALIGN * self.ALIGN)
        self._stores = {}            # (device, dtype) -> flat tensor holding that group's buckets back to back
        self._buf = [None] * len(self.buckets)
        self._begun = False
        self._reset()
        self.enabled = True          # False: hooks, slots and finish() do nothing (single-rank diagnostic steps, graph capture)
        self._hooks = []
        self._present = {}           # (bucket, missing pattern) -> device row, see _launch
        if self.active:
            self._allocate()
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
                p.__dict__["_dsf_hooks_join"] = True     # (nn_conv._side_ok: this hook joins the side stream before it reads)
                p.__dict__["_dsf_grad_slot"] = self      # (nn_conv._grad_out: the weight gradient is written into its slot)
        from . import streams
        streams.DP_EPOCH[0] += 1     # every module's cached "may this level fork under data parallelism" answer is stale now

    def _allocate(self):
        sizes = {}
        for b, plist in enumerate(self.buckets):
            key = (plist[0].device, plist[0].dtype)
            sizes.setdefault(key, []).append(b)
        for key, bs in sizes.items():
            store = torch.zeros(sum(self._len[b] for b in bs), device=key[0], dtype=key[1])
            self._stores[key] = store
            off = 0
            for b in bs:
                self._buf[b] = store[off:off + self._len[b]]
                off += self._len[b]

    def _view(self, p):
        """p's slot as a tensor of p's shape and strides (its memory order)"""
        b, off = self._slot[p]
        return self._buf[b][off:off + p.numel()].as_strided(p.shape, p.stride()) if _dense(p) else \
            self._buf[b][off:off + p.numel()].view(p.shape)

    def grad_slot(self, p):
        """The zeroed slot of ``p`` (flat, in p's memory order) for a kernel that WRITES the gradient, or None: the reducer is
        idle / disabled, the pass has not begun (nothing zeroed the store yet), or p's memory is not dense."""
        if not (self.active and self.enabled and self._begun and _dense(p)):
            return None
        b, off = self._slot[p]
        if self._flat[b] is not None:                    # the bucket already left (a second backward pass before finish())
            return None
        return self._buf[b][off:off + p.numel()]

    def begin_backward(self):
        """Zeroes the store for the backward pass that starts now: ONE fill per (device, dtype).  Called by
        ``nn_conv.grad_pool(.., reducer=self)`` on the stream the pass starts on (everything that writes a slot is ordered behind
        it); a pass that never announced itself is caught by the first hook (the weight gradients launched before that hook went
        to the ordinary pool and are copied into their slots with the bucket's other stragglers)."""
        if not (self.active and self.enabled) or self._begun:
            return
        for b, w in enumerate(self._work):               # a pass that was abandoned before finish(): its collectives first
            if w is not None:
                w.wait()
        self._reset()
        for store in self._stores.values():
            store.zero_()
        self._begun = True

    def detach(self):
        """Removes the hooks and the per-parameter marks (the parameters go back to plain autograd accumulation)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            p.__dict__.pop("_dsf_hooks_join", None)
            p.__dict__.pop("_dsf_grad_slot", None)
        self.active = False
        from . import streams
        streams.DP_EPOCH[0] += 1

    def _seal(self, plist):
        for p in plist:
            self._bucket_of[p] = len(self.buckets)
        self.buckets.append(list(plist))

    def _reset(self):
        self._streams = [set() for _ in self.buckets]     # streams the bucket's gradients arrived on (forked chains: streams.py)
        self._arrived = [0] * len(self.buckets)
        self._flat = [None] * len(self.buckets)           # the bucket's buffer once its all-reduce has been launched
        self._work = [None] * len(self.buckets)
        self._missing = [None] * len(self.buckets)

    def _on_grad(self, p):
        if not self.enabled:
            return
        if not self._begun:
            self.begin_backward()                          # (nothing has been handed a slot yet: zeroing here is still in time)
        b = self._bucket_of[p]
        if p.is_cuda:
            self._streams[b].add(torch.cuda.current_stream(p.device))
        self._arrived[b] += 1
        if self._arrived[b] == len(self.buckets[b]) and self._flat[b] is None:
            self._launch(b)

    def _launch(self, b):
        plist = self.buckets[b]
        if plist and plist[0].is_cuda:
            from . import nn_conv
            nn_conv.join_side_streams()        # weight gradients still being written on the backward-weights stream
            # gradients of the bucket that arrived on OTHER streams (the forked arms of an hourglass level run their backward
            # nodes, AccumulateGrad included, on their own streams): the collective below reads them on this one
            cur = torch.cuda.current_stream(plist[0].device)
            for st in self._streams[b]:
                if st != cur:
                    cur.wait_stream(st)
        buf = self._buf[b]
        missing = [p.grad is None for p in plist]
        # gradients that were not written into their slot (produced by torch's own kernels, by a launch that could not take the
        # slot, or replayed from a HIP graph into its own tensors): one multi-tensor copy, then p.grad IS the slot
        src, dst = [], []
        for p in plist:
            g = p.grad
            if g is None:
                continue
            v = self._view(p)
            if g.data_ptr() != v.data_ptr() or g.stride() != v.stride():
                src.append(g.detach()); dst.append(v)
                p.grad = v
        if src:
            torch._foreach_copy_(dst, src)
        # presence row: one value per parameter, 1 where this rank produced a gradient.  After the reduction it tells which
        # parameters got a gradient on NO rank: those keep ``grad = None`` (as in a single-GPU run, where AdamW then skips
        # them -- no weight decay, no moment decay), see finish().  The row lives on the device: built ONCE per (bucket,
        # pattern of missing gradients) -- the pattern is a property of the step, the same every iteration -- so the hook
        # issues no host-to-device copy
        key = (b, tuple(missing))
        present = self._present.get(key)
        if present is None:
            present = self._present[key] = torch.tensor([0.0 if m else 1.0 for m in missing], dtype=buf.dtype).to(buf.device)
        n_par = len(plist)
        buf[self._len[b] - self._row(b):self._len[b] - self._row(b) + n_par].copy_(present)
        self._flat[b] = buf
        self._missing[b] = missing
        op = dist.ReduceOp.AVG if self._avg() else dist.ReduceOp.SUM
        self._work[b] = dist.all_reduce(buf, op=op, group=self.group, async_op=True)

    def _row(self, b):
        return (len(self.buckets[b]) + self.ALIGN - 1) // self.ALIGN * self.ALIGN

    def _avg(self):
        """RCCL averages inside the collective; gloo (CPU tests, the shared-GPU flow tests) has no AVG: sum, then divide in finish()"""
        if not hasattr(self, "_avg_ok"):
            self._avg_ok = dist.get_backend(self.group) == "nccl"
        return self._avg_ok

    def reduce_now(self):
        """All-reduces EVERY bucket from the gradients as they stand, then ``finish()``: for steps whose backward pass did not
        run the hooks -- train_step.GraphedStep replays forward + backward from a HIP graph (captured with ``enabled = False``)
        and calls this before the optimizer.  Nothing overlaps with the backward pass here; what the graph buys instead is the
        host time of ~700-1500 launches per step (the regime of the small-batch / many-rank runs)."""
        if self.active and self.enabled:
            self._begun = False
            self.begin_backward()                          # (zeros for the parameters that have no gradient on this rank)
            for b in range(len(self.buckets)):
                self._launch(b)
        self.finish()

    def finish(self):
        """Call after backward, before optimizer.step()."""
        if self.active and self.enabled:
            if not self._begun:
                self.begin_backward()                      # a pass in which no managed parameter got a gradient
            for b in range(len(self.buckets)):
                if self._flat[b] is None:                # some parameter of the bucket got no gradient this step
                    self._launch(b)
            for b, plist in enumerate(self.buckets):
                self._work[b].wait()
                if not self._avg():
                    self._flat[b].div_(self.world)
                n_par = len(plist)
                anywhere = None
                if any(self._missing[b]):                # only a rank that lacks a gradient has to look (one small D2H copy)
                    r0 = self._len[b] - self._row(b)
                    anywhere = self._flat[b][r0:r0 + n_par].tolist()
                for i, p in enumerate(plist):
                    if self._missing[b][i]:
                        # no rank produced it: stays None; else: this rank's zeros + the others' contributions
                        p.grad = None if anywhere[i] == 0.0 else self._view(p)
        self._reset()
        self._begun = False


def _dense(t):
    """the tensor's elements fill its storage range without gaps (any permutation of a contiguous layout)"""
    if t.is_contiguous():
        return True
    order = sorted(range(t.dim()), key=lambda d: -t.stride(d))
    return t.permute(*order).is_contiguous()


def all_reduce_mean_pair(total, count, group=None):
    """Exact masked mean under sharding: reduce (sum, count) before dividing."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        both = torch.stack([total, count.to(total.dtype)])
        dist.all_reduce(both, group=group)
        total, count = both[0], both[1]
    return total / (count + 1e-8)


def convert_sync_batchnorm(net, process_group=None):
    """Optional: global-batch BN statistics (per-replica statistics are the default).  Every fused BN(+add+ReLU)
    module becomes a ``nn_norm.FusedSyncBatchNorm2d`` IN PLACE (same parameters, buffers, keys and call signature): its
    statistics cross the ranks as ONE all-reduce of 2C + 1 doubles per layer and pass and feed the same fused HIP apply kernels
    (round 2 fell back to torch.nn.SyncBatchNorm's own kernels and per-layer gathers).  Plain ``torch.nn.BatchNorm*`` modules,
    if any, are converted by torch."""
    from .nn_norm import FusedBatchNorm2d, FusedSyncBatchNorm2d

    def walk(mod):
        for name, child in list(mod.named_children()):
            if isinstance(child, FusedBatchNorm2d):
                if not isinstance(child, FusedSyncBatchNorm2d):
                    child.__class__ = FusedSyncBatchNorm2d
                child.process_group = process_group
            elif isinstance(child, torch.nn.modules.batchnorm._BatchNorm):
                # (torch's converter would also turn OUR modules -- BatchNorm2d subclasses -- into its own: plain ones only)
                setattr(mod, name, torch.nn.SyncBatchNorm.convert_sync_batchnorm(child, process_group))
            else:
                walk(child)
    if isinstance(net, FusedBatchNorm2d):
        net.__class__ = FusedSyncBatchNorm2d
        net.process_group = process_group
        return net
    if isinstance(net, torch.nn.modules.batchnorm._BatchNorm):
        return torch.nn.SyncBatchNorm.convert_sync_batchnorm(net, process_group)
    walk(net)
    return net
