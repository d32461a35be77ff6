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
    """Bucketed, backward-overlapped gradient averaging.

    Parameters are grouped (reverse registration order ~ the order backward produces their grads) into
    ~32 MiB buckets.  ``post_accumulate_grad`` hooks count arrivals; when the last gradient of a bucket
    has been produced the bucket is packed with ONE ``torch.cat`` launch, pre-scaled by 1/world and
    all-reduced asynchronously, overlapping with the rest of backward.  ``finish()`` waits and rebinds
    every ``p.grad`` to its slice of the reduced bucket (a view, no copy), so steps can run with
    ``zero_grad(set_to_none=True)``: no per-parameter fill or accumulate kernels (~500 launches per step
    for ResNet-18 two-stage).
    """

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
            if cur and (cur_bytes + nbytes > bucket_bytes or cur[0].dtype != p.dtype):
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
                if self.buckets and rest < bucket_bytes // 4 and self.buckets[-1][0].dtype == cur[0].dtype:
                    for p in cur:                                   # a small remainder rides with the previous bucket
                        self._bucket_of[p] = len(self.buckets) - 1
                    self.buckets[-1].extend(cur)
                else:
                    self._seal(cur)
            if tail:
                self._seal(tail)
        self._reset()
        self.enabled = True          # False: hooks and finish() do nothing (single-rank diagnostic steps)
        self._hooks = []
        self._present = {}           # (bucket, missing pattern) -> device row, see _launch
        if self.active:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
                p.__dict__["_dsf_hooks_join"] = True     # (nn_conv._side_ok: this hook joins the side stream before it reads)
        from . import streams
        streams.DP_EPOCH[0] += 1     # every module's cached "may this level fork under data parallelism" answer is stale now

    def detach(self):
        """Removes the hooks and the per-parameter marks (the parameters go back to plain autograd accumulation)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            p.__dict__.pop("_dsf_hooks_join", None)
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
        self._flat = [None] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._missing = [None] * len(self.buckets)

    def _on_grad(self, p):
        if not self.enabled:
            return
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
            # nodes, AccumulateGrad included, on their own streams): the pack below reads them on this one
            cur = torch.cuda.current_stream(plist[0].device)
            for st in self._streams[b]:
                if st != cur:
                    cur.wait_stream(st)
        missing = [p.grad is None for p in plist]
        parts = [_flat(p.grad if p.grad is not None and p.grad.stride() == p.stride() else
                       (torch.zeros_like(p) if p.grad is None else torch.empty_like(p).copy_(p.grad))) for p in plist]
        # presence row: one float per parameter, 1 where this rank produced a gradient.  After the sum it tells which
        # parameters got a gradient on NO rank: those keep ``grad = None`` (as in a single-GPU run, where AdamW then skips
        # them -- no weight decay, no moment decay), see finish().  The row lives on the device: built ONCE per (bucket,
        # pattern of missing gradients) -- the pattern is a property of the step, the same every iteration -- so the hook
        # issues no host-to-device copy (round 4 built it from pageable memory per bucket per step: a host-blocking copy in
        # the middle of the backward pass, which gloo hides and an overlapped RCCL collective does not)
        key = (b, tuple(missing))
        present = self._present.get(key)
        if present is None:
            present = self._present[key] = torch.tensor([0.0 if m else float(self.world) for m in missing], dtype=parts[0].dtype).to(parts[0].device)
        flat = torch.cat(parts + [present])
        flat.div_(self.world)
        self._flat[b] = flat
        self._missing[b] = missing
        self._work[b] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def reduce_now(self):
        """Packs and all-reduces EVERY bucket from the gradients as they stand, then ``finish()``: for steps whose backward
        pass did not run the hooks -- train_step.GraphedStep replays forward + backward from a HIP graph (captured with
        ``enabled = False``) and calls this before the optimizer.  Nothing overlaps with the backward pass here; what the graph
        buys instead is the host time of ~700-1500 launches per step (the regime of the small-batch / many-rank runs)."""
        if self.active and self.enabled:
            self._reset()
            for b in range(len(self.buckets)):
                self._launch(b)
        self.finish()

    def finish(self):
        """Call after backward, before optimizer.step()."""
        if self.active and self.enabled:
            for b in range(len(self.buckets)):
                if self._flat[b] is None:                # some parameter of the bucket got no gradient this step
                    self._launch(b)
            for b, plist in enumerate(self.buckets):
                self._work[b].wait()
                n_par = len(plist)
                anywhere = None
                if any(self._missing[b]):                # only a rank that lacks a gradient has to look (one small D2H copy)
                    anywhere = self._flat[b][-n_par:].tolist()
                off = 0
                for i, p in enumerate(plist):
                    n = p.numel()
                    if anywhere is not None and self._missing[b][i] and anywhere[i] == 0.0:
                        p.grad = None                    # no rank produced it
                    else:
                        p.grad = self._flat[b][off:off + n].as_strided(p.shape, p.stride())  # same (possibly kernel) layout as p
                    off += n
        self._reset()


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
