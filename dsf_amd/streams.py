"""Fork / join of independent launch chains over HIP streams.

A chain of small launches (the 8 x 8 ... 2 x 2 levels of an hourglass at 2 ... 32 tiles each, the geometry losses after the MANO
layer) is latency: issued one after the other on one stream it leaves most of the 256 CUs idle.  Where the data flow forks,
the arms are issued on different streams -- ``fork(device)`` hands out branch streams that start behind the current one and
``join()`` orders the current one behind them again -- so that the GPU runs them beside each other; autograd replays every
backward node on the stream of its forward and orders the streams itself, and a captured HIP graph (train_step.GraphedStep)
keeps the fork as independent dependency chains.

Allocator safety: a branch stream only ever works between a fork (``wait_stream(current)``) and the join that follows, so a
block of ITS pool that is freed after the join is re-used behind everything the forking stream had queued by the next fork.
What a branch READS from the forking stream is named in ``branch(slot, *reads)`` and recorded on the branch stream
(``Tensor.record_stream``: a handful of tensors per step -- per layer, as nn_conv._on_side_stream would have needed it, its
event-deferred frees made the caching allocator fall back to hipMalloc): autograd saves such a tensor for a backward node that
runs on the branch stream and drops it as soon as that node has been ISSUED, and without the record its block would go straight
back to the forking stream's pool while the node's kernel may still be reading it.  Gradients that cross streams in the backward
pass are recorded by the autograd engine itself.

Eager steps keep to ONE branch stream (slot 0) beside the current one and the weight-gradient stream: HIP maps streams onto
GPU_MAX_HW_QUEUES = 4 hardware queues, and with five streams config 2 ran whole benchmark runs at 20.2-20.9 ms instead of
18.9-19.4 (2 of 10 runs; 0 of 8 with one branch stream) -- two streams sharing a queue is the likely cause, not an isolated one.  A replayed HIP graph has
its own queue assignment: inside a capture the hourglass levels and the loss chains of config 3 use a stream each (``slot``).
DSF_BRANCHES=0 keeps everything on one stream.  In a multi-rank process group a chain that produces parameter gradients
(``params=<module>``) forks only when every parameter of the module is managed by parallel.GradAllReducer, which notes the
stream each gradient arrives on and orders its bucket pack behind them; any other data-parallel wrapper reads gradients on
streams this module cannot know.
"""
import contextlib
import os

import torch

ENABLED = [os.environ.get("DSF_BRANCHES", "1") == "1"]
SLOTS = int(os.environ.get("DSF_BRANCH_SLOTS", "4"))           # distinct branch streams inside a capture
_STREAMS = {}
DP_EPOCH = [0]                 # bumped whenever a GradAllReducer is attached or detached (parallel.py): invalidates fork()'s per-module cache


def _stream(device, slot):
    # eager: ONE branch stream (see above); inside a stream capture the slots are distinct streams = independent chains of the graph
    slot = slot % SLOTS if torch.cuda.is_current_stream_capturing() else 0
    key = (device.index if device.index is not None else torch.cuda.current_device(), slot)
    s = _STREAMS.get(key)
    if s is None:
        s = _STREAMS[key] = torch.cuda.Stream(device=device)
    return s


class disabled:
    """``with streams.disabled():`` keeps everything inside on one stream (a step whose chains do not pay: measured per step class)"""

    def __enter__(self):
        self.was, ENABLED[0] = ENABLED[0], False
        return self

    def __exit__(self, *a):
        ENABLED[0] = self.was


class fork:
    """``f = fork(x.device); with f.branch(0, x): a = g(x); b = h(x); f.join(); a + b``  (a no-op on the CPU or when switched off)"""

    def __init__(self, device, params=None):
        device = torch.device(device)
        self.on = ENABLED[0] and device.type == "cuda"
        if self.on and params is not None and torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size() > 1:
            cached = params.__dict__.get("_dsf_fork_ok")
            if cached is None or cached[0] != DP_EPOCH[0]:  # decided once per module and per set of attached reducers
                ok = all(bool(p.__dict__.get("_dsf_hooks_join")) for p in params.parameters() if p.requires_grad)
                params.__dict__["_dsf_fork_ok"] = cached = (DP_EPOCH[0], ok)
            self.on = cached[1]
        self.device, self.used = device, []
        self.cur = torch.cuda.current_stream(device) if self.on else None
        self.ev = None

    def mark(self):
        """Fixes the fork POINT here: branches entered later start behind what the current stream holds NOW, not behind what the host
        has queued on it by then -- the caller can issue the current stream's (long) chain first and the branch's afterwards, and the
        two still run beside each other (the host needs ~0.5 ms to enqueue config 2's stage-2 bridge; issued first, the decoder's
        launches waited for the host that long)."""
        if self.on:
            self.ev = torch.cuda.Event()
            self.ev.record(self.cur)
        return self

    def branch(self, slot, *reads):
        if not self.on:
            return contextlib.nullcontext()
        s = _stream(self.device, slot)
        if s not in self.used:
            if self.ev is not None:
                s.wait_event(self.ev)
            else:
                s.wait_stream(self.cur)
            self.used.append(s)
        for t in reads:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(s)
        return torch.cuda.stream(s)

    def join(self):
        for s in self.used:
            self.cur.wait_stream(s)
        self.used = []
