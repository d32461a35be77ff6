"""AdamW with the whole parameter set updated by ONE kernel launch (``dsf_adamw_multi``).

Drop-in for ``torch.optim.AdamW(params, lr, betas, eps, weight_decay)`` as the reference constructs it
(train_render.py:131-139; amsgrad / maximize / capturable are not used there and not supported here): same
hyper-parameter names, ``param_groups``, ``state_dict`` layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter), so
``StepLR`` and checkpoints work unchanged.  GPU fp32 dense parameters only; anything else raises."""
import ctypes
import math

import torch

from . import _lib as L
from ._lib import I, check, stream_ptr

D = ctypes.c_double


RING = 8          # pinned staging buffers of a pointer table: one is reused eight refreshes later

class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}
        from . import nn_conv
        nn_conv.manage_weights(p for g in self.param_groups for p in g["params"])     # this optimizer announces its writes

    def _table(self, slot, plist):
        """static chunk table of one launch + reusable pointer buffers (host pinned, device)"""
        key = tuple(id(p) for p in plist)
        tb = self._tables.get(slot)
        if tb is not None and tb["key"] == key:
            return tb
        dev = plist[0].device
        E = int(L.lib().dsf_adamw_chunk_elems())
        ct, ci = [], []
        for t, p in enumerate(plist):
            n = (p.numel() + E - 1) // E
            ct += [t] * n
            ci += list(range(n))
        tb = {"key": key,
              "chunk_tensor": torch.tensor(ct, dtype=torch.int32, device=dev), "chunk_index": torch.tensor(ci, dtype=torch.int32, device=dev),
              "sizes": torch.tensor([p.numel() for p in plist], dtype=torch.int64, device=dev), "n_chunks": len(ct),
              "rows": None, "ptrs": torch.empty((len(plist), 4), dtype=torch.int64, device=dev),
              # ring of pinned staging buffers for pointer-table refreshes: the copy is asynchronous (the host may run a
              # whole step ahead of the GPU), a buffer is reused only after the copy that read it has completed
              "ring": [torch.empty((len(plist), 4), dtype=torch.int64).pin_memory() for _ in range(RING)],
              "ring_ev": [None] * RING, "ring_i": 0}
        self._tables[slot] = tb
        return tb

    def _prepare(self, gi, group):
        """per-group cache: the parameters with gradients, partitioned by their step count (torch.optim.AdamW keeps one
        step per parameter: a parameter that sat out some steps -- no gradient under zero_grad(set_to_none=True) -- has
        its own bias corrections).  Normally there is ONE partition = one launch; k distinct counts cost k launches."""
        plist = [p for p in group["params"] if p.grad is not None]
        for p in plist:
            if not (p.is_cuda and p.dtype == torch.float32) or p.grad.is_sparse:
                raise RuntimeError("FusedAdamW handles dense fp32 GPU parameters only")
            st = self.state[p]
            if not st:
                st["step"] = torch.tensor(0.0)                           # host scalar, as torch's non-capturable AdamW keeps it
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if st["exp_avg"].stride() != p.stride():                     # e.g. state loaded from a checkpoint: adopt the parameter's layout
                st["exp_avg"] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st["exp_avg"])
                st["exp_avg_sq"] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st["exp_avg_sq"])
        by_step = {}
        for p in plist:
            by_step.setdefault(int(self.state[p]["step"]), []).append(p)
        parts = [{"plist": ps, "step": s, "m": [self.state[p]["exp_avg"] for p in ps],
                  "v": [self.state[p]["exp_avg_sq"] for p in ps]} for s, ps in sorted(by_step.items())]
        return {"ids": tuple(id(p) for p in plist), "parts": parts}

    def _sync_steps(self):
        """the per-parameter `step` entries of the state dict are refreshed from the partition counters on demand"""
        for c in getattr(self, "_cache", {}).values():
            for part in c["parts"]:
                for p in part["plist"]:
                    self.state[p]["step"].fill_(float(part["step"]))

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._cache, self._tables = {}, {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not hasattr(self, "_cache"):
            self._cache = {}
        for gi, group in enumerate(self.param_groups):
            c = self._cache.get(gi)
            if c is None or c["ids"] != tuple(id(p) for p in group["params"] if p.grad is not None):
                if c is not None:
                    self._sync_steps()
                c = self._cache[gi] = self._prepare(gi, group)
            b1, b2 = group["betas"]
            for pi, part in enumerate(c["parts"]):
                plist = part["plist"]
                part["step"] += 1
                step = part["step"]
                rows = []
                for p, m, v in zip(plist, part["m"], part["v"]):
                    g = p.grad
                    if g.stride() != p.stride():                             # walk everything in the parameter's memory order
                        g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                        p.grad = g
                    rows.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()))
                tb = self._table((gi, pi), plist)
                if rows != tb["rows"]:
                    # addresses changed (first step, a different gradient block from the allocator, or the per-step flat
                    # buckets of data-parallel runs): refresh the device table without draining the stream
                    i = tb["ring_i"]
                    # (query first: hipEventSynchronize on an event that completed long ago still waited until the GPU had nearly
                    #  caught up with the host -- 2 ms per step in which the host could have been enqueuing the next forward pass,
                    #  tools/opt_block.py; the query does not)
                    if tb["ring_ev"][i] is not None and not tb["ring_ev"][i].query():
                        tb["ring_ev"][i].synchronize()
                    tb["ring"][i].copy_(torch.tensor(rows, dtype=torch.int64))
                    tb["ptrs"].copy_(tb["ring"][i], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    tb["ring_ev"][i], tb["ring_i"], tb["rows"] = ev, (i + 1) % RING, rows
                vp = lambda t: ctypes.c_void_p(t.data_ptr())
                check(L.lib().dsf_adamw_multi(vp(tb["ptrs"]), vp(tb["sizes"]), vp(tb["chunk_tensor"]), vp(tb["chunk_index"]),
                                              I(tb["n_chunks"]), D(group["lr"]), D(b1), D(b2), D(group["eps"]),
                                              D(group["weight_decay"]), D(1.0 - math.pow(b1, step)), D(1.0 - math.pow(b2, step)),
                                              stream_ptr()), "dsf_adamw_multi")
        from . import nn_conv
        mine = [p for g in self.param_groups for p in g["params"]]
        nn_conv.weights_changed(mine)        # the kernel wrote THESE parameters behind torch's version counters
        nn_conv.refresh_images(mine, owner=self)     # one launch
        return loss
