"""``latest.pth`` / ``best.pth`` in the reference trainer's format (train_render.py:117-145 load, :282-308 save), so that a run
can move between the two code bases in either direction.

A checkpoint is ``{"model": net.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch}``.  What the reference does
with one, mirrored here:

* ``load_model`` (resume, :118-134): every key of ``checkpoint["model"]`` that the network has is taken, keys it does not have
  are dropped, keys the checkpoint lacks keep the network's current values; ``start_epoch = checkpoint["epoch"] + 1``; the
  optimizer state in the file is NOT loaded (the reference only moves whatever state the fresh optimizer already holds to
  the GPU) -- ``load_optimizer=True`` is this package's extension.
* ``finetune_dir`` (:137-144): the same key filter, epoch untouched (``resume=False``).
* after every epoch ``latest.pth`` is written; ``best.pth`` when the test error is <= the best so far (:282-308).

The HIP convolutions keep their parameters' LOGICAL shape (Co, Ci, KH, KW) and state-dict keys; only the memory order differs
(``nn_conv.kernel_layout_``), which ``state_dict`` / ``load_state_dict`` carry as strides, so files written here load in the
reference and the other way round (tests/test_library_abi.py::test_checkpoint_round_trip)."""
import os

import torch


def filter_state(model_state, net):
    """the reference's ``{k: v for k, v in checkpoint.items() if k in model_dict}`` merged over the network's own state;
    -> (merged state dict, keys taken, keys of the file dropped, keys of the network the file lacks)"""
    own = net.state_dict()
    taken = {k: v for k, v in model_state.items() if k in own}
    dropped = [k for k in model_state if k not in own]
    missing = [k for k in own if k not in model_state]
    merged = dict(own)
    merged.update(taken)
    return merged, list(taken), dropped, missing


def load_checkpoint(path, net, optimizer=None, resume=True, load_optimizer=False, map_location="cpu", trusted=False):
    """Loads ``path`` into ``net`` with the reference's key filter.  -> ``start_epoch`` (``checkpoint["epoch"] + 1`` when
    resuming, 0 for a fine-tune start).  ``load_optimizer``: also restore the optimizer state (the reference never does).
    The reference's files hold tensors, ints and the optimizer's param_groups only, so they load with ``weights_only=True``;
    ``trusted=True`` falls back to the full unpickler (arbitrary code execution: only for files you wrote yourself)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=not trusted)
    merged, _, _, _ = filter_state(ckpt["model"], net)
    net.load_state_dict(merged)
    from . import nn_conv
    nn_conv.weights_changed()                     # nothing keyed on the old parameter contents (split weight images) may survive
    if optimizer is not None:
        if load_optimizer and "optimizer" in ckpt:
            optimizer.load_state_dict(ckpt["optimizer"])
        dev = next(net.parameters()).device
        for state in optimizer.state.values():     # (:131-134) optimizer state follows the network's device
            for k, v in state.items():
                if torch.is_tensor(v) and v.device != dev and v.dim() > 0:
                    state[k] = v.to(dev)
    return int(ckpt["epoch"]) + 1 if resume else 0


def save_checkpoint(path, net, optimizer, epoch):
    """``torch.save({"model", "optimizer", "epoch"}, path)`` (:282-296), written through a temporary file so that an
    interrupted save cannot leave a truncated ``latest.pth``."""
    tmp = "%s.%d.tmp" % (path, os.getpid())             # data-parallel ranks / two Checkpointers on one path do not share it
    torch.save({"model": net.state_dict(), "optimizer": optimizer.state_dict(), "epoch": int(epoch)}, tmp)
    os.replace(tmp, path)


class Checkpointer:
    """End-of-epoch bookkeeping of ``Trainer.train`` (:282-308): ``latest.pth`` always, ``best.pth`` when the test error does
    not exceed the minimum seen (``<=``, as the reference compares)."""

    def __init__(self, model_dir, net, optimizer, min_error=100.0):
        self.dir, self.net, self.opt, self.min_error = model_dir, net, optimizer, float(min_error)
        os.makedirs(model_dir, exist_ok=True)

    def end_of_epoch(self, epoch, test_error=None):
        """-> True when ``best.pth`` was (re)written"""
        save_checkpoint(os.path.join(self.dir, "latest.pth"), self.net, self.opt, epoch)
        if test_error is not None and test_error <= self.min_error:
            self.min_error = float(test_error)
            save_checkpoint(os.path.join(self.dir, "best.pth"), self.net, self.opt, epoch)
            return True
        return False
