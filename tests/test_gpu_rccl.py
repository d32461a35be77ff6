"""RCCL itself under the data-parallel code, on the ONE GPU of the test box: a process group with backend "nccl" (= RCCL on ROCm)
at world size 1 and a GradAllReducer whose hooks are forced on (``force=True``).  A one-rank all-reduce is the identity, so every
gradient, loss and parameter must be BITWISE what the plain single-process step gives in deterministic mode -- but the code
that runs is the real one: asynchronous collectives on RCCL's own stream, ``work.wait()`` as a stream dependency (not a host
block), buckets packed from autograd hooks in the middle of the backward pass, gradients arriving on forked streams, and
``reduce_now`` behind a HIP-graph replay.  Every other multi-rank test of this suite uses gloo (host-synchronous collectives),
which cannot see an ordering bug between the pack, the collective and the optimizer.  The reference has no counterpart
(/root/reference/train_render.py:86 pins device 0).

The worker runs in a child process (a process group is process-wide state) and leaves ``gpurun_out/rccl_world1.log`` behind:
the backend torch reports, RCCL's version line and the per-check results."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, __REPO__)
import torch
import torch.distributed as dist
from dsf_amd import _lib as L, nn_conv, streams
from dsf_amd.parallel import init_distributed, GradAllReducer
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.model.hourglass import PoseNetMANO
from dsf_amd.train_step import RenderSupervisedStep, MeshLossStep, GraphedStep, synthetic_batch, Config
from dsf_amd.optim import FusedAdamW

out = {}
rank, local, world = init_distributed("nccl", force=True)
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
out["backend"] = dist.get_backend()
try:
    out["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
except Exception as e:
    out["nccl_version"] = "unknown (%s)" % type(e).__name__
L.set_deterministic(True)
render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()


def grads_of(net):
    return [None if p.grad is None else p.grad.detach().clone() for p in net.parameters()]


def same(a, b):
    return len(a) == len(b) and all((x is None and y is None) or (x is not None and y is not None and torch.equal(x, y)) for x, y in zip(a, b))


# ---- 1. eager two-stage ResNet-18 step: hooks pack 1 MB buckets mid-backward, async all-reduce, finish(), FusedAdamW -----------
def build18():
    torch.manual_seed(0)
    net = MANO_OCR_stage("ResNet_stage_18", 21, True).cuda()
    with torch.no_grad():
        for head in (net.mano_regress[2], net.mano_regress_s2[2]):
            head.bias[58] = 1.0
    return net

p, c, cube = synthetic_batch(4, "cuda", seed=3)
net_a, net_b = build18(), build18()
step_a = RenderSupervisedStep(net_a, render, Config)
tgt = step_a.make_targets(p, c, cube)
sync = GradAllReducer(net_b.parameters(), bucket_bytes=1 << 20, force=True)
assert sync.active and len(sync.buckets) > 8 and len(sync._hooks) == len(sync.params)
step_b = RenderSupervisedStep(net_b, render, Config, grad_sync=sync)
ok_g, ok_w, launched = True, True, 0
for it in range(3):
    la, _ = step_a.forward_backward(tgt)
    ga = grads_of(net_a)
    step_a.opt.step()
    lb, _ = step_b.forward_backward(tgt)
    launched += sum(w is not None for w in sync._work)          # collectives already in flight when backward returns
    sync.finish()
    gb = grads_of(net_b)
    step_b.opt.step()
    torch.cuda.synchronize()
    ok_g &= same(ga, gb) and bool(torch.equal(la, lb))
    ok_w &= all(torch.equal(x, y) for x, y in zip(net_a.parameters(), net_b.parameters()))
out["eager_r18_gradients_bitwise"] = ok_g
out["eager_r18_parameters_bitwise_after_3_steps"] = ok_w
out["eager_r18_collectives_launched_from_hooks"] = launched
out["eager_r18_buckets"] = len(sync.buckets)
sync.detach()

# ---- 1b. the default 32 MiB buckets: weight gradients are written INTO the persistent bucket store (no pack) ---------------------
net_c, net_d = build18(), build18()
step_c = RenderSupervisedStep(net_c, render, Config)
sync32 = GradAllReducer(net_d.parameters(), force=True)                       # bucket_bytes = 32 MiB
step_d = RenderSupervisedStep(net_d, render, Config, grad_sync=sync32)
ok32, in_place, same_addr = True, 0, True
addr = None
for it in range(3):
    lc_, _ = step_c.forward_backward(tgt)
    gc_ = grads_of(net_c)
    step_c.opt.step()
    ld_, _ = step_d.forward_backward(tgt)
    # gradients that autograd adopted straight from their bucket slot (written there by the backward-weights kernels)
    in_place = sum(1 for q in sync32.params if q.grad is not None and q.grad.data_ptr() == sync32._view(q).data_ptr())
    sync32.finish()
    gd_ = grads_of(net_d)
    now = [q.grad.data_ptr() for q in sync32.params if q.grad is not None]
    same_addr &= addr is None or addr == now                                   # persistent slots: the optimizer's pointer table never changes
    addr = now
    step_d.opt.step()
    torch.cuda.synchronize()
    ok32 &= same(gc_, gd_) and bool(torch.equal(lc_, ld_)) and all(torch.equal(x, y) for x, y in zip(net_c.parameters(), net_d.parameters()))
out["default_buckets"] = len(sync32.buckets)
out["default_buckets_bitwise_3_steps"] = ok32
out["default_buckets_gradients_written_in_place_before_finish"] = in_place
out["default_buckets_gradient_addresses_persistent"] = same_addr
out["default_buckets_all_gradients_in_store"] = all(q.grad.data_ptr() == sync32._view(q).data_ptr() for q in sync32.params if q.grad is not None)
sync32.detach()

# ---- 1c. FusedSyncBatchNorm2d's collectives on RCCL (forced at world size 1): = the per-replica fused BatchNorm ---------------------
from dsf_amd.parallel import convert_sync_batchnorm
from dsf_amd.nn_norm import FusedSyncBatchNorm2d
net_e, net_f = build18(), build18()
convert_sync_batchnorm(net_f)
FusedSyncBatchNorm2d.force_sync = True
n_sync = sum(isinstance(m, FusedSyncBatchNorm2d) for m in net_f.modules())
step_e = RenderSupervisedStep(net_e, render, Config)
step_f = RenderSupervisedStep(net_f, render, Config, grad_sync=GradAllReducer(net_f.parameters(), force=True))
le_, _ = step_e.forward_backward(tgt)
lf_, _ = step_f.forward_backward(tgt)
step_f.grad_sync.finish()
torch.cuda.synchronize()
num = sum(((a_.grad - b_.grad).double() ** 2).sum() for a_, b_ in zip(net_e.parameters(), net_f.parameters()))
den = sum((a_.grad.double() ** 2).sum() for a_ in net_e.parameters())
out["syncbn_modules"] = n_sync
out["syncbn_loss_rel"] = abs(float(le_) - float(lf_)) / abs(float(le_))
out["syncbn_grad_rel"] = float((num / den) ** 0.5)
out["syncbn_running_stats_rel"] = max(float((a_ - b_).abs().max() / a_.abs().max().clamp_min(1e-12)) for (k_, a_), (_, b_) in
                                      zip(net_e.state_dict().items(), net_f.state_dict().items()) if "running" in k_)
FusedSyncBatchNorm2d.force_sync = False
step_f.grad_sync.detach()

# ---- 1d. two networks, two reducers: a backward pass zeroes ITS reducer's store only (the other's reduced gradients may still be
#          waiting for their optimizer: GAN-style training) ----------------------------------------------------------------------
net_g, net_h = build18(), build18()
sync_gg, sync_hh = GradAllReducer(net_g.parameters(), force=True), GradAllReducer(net_h.parameters(), force=True)
step_g = RenderSupervisedStep(net_g, render, Config, grad_sync=sync_gg)
step_h = RenderSupervisedStep(net_h, render, Config, grad_sync=sync_hh)
step_g.forward_backward(tgt); sync_gg.finish()
held = grads_of(net_g)                                          # reduced, not yet consumed by an optimizer
step_h.forward_backward(tgt); sync_hh.finish()                  # another network's pass in between
out["other_reducers_gradients_survive_a_foreign_pass"] = same(held, grads_of(net_g)) and same(grads_of(net_g), grads_of(net_h))
sync_gg.detach(); sync_hh.detach()

# ---- 2. forked streams (hourglass arms run backward nodes on branch streams) under the reducer, eager --------------------------
def build_hg():
    torch.manual_seed(1)
    return PoseNetMANO(1, 21).cuda()

p6, c6, cube6 = synthetic_batch(6, "cuda", seed=5)
hg_a, hg_b = build_hg(), build_hg()
sa = MeshLossStep(hg_a, render, Config, n_points=512)
tg = sa.make_targets(p6, c6, cube6)
with streams.disabled():
    sa.forward_backward(tg)
    ref = grads_of(hg_a)
sync_h = GradAllReducer(hg_b.parameters(), bucket_bytes=1 << 20, force=True)
sb = MeshLossStep(hg_b, render, Config, n_points=512, grad_sync=sync_h)
ok = True
seen = set()
for it in range(3):
    sb.forward_backward(tg)
    for s_ in sync_h._streams:
        seen |= set(int(x.cuda_stream) for x in s_)
    sync_h.finish()
    torch.cuda.synchronize()
    ok &= same(ref, grads_of(hg_b))
out["forked_hourglass_gradients_bitwise"] = ok
out["forked_hourglass_arrival_streams"] = len(seen)

# ---- 3. reduce_now behind a HIP-graph replay ------------------------------------------------------------------------------------
hg_c, hg_d = build_hg(), build_hg()
sc = MeshLossStep(hg_c, render, Config, n_points=512)
sync_g = GradAllReducer(hg_d.parameters(), bucket_bytes=1 << 20, force=True)
sd = MeshLossStep(hg_d, render, Config, n_points=512, grad_sync=sync_g)
gs = GraphedStep(sd, tg)
ok_l, ok_p = True, True
for it in range(3):
    lc, _ = sc(tg)
    ld, _ = gs(tg)
    torch.cuda.synchronize()
    ok_l &= bool(torch.equal(lc, ld))
    ok_p &= all(torch.equal(x, y) for x, y in zip(hg_c.parameters(), hg_d.parameters()))
out["graphed_step_loss_bitwise"] = ok_l
out["graphed_step_parameters_bitwise_after_3_steps"] = ok_p
out["hooks_left_enabled"] = bool(sync_g.enabled)

# ---- 4. detach(): the parameters go back to plain accumulation and the fork cache is invalidated ---------------------------------
e0 = streams.DP_EPOCH[0]
sync_g.detach()
out["detach_removed_marks"] = not any("_dsf_hooks_join" in p_.__dict__ for p_ in hg_d.parameters()) and streams.DP_EPOCH[0] == e0 + 1

dist.barrier()
dist.destroy_process_group()
print("RCCL_RESULT " + json.dumps(out))
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gradient_all_reduce_on_rccl_at_world_size_one(tmp_path):
    import json
    if os.environ.get("DSF_CONV_MATH", "x6") != "x6":
        pytest.skip("DSF_CONV_MATH=f32: the worker's launch counts are those of the split kernels")
    nccl_log = tmp_path / "rccl_debug.log"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT", NCCL_DEBUG_FILE=str(nccl_log), HSA_ENABLE_IPC_MODE_LEGACY="0",
               DSF_DETERMINISTIC="1")
    r = subprocess.run([sys.executable, "-c", WORKER.replace("__REPO__", repr(REPO))], env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_RESULT ")]
    assert line, tail
    res = json.loads(line[-1][len("RCCL_RESULT "):])
    rccl_lines = [l for l in (nccl_log.read_text().splitlines() if nccl_log.exists() else []) if "RCCL" in l or "NCCL version" in l or "comm 0x" in l][:12]
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "rccl_world1.log"), "w") as fh:
        fh.write("tests/test_gpu_rccl.py: GradAllReducer(force=True) on torch.distributed backend nccl (= RCCL), world size 1, one MI355X\n")
        fh.write("distributed.backend: %s   torch.cuda.nccl.version(): %s\n" % (res["backend"], res["nccl_version"]))
        for k, v in res.items():
            fh.write("  %-48s %s\n" % (k, v))
        fh.write("RCCL's own log (NCCL_DEBUG=INFO, INIT):\n" + "\n".join("  " + l for l in rccl_lines) + "\n")
    # bench.py's reader of RCCL's own log, on what RCCL really wrote here (round 4: "an untested guess")
    sys.path.insert(0, REPO)
    import bench
    facts = bench.rccl_summary(str(nccl_log))["log"]
    assert facts and facts["lines"] > 0 and facts["nranks_reported"] == [1], facts
    assert facts["version"] and facts["version"][0].isdigit(), facts
    assert res["backend"] == "nccl"
    assert res["eager_r18_buckets"] > 8 and res["eager_r18_collectives_launched_from_hooks"] >= 3 * (res["eager_r18_buckets"] - 1), res
    assert res["eager_r18_gradients_bitwise"] and res["eager_r18_parameters_bitwise_after_3_steps"], res
    assert res["forked_hourglass_gradients_bitwise"], res
    assert res["forked_hourglass_arrival_streams"] >= 2, res          # gradients really arrived on more than one stream
    assert res["graphed_step_loss_bitwise"] and res["graphed_step_parameters_bitwise_after_3_steps"] and res["hooks_left_enabled"], res
    assert res["detach_removed_marks"], res
    # round 6: the in-place buckets at their default size, and the cross-replica BatchNorm's collectives on RCCL
    assert res["default_buckets"] <= 6 and res["default_buckets_bitwise_3_steps"], res
    assert res["default_buckets_gradients_written_in_place_before_finish"] >= 30, res       # the convolution weights' dW
    assert res["default_buckets_gradient_addresses_persistent"] and res["default_buckets_all_gradients_in_store"], res
    assert res["syncbn_modules"] >= 40 and res["syncbn_loss_rel"] < 1e-5 and res["syncbn_running_stats_rel"] < 1e-5, res
    # (same forward to the bit; the backward sums are folded in another order -- fp32 rounding through ~40 BatchNorm layers at B = 4: the
    #  bar of tests/test_gpu_determinism.py's cross-path comparisons)
    assert res["syncbn_grad_rel"] < 2e-2, res
    assert res["other_reducers_gradients_survive_a_foreign_pass"], res
