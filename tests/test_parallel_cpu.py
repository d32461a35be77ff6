"""N>1 path on CPU: world_size-2 gloo processes exercise the bucketed, backward-overlapped
gradient all-reducer, per-image sharding and the exact masked mean."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(1, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(),
                         nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, 62))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from dsf_amd.parallel import init_distributed, GradAllReducer, shard_batch, all_reduce_mean_pair
    r, _, w = init_distributed("gloo")
    assert (r, w) == (rank, world)
    net = _net()
    sync = GradAllReducer(net.parameters(), bucket_bytes=1024)        # tiny buckets -> several collectives
    assert len(sync.buckets) > 2
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 1, 16, 16, generator=g)
    y = torch.randn(8, 62, generator=g)
    xs, ys = shard_batch([x, y], rank, world)
    out = []
    for it in range(2):                                                # twice: bucket state must reset
        net.zero_grad(set_to_none=True)
        loss = ((net(xs) - ys) ** 2).mean()
        loss.backward()
        sync.finish()
        out.append([p.grad.clone() for p in net.parameters()])
    # a head that no rank uses keeps grad None (a single-GPU run leaves it None too: AdamW must skip it); a head only
    # rank 1 uses arrives averaged on both ranks
    sync.enabled = False
    torch.manual_seed(3)
    idle, lone = nn.Linear(4, 2), nn.Linear(4, 2)
    sync2 = GradAllReducer(list(net.parameters()) + list(idle.parameters()) + list(lone.parameters()), bucket_bytes=1024)
    net.zero_grad(set_to_none=True)
    loss = ((net(xs) - ys) ** 2).mean()
    if rank == 1:
        loss = loss + lone(torch.ones(1, 4)).sum()
    loss.backward()
    sync2.finish()
    assert all(p.grad is None for p in idle.parameters()), "unused parameters must keep grad=None under DP"
    assert all(p.grad is not None for p in lone.parameters())
    assert torch.allclose(lone.bias.grad, torch.full((2,), 0.5)) and torch.allclose(lone.weight.grad, torch.full((2, 4), 0.5))
    for got, p in zip(out[1], net.parameters()):
        assert torch.allclose(got, p.grad, atol=1e-6)
    for h in sync2._hooks:
        h.remove()
    # exact masked mean across shards
    vals = torch.arange(4.0) + 4 * rank
    mask = (vals % 3 == 0).float()
    mm = all_reduce_mean_pair((vals * mask).sum(), mask.sum())
    if rank == 0:
        q.put(([[t.numpy() for t in o] for o in out], float(mm)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    grads, mm = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    net = _net()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 1, 16, 16, generator=g)
    y = torch.randn(8, 62, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    for it in range(2):
        for got, p in zip(grads[it], net.parameters()):
            assert torch.allclose(torch.tensor(got), p.grad, atol=1e-6, rtol=1e-5)
    assert abs(mm - (0 + 3 + 6) / 3) < 1e-5          # values 0..7, multiples of 3 -> mean 3


def test_single_process_reducer_is_a_noop():
    from dsf_amd.parallel import GradAllReducer
    net = _net()
    sync = GradAllReducer(net.parameters())
    net(torch.randn(2, 1, 16, 16)).sum().backward()
    sync.finish()
    assert all(p.grad is not None for p in net.parameters())
