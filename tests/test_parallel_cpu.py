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
    # reduce_now(): the hooks stay off during the backward pass (as under a HIP-graph replay, train_step.GraphedStep), the
    # buckets are packed and reduced afterwards -- same averaged gradients as the hook-driven path
    sync.enabled = False
    net.zero_grad(set_to_none=True)
    ((net(xs) - ys) ** 2).mean().backward()
    sync.enabled = True
    sync.reduce_now()
    for got, p in zip(out[1], net.parameters()):
        assert torch.allclose(got, p.grad, atol=1e-6)
    sync.enabled = False
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


def _uneven_net():
    """parameter sizes chosen so that no bucket boundary falls on a layer boundary: 4 B .. 37 KB tensors, a frozen parameter in
    the middle, a tensor larger than a whole bucket, and a small tail"""
    torch.manual_seed(5)
    net = nn.Sequential(nn.Conv2d(1, 3, 3, padding=1), nn.ReLU(), nn.Conv2d(3, 32, 5, padding=2), nn.ReLU(),
                        nn.Conv2d(32, 32, 3, padding=1), nn.ReLU(), nn.AdaptiveAvgPool2d(1), nn.Flatten(),
                        nn.Linear(32, 7), nn.ReLU(), nn.Linear(7, 1, bias=True))
    net[2].bias.requires_grad_(False)
    return net


def _worker4(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from dsf_amd.parallel import init_distributed, GradAllReducer, shard_batch
    init_distributed("gloo")
    net = _uneven_net()
    params = [p for p in net.parameters() if p.requires_grad]
    sync = GradAllReducer(params, bucket_bytes=6000, tail_bucket_bytes=200)
    sizes = [sum(p.numel() * 4 for p in b) for b in sync.buckets]
    g = torch.Generator().manual_seed(9)
    x, y = torch.randn(12, 1, 8, 8, generator=g), torch.randn(12, 1, generator=g)
    xs, ys = shard_batch([x, y], rank, world)                            # 3 images per rank
    out = []
    for it in range(3):
        net.zero_grad(set_to_none=True)
        ((net(xs) - ys) ** 2).mean().backward()
        sync.finish()
        out.append([p.grad.clone().numpy() for p in params])
    if rank == 3:
        q.put((out, sizes, [len(b) for b in sync.buckets]))
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_uneven_buckets_equal_the_full_batch():
    """world size 4 (gloo), buckets that split the parameter list unevenly (one tensor larger than a bucket, a frozen parameter
    between trainable ones, a small tail bucket): every rank's averaged gradient = the gradient of the whole batch, step after
    step."""
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    grads, sizes, counts = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(sizes) >= 4 and max(sizes) > 6000 and min(sizes) <= 200 and len(set(counts)) > 1, (sizes, counts)
    net = _uneven_net()
    g = torch.Generator().manual_seed(9)
    x, y = torch.randn(12, 1, 8, 8, generator=g), torch.randn(12, 1, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    ref = [p.grad for p in net.parameters() if p.requires_grad]
    for it in range(3):
        for got, want in zip(grads[it], ref):
            assert torch.allclose(torch.tensor(got), want, atol=1e-6, rtol=1e-5)


def test_single_process_reducer_is_a_noop():
    from dsf_amd.parallel import GradAllReducer
    net = _net()
    sync = GradAllReducer(net.parameters())
    net(torch.randn(2, 1, 16, 16)).sum().backward()
    sync.finish()
    assert all(p.grad is not None for p in net.parameters())


def test_bench_refuses_a_rank_count_other_than_gpus():
    """`bench.py --gpus 2` inside a 1-rank launch must fail (exit 2), not measure one GPU and label it n_gpus 1."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 2" in r.stderr and r.stdout.strip() == ""


def test_bench_launches_its_own_ranks(monkeypatch):
    """Bare `python bench.py --gpus 4` (no WORLD_SIZE): one torch.distributed.run child with 4 ranks on 127.0.0.1, started
    before any GPU call, its exit code passed on."""
    import importlib
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_relay(cmd, env):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(bench, "relay_one_line", fake_relay)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_relays_exactly_the_json_line(capfd):
    """The self-launching parent passes on ONE stdout line (the ranks' JSON); a backend banner on the child's stdout goes to
    stderr, and the child's exit code is returned."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    code = 'print("[Gloo] Rank 0 is connected to 1 peer ranks"); print(\'{"metric": "x", "value": 1}\'); print("tail"); raise SystemExit(3)'
    rc = bench.relay_one_line([sys.executable, "-c", code], dict(os.environ))
    out, err = capfd.readouterr()
    assert rc == 3 and out == '{"metric": "x", "value": 1}\n' and "[Gloo]" in err and "tail" in err


def test_rccl_log_summary_parses_what_it_can(tmp_path):
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    f = tmp_path / "rccl.log"
    f.write_text("host:1:1 [0] NCCL INFO NCCL version 2.22.3+hip\n"
                 "host:1:1 [0] NCCL INFO comm 0x1 rank 0 nranks 8 cudaDev 0 busId 5000 - Init START\n"
                 "host:1:1 [0] NCCL INFO Ring 00 : 0 1 2 3 4 5 6 7\nhost:1:1 [0] NCCL INFO Ring 01 : 0 7 6 5 4 3 2 1\n"
                 "host:1:1 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1\n"
                 "host:1:1 [0] NCCL INFO 16 coll channels, 16 collnet channels, 0 nvls channels, 16 p2p channels\n")
    s = bench.rccl_summary(str(f))["log"]
    assert s["ring_lines"] == 2 and s["tree_lines"] == 1 and s["channels"] == 16 and s["nranks_reported"] == [8]
    assert bench.rccl_summary(str(tmp_path / "absent.log")) == {"log": None}


def test_convert_sync_batchnorm_keeps_the_fused_modules():
    """parallel.convert_sync_batchnorm: fused BatchNorm modules become FusedSyncBatchNorm2d IN PLACE (class swap: parameters,
    buffers, keys, call signature kept); plain torch BatchNorm modules go through torch's converter; nothing else changes."""
    from dsf_amd.parallel import convert_sync_batchnorm
    from dsf_amd import nn_norm
    net = nn.ModuleList([nn.Conv2d(3, 4, 1), nn_norm.FusedBatchNorm2d(4, fuse_relu=True), nn.Sequential(nn.BatchNorm2d(4), nn_norm.FusedBatchNorm2d(8))])
    keys = list(net.state_dict().keys())
    fused = net[1]
    out = convert_sync_batchnorm(net)
    assert out is net and net[1] is fused and type(net[1]) is nn_norm.FusedSyncBatchNorm2d and net[1].fuse_relu
    assert type(net[2][0]) is nn.SyncBatchNorm and type(net[2][1]) is nn_norm.FusedSyncBatchNorm2d
    assert list(net.state_dict().keys()) == keys
    assert type(convert_sync_batchnorm(nn_norm.FusedBatchNorm2d(4))) is nn_norm.FusedSyncBatchNorm2d



def test_rccl_log_reader_on_lines_rccl_wrote(tmp_path):
    """bench.rccl_summary on the init lines RCCL 2.26.6 itself wrote on an MI355X box (profiles/r05_rccl_world1.log, round 5) plus
    the graph lines of a multi-rank communicator in RCCL's wording: ranks, channels, rings / trees, version."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    log = tmp_path / "rccl.log"
    log.write_text("\n".join([
        "runc:293:553 [0] NCCL INFO RCCL version : 2.26.6-HEAD:64f48b6",
        "runc:293:555 [0] NCCL INFO ncclCommInitRankConfig_impl comm 0x7e795ed6b370 rank 0 nranks 8 cudaDev 0 nvmlDev 0 busId a4000 commId 0xf88f353abe2ebc93 - Init START",
        "runc:293:555 [0] NCCL INFO comm 0x7e795ed6b370 rank 0 nRanks 8 nNodes 1 localRanks 8 localRank 0 MNNVL 0",
        "runc:293:555 [0] NCCL INFO Ring 00 : 7 -> 0 -> 1 comm 0x7e795ed6b370 nRanks 08 busId a4000",
        "runc:293:555 [0] NCCL INFO Ring 01 : 7 -> 0 -> 1 comm 0x7e795ed6b370 nRanks 08 busId a4000",
        "runc:293:555 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1 [1] 1/-1/-1->0->-1 comm 0x7e795ed6b370 nRanks 08 busId a4000",
        "runc:293:555 [0] NCCL INFO 32 coll channels, 32 collnet channels, 0 nvls channels, 32 p2p channels, 2 p2p channels per peer",
        "runc:293:555 [0] NCCL INFO ncclCommInitRankConfig_impl comm 0x7e795ed6b370 rank 0 nranks 8 cudaDev 0 nvmlDev 0 busId a4000 commId 0xf88f353abe2ebc93 - Init COMPLETE"]))
    f = bench.rccl_summary(str(log))["log"]
    assert f == {"lines": 8, "ring_lines": 2, "tree_lines": 1, "channels": 32, "nranks_reported": [8], "version": "2.26.6-HEAD:64f48b6"}
    assert bench.rccl_summary(str(tmp_path / "absent.log")) == {"log": None}
