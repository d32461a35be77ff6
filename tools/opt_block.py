import os, sys, time, types, math, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from dsf_amd import nn_conv, _lib as L
from dsf_amd._lib import I, check, stream_ptr
D = ctypes.c_double
args = types.SimpleNamespace(config=2, batch=0, backbone="", graph=False, no_graph=False, cpu_steps=0)
w = bench.build_workload(args, torch.device("cuda", 0), 0, 1)
step, tgt = w["step"], w["tgt"]
for _ in range(6):
    w["run"]()
torch.cuda.synchronize()
opt = step.opt
acc = {}
def T(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
N = 30
for it in range(N):
    step.forward_backward(tgt)
    t = time.perf_counter()
    for gi, group in enumerate(opt.param_groups):
        c = opt._cache.get(gi)
        ids = tuple(id(p) for p in group["params"] if p.grad is not None)
        assert c is not None and c["ids"] == ids
        t = T("ids", t)
        b1, b2 = group["betas"]
        for pi, part in enumerate(c["parts"]):
            plist = part["plist"]
            part["step"] += 1
            stepn = part["step"]
            rows = []
            for p, m, v in zip(plist, part["m"], part["v"]):
                g = p.grad
                assert g.stride() == p.stride()
                rows.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()))
            t = T("rows", t)
            tb = opt._table((gi, pi), plist)
            t = T("table", t)
            if rows != tb["rows"]:
                i = tb["ring_i"]
                if tb["ring_ev"][i] is not None and not tb["ring_ev"][i].query():
                    acc["ring not ready"] = acc.get("ring not ready", 0) + 1
                    tb["ring_ev"][i].synchronize()
                t = T("ring sync", t)
                tb["ring"][i].copy_(torch.tensor(rows, dtype=torch.int64))
                t = T("host copy", t)
                tb["ptrs"].copy_(tb["ring"][i], non_blocking=True)
                t = T("h2d copy", t)
                ev = torch.cuda.Event()
                ev.record()
                tb["ring_ev"][i], tb["ring_i"], tb["rows"] = ev, (i + 1) % len(tb["ring"]), rows
                t = T("event", t)
            vp = lambda x: ctypes.c_void_p(x.data_ptr())
            check(L.lib().dsf_adamw_multi(vp(tb["ptrs"]), vp(tb["sizes"]), vp(tb["chunk_tensor"]), vp(tb["chunk_index"]),
                                          I(tb["n_chunks"]), D(group["lr"]), D(b1), D(b2), D(group["eps"]),
                                          D(group["weight_decay"]), D(1.0 - math.pow(b1, stepn)), D(1.0 - math.pow(b2, stepn)),
                                          stream_ptr()), "dsf_adamw_multi")
            t = T("adamw launch", t)
    mine = [p for g in opt.param_groups for p in g["params"]]
    nn_conv.weights_changed(mine)
    t = T("weights_changed", t)
    nn_conv.refresh_images(mine, owner=opt)
    t = T("refresh_images", t)
torch.cuda.synchronize()
for k, v in acc.items():
    print("%-18s %8.1f us per step" % (k, v / N * 1e6) if k != "ring not ready" else "ring events not ready at reuse: %d of %d" % (v, N))
