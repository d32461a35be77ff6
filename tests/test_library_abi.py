"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/dsf_hip.h declares; the product path refuses to run without a GPU (no fallback)."""
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(REPO, "include", "dsf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dsf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from dsf_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _header_symbols()
    assert len(declared) >= 28
    for s in declared:
        assert hasattr(lib, s), "libdsf_hip.so lacks %s" % s
    assert sorted(_lib.SYMBOLS) == declared
    lib.dsf_abi_version.restype = ctypes.c_int
    assert lib.dsf_abi_version() == _lib.EXPECTED_ABI == 5
    lib.dsf_status_string.restype = ctypes.c_char_p
    assert lib.dsf_status_string(2) == b"unsupported configuration"


def test_product_path_has_no_cpu_fallback():
    from dsf_amd.render_model.mano_layer import MANO_SMPL
    from dsf_amd.metric.meshLoss import ICPLoss
    from dsf_amd.util.generateFeature import GFM
    m = MANO_SMPL("synthetic", "nyu")
    with pytest.raises(RuntimeError):
        m.get_mano_vertices(torch.zeros(1, 3), torch.zeros(1, 45), torch.zeros(1, 10), torch.ones(1, 4))
    with pytest.raises(RuntimeError):
        ICPLoss(torch.zeros(1, 779, 3), torch.zeros(1, 8, 3), m.faces)
    with pytest.raises(RuntimeError):
        GFM().joint2offset(torch.zeros(1, 21, 3), torch.zeros(1, 1, 128, 128), 0.8, 64)
    from dsf_amd.metric.losses import SmoothL1Loss
    from dsf_amd.nn_conv import Conv2d
    from dsf_amd.nn_norm import FusedBatchNorm2d
    with pytest.raises(RuntimeError):
        SmoothL1Loss()(torch.zeros(2, 3), torch.zeros(2, 3))
    with pytest.raises(RuntimeError):
        Conv2d(4, 4, 3)(torch.zeros(1, 4, 8, 8))
    with pytest.raises(RuntimeError):
        FusedBatchNorm2d(4)(torch.zeros(1, 4, 8, 8))


def test_product_never_imports_the_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "dsf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                txt = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle/" in txt and f.endswith(".sh"):
                    bad.append(f)
    assert not bad, bad


def test_model_buffers_match_reference(golden):
    import numpy as np
    from dsf_amd.render_model.mano_layer import MANO_SMPL
    m = MANO_SMPL("synthetic", "nyu")
    assert np.array_equal(m.faces.numpy().astype(np.int32), golden["mano_faces"])
    assert [f.shape[0] for f in m.joint_faces] == golden["mano_joint_faces_len"].tolist()
    assert np.array_equal(torch.cat(m.joint_faces).numpy().astype(np.int32), golden["mano_joint_faces_cat"])
    assert np.array_equal(torch.cat(m.finger_faces).numpy().astype(np.int32), golden["mano_finger_faces_cat"])
    assert np.array_equal(m.mask.numpy().astype(np.uint8), golden["mano_coll_mask"])
    assert np.array_equal(m.parents, golden["mano_parents"])
    assert m.transfer == [18, 8, 19, 11, 17, 5, 16, 2, 20, 15, 14, 0]


def test_library_is_capturable_no_memset_nodes():
    """Every launcher must be capturable into a HIP graph (train_step.GraphedStep); a captured hipMemsetAsync does not
    replay correctly on ROCm 7.2 (tools/graph_memset.py), so csrc/ zero-fills with a kernel (dsf_zero_async), and nothing
    there synchronises or copies through host memory."""
    import glob
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dsf_amd", "csrc")
    banned = re.compile(r"\b(hipMemsetAsync|hipMemset|hipMemcpyAsync|hipMemcpy|hipStreamSynchronize|hipDeviceSynchronize|hipMalloc|hipFree)\s*\(")
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        for i, line in enumerate(open(f), 1):
            code = line.split("//")[0]
            assert not banned.search(code), "%s:%d %s" % (os.path.basename(f), i, line.strip())


def test_torch_library_ops_are_registered_with_fake_kernels():
    """dsf_amd.torch_ops (torch.library registration of the pytorch3d._C boundary): schemas, shape propagation
    through FakeTensorMode without touching a device, and no CPU kernel behind them."""
    import pytest
    import torch
    import dsf_amd.torch_ops  # noqa: F401  (registers)
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name in ("rasterize_meshes", "rasterize_meshes_backward", "point_face_dist_forward", "point_face_dist_backward"):
        assert hasattr(torch.ops.dsf, name)
    assert "image_size" in str(torch.ops.dsf.rasterize_meshes.default._schema)
    with FakeTensorMode():
        fv = torch.empty(2 * 1554, 3, 3, device="cuda")
        first = torch.empty(2, dtype=torch.int64, device="cuda")
        p2f, zbuf, bary, dists = torch.ops.dsf.rasterize_meshes(fv, first, first, 640)
        assert p2f.shape == (2, 640, 640, 1) and p2f.dtype == torch.int64 and bary.shape == (2, 640, 640, 1, 3)
        assert torch.ops.dsf.rasterize_meshes_backward(fv, p2f, zbuf).shape == fv.shape
        d, i = torch.ops.dsf.point_face_dist_forward(torch.empty(4096, 3, device="cuda"), first, fv, first, 2048)
        assert d.shape == (4096,) and i.dtype == torch.int64
        gp, gt = torch.ops.dsf.point_face_dist_backward(torch.empty(4096, 3, device="cuda"), fv, i, d)
        assert gp.shape == (4096, 3) and gt.shape == fv.shape
    with pytest.raises((NotImplementedError, RuntimeError)):             # CPU tensors: no kernel registered, no fallback
        torch.ops.dsf.point_face_dist_forward(torch.zeros(4, 3), torch.zeros(1, dtype=torch.int64), torch.zeros(2, 3, 3),
                                              torch.zeros(1, dtype=torch.int64), 4)


def test_side_stream_capability_guard(monkeypatch):
    """The weight-gradient side stream rests on private torch interfaces (graph task id, the engine's end-of-pass callback, the
    tensor hook dictionaries): nn_conv probes them once and keeps everything on one stream when one is missing."""
    from dsf_amd import nn_conv
    assert nn_conv._side_api_ok() and nn_conv.SIDE_API                      # this torch has them all
    monkeypatch.delattr(torch._C, "_current_graph_task_id")
    assert not nn_conv._side_api_ok()
    monkeypatch.undo()
    monkeypatch.setattr(torch._C, "_current_graph_task_id", lambda: 0)      # present but not -1 outside a backward pass: unknown semantics
    assert not nn_conv._side_api_ok()
    monkeypatch.undo()
    # with the interfaces reported missing no weight qualifies, whatever else holds
    w = torch.nn.Parameter(torch.zeros(8, 8, 3, 3).permute(2, 3, 1, 0).contiguous().permute(3, 2, 0, 1))
    assert w.permute(2, 3, 1, 0).is_contiguous()
    monkeypatch.setattr(nn_conv, "WRW_STREAM", [False])
    assert not nn_conv._side_ok(w)
    monkeypatch.setattr(nn_conv, "WRW_STREAM", [True])
    assert nn_conv._side_ok(w)
    assert not nn_conv._side_ok(torch.nn.Parameter(torch.zeros(8, 8, 3, 3)))   # standard layout: AccumulateGrad would clone dW
    nn_conv.no_side_stream([w])
    assert not nn_conv._side_ok(w)


def test_integration_index_lists_every_declared_symbol():
    """INTEGRATION.md section 5 (tools/abi_index.py) names every entry point the header declares."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "dsf_hip.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    index = doc[doc.index("## 5. Entry-point index"):]
    declared = re.findall(r"^(?:int|int64_t|const char\*) (dsf_[a-z0-9_]+)\(", hdr, re.M)
    assert len(declared) >= 70
    missing = [s for s in declared if "| `%s` |" % s not in index]
    assert not missing, missing


def test_checkpoint_round_trip(tmp_path):
    """dsf_amd/checkpoint.py against the reference trainer's rules (train_render.py:117-145, 282-308): the file format, the key
    filter (foreign keys dropped, absent keys keep the network's values), ``start_epoch = epoch + 1``, the optimizer state
    that is saved but never loaded, ``best.pth`` on ``<=``, and exchange with a plain-torch (reference-layout) state dict."""
    import torch
    from oracle import nets
    from dsf_amd import checkpoint as C
    from dsf_amd.model.backbone import MANO_OCR
    torch.manual_seed(0)
    net = MANO_OCR("ResNet_stage_18", 21)                         # product modules (parameters in kernel memory order), on the CPU
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    for q in net.parameters():
        q.grad = torch.ones_like(q) * 1e-3
    opt.step()
    ck = C.Checkpointer(str(tmp_path), net, opt)
    assert ck.end_of_epoch(3, test_error=12.0) is True            # 12 <= 100: best.pth written
    assert ck.end_of_epoch(4, test_error=12.0) is True            # `<=`: an equal error rewrites it
    assert ck.end_of_epoch(5, test_error=13.0) is False
    raw = torch.load(str(tmp_path / "latest.pth"), weights_only=False)
    assert sorted(raw) == ["epoch", "model", "optimizer"] and raw["epoch"] == 5
    assert torch.load(str(tmp_path / "best.pth"), weights_only=False)["epoch"] == 4
    assert list(raw["model"]) == list(net.state_dict())
    # the reference side: a plain torch.nn network with contiguous parameters takes the file with its own load rule
    torch.manual_seed(1)
    twin = nets.build(MANO_OCR, "ResNet_stage_18", 21)
    own = twin.state_dict()
    own.update({k: v for k, v in raw["model"].items() if k in own})
    twin.load_state_dict(own)
    for (k, a), (_, b) in zip(net.state_dict().items(), twin.state_dict().items()):
        assert a.shape == b.shape and torch.equal(a, b), k
    # and back: a reference-style file with a foreign key and a missing key into a fresh product network
    ref_state = {k: v.clone().contiguous() for k, v in twin.state_dict().items()}
    ref_state["module.not_ours"] = torch.zeros(3)
    gone = "mano_regress.2.bias"
    kept = None
    del ref_state[gone]
    torch.save({"model": ref_state, "optimizer": {"state": {}, "param_groups": []}, "epoch": 7}, str(tmp_path / "ref.pth"))
    torch.manual_seed(2)
    fresh = MANO_OCR("ResNet_stage_18", 21)
    kept = fresh.state_dict()[gone].clone()
    opt2 = torch.optim.AdamW(fresh.parameters(), lr=1e-3)
    merged, taken, dropped, missing = C.filter_state(ref_state, fresh)
    assert dropped == ["module.not_ours"] and missing == [gone] and len(taken) == len(fresh.state_dict()) - 1
    assert C.load_checkpoint(str(tmp_path / "ref.pth"), fresh, opt2) == 8            # resume: epoch + 1
    assert C.load_checkpoint(str(tmp_path / "ref.pth"), fresh, opt2, resume=False) == 0
    for k, v in fresh.state_dict().items():
        assert torch.equal(v, kept if k == gone else twin.state_dict()[k]), k
    assert len(opt2.state) == 0                                    # the file's optimizer state is not loaded (the reference never does)
    # the extension: restoring the optimizer state too
    opt3 = torch.optim.AdamW(net.parameters(), lr=1e-3)
    C.load_checkpoint(str(tmp_path / "latest.pth"), net, opt3, load_optimizer=True)
    assert len(opt3.state) == len(opt.state) > 0


def test_forked_stream_helpers_are_no_ops_without_a_gpu():
    """dsf_amd/streams.py on CPU tensors (the oracle twins run the product's module classes on the CPU): ``fork`` hands out
    null contexts and joins nothing; ``disabled()`` restores the switch; the zero pool of nn_conv stays closed."""
    import torch
    from dsf_amd import streams, nn_conv
    f = streams.fork("cpu")
    assert not f.on
    x = torch.ones(3)
    with f.branch(0, x):
        y = x + 1
    with f.branch(2):
        z = y * 2
    f.join()
    assert float(z.sum()) == 12.0
    was = streams.ENABLED[0]
    with streams.disabled():
        assert streams.ENABLED[0] is False
        with streams.disabled():
            pass
        assert streams.ENABLED[0] is False
    assert streams.ENABLED[0] == was

    class Owner:
        pass
    o = Owner()
    with nn_conv.zero_pool(o, "cpu"):
        assert nn_conv._ZERO is None
    assert "_zero_pool_floats" not in o.__dict__


def test_lanes_tool_accounts_for_every_nanosecond(tmp_path, capsys):
    """tools/lanes.py on a hand-made kernel trace: idle + one + two (+ more) kernels in flight = the step's wall time, the alone-in-
    flight time goes to the right workgroup bucket, gaps are counted (the config-3 work of round 4 was steered by this tool)."""
    import csv, importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lanes", os.path.join(root, "tools", "lanes.py"))
    lanes = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lanes)
    ms = 1000000
    rows = [  # start, end (ms), name, queue, workgroups  (one step between the two optimizer marks)
        (0, 1, "adamw_multi_kernel(x)", 1, 10),
        (2, 12, "void (anonymous namespace)::big_kernel<4>(float*)", 1, 512),      # alone 2-5 and 9-12
        (5, 9, "void (anonymous namespace)::side_kernel(float*)", 2, 8),           # overlaps the big one entirely
        (15, 16, "void (anonymous namespace)::tiny_kernel(float*)", 1, 2),          # alone, after a 3 ms gap
        (20, 21, "adamw_multi_kernel(x)", 1, 10),
    ]
    rows = [(s_ * ms, e_ * ms, n, q, g) for s_, e_, n, q, g in rows]
    path = tmp_path / "k.csv"
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Start_Timestamp", "End_Timestamp", "Kernel_Name", "Queue_Id", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z",
                    "Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"])
        for s, e, n, q, g in rows:
            w.writerow([s, e, n, q, g * 256, 1, 1, 256, 1, 1])
    lanes.main(str(path), 1)
    out = capsys.readouterr().out
    # the step runs from the first mark's end (1 ms) to the second's end (21 ms); kernels inside: big, side, tiny, adamw
    assert "step 20.00 ms, 4 kernels: idle 8.00 ms, one kernel in flight 8.00, two 4.00, three or more 0.00; kernel time 16.00 ms" in out
    assert "1: 3 launches 12.00 ms, 2: 1 launches 4.00 ms" in out
    assert "alone in flight, by workgroup count: 1+: 2.00 ms, 16+: 0.00 ms, 64+: 0.00 ms, 256+: 6.00 ms, 1024+: 0.00 ms" in out
    assert "alone    6.00 ms in    2 intervals  big_kernel<4>" in out
    assert "2 idle gaps, 7.00 ms in total, median 4000.0 us, 2 above 10 us" in out


def test_part_table_cache_never_serves_a_stale_table():
    """dsf_amd/metric/meshLoss.py::_cached_parts (round-5 verdict): the cache entry holds the face tensors it was built from (their
    addresses cannot be reused while it lives) and follows in-place edits through the version counters."""
    from dsf_amd.metric import meshLoss
    meshLoss._PART_CACHE.clear()
    a = [torch.tensor([[0, 1, 2], [2, 3, 0]]), torch.tensor([[4, 5, 6]])]
    cat, first = meshLoss._cached_parts(a, "cpu")
    assert cat.tolist() == [[0, 1, 2], [2, 3, 0], [4, 5, 6]] and first.tolist() == [0, 2, 3]
    assert meshLoss._cached_parts(a, "cpu")[0] is cat                      # served from the cache
    a[1][0, 0] = 7                                                        # in-place edit: a new entry
    assert meshLoss._cached_parts(a, "cpu")[0].tolist()[2] == [7, 5, 6]
    held = [t for e in meshLoss._PART_CACHE.values() for t in e[2]]
    assert any(t is a[0] for t in held)                                   # the entry keeps its sources alive
    addr = a[0].data_ptr()
    del a, held
    b = [torch.tensor([[9, 9, 9], [8, 8, 8]]), torch.tensor([[1, 1, 1]])]
    assert b[0].data_ptr() != addr or meshLoss._cached_parts(b, "cpu")[0].tolist()[0] == [9, 9, 9]
    assert meshLoss._cached_parts(b, "cpu")[0].tolist() == [[9, 9, 9], [8, 8, 8], [1, 1, 1]]
