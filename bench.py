#!/usr/bin/env python
"""Headline benchmark: images/sec (fwd+bwd+optimizer) of the DSF training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): per-GPU batch 32, ResNet-18 two-stage backbone
(``MANO_OCR_stage('ResNet_stage_18', 21, refine=True)``, reference config.py:38,80,93) + MANO layer +
fused crop rasteriser + GFM offset maps + SmoothL1 / m2d losses, backward, AdamW -- synthetic
NYU-shape inputs, random-init weights, fp32 (the reference's precision; the convolutions form every fp32
product from six bf16 MFMAs on exactly split operands -- same error against float64 as the fp32 MFMA, tests/test_gpu_conv.py;
DSF_CONV_MATH=f32 selects the fp32 MFMA kernels).
Weak scaling: every rank processes its own 32 images, gradients are averaged with a bucketed,
backward-overlapped RCCL all-reduce.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

def cpu_model():
    """the host CPU's model string (stated beside the core count of the CPU baseline)"""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def usable_cpus():
    """Cores this process may really use: min(affinity mask, cgroup quota).  os.cpu_count() reports the
    host's cores even inside a CPU-limited container; oversubscribing the OpenMP pools stalls for minutes."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_MATRIX_PEAK_TFLOPS = 157.3
BF16_MATRIX_PEAK_TFLOPS = 2500.0   # dense (MI355X_MICROARCH.md); the split kernels issue 6 bf16 MFMAs per fp32 product


def matrix_peak(kernel):
    """(peak in fp32-equivalent TFLOP/s, note) of the matrix pipe a convolution kernel runs on."""
    if "x6" in kernel:
        return (round(BF16_MATRIX_PEAK_TFLOPS / 6.0, 1),
                "fp32 products as 6 bf16 MFMAs (v_mfma_f32_32x32x16_bf16, exact 3-way operand split, fp32 accumulate): peak = "
                "bf16 dense 2500 TFLOP/s / 6 = 416.7 fp32-equivalent TFLOP/s; achieved counts ALGORITHMIC fp32 flops "
                "(2*M*N*K), i.e. the MFMA pipe executes 6x that; HBM traffic is not the bound")
    return (FP32_MATRIX_PEAK_TFLOPS,
            "fp32-in/fp32-acc MFMA (v_mfma_f32_32x32x2_f32) dense peak 157.3 TFLOP/s; HBM traffic is not the bound")


PMC_CONFIG = [2]          # which BASELINE config's committed PMC passes describe the running workload (set by main)


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes (profiles/rNN[_configK]_pmc_traffic.json, made by
    tools/pmc_summary.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this script;
    FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md).  A profiler cannot run inside the timed
    process, so the figure is read back from the profile of the same command; null when the file has no entry."""
    here = os.path.dirname(os.path.abspath(__file__))
    mid = "" if PMC_CONFIG[0] == 2 else "_config%d" % PMC_CONFIG[0]
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):              # the newest committed round that measured this kernel
        name = "%s%s_pmc_traffic.json" % (tag, mid)
        try:
            with open(os.path.join(here, "profiles", name)) as f:
                e = json.load(f)["kernels"][kernel]
            return {"traffic": e["hbm_bytes_per_launch"], "traffic_source": "profiles/%s (fetch x2 + write)" % name}
        except (OSError, KeyError, ValueError):
            continue
    return {"traffic": None}


def conv_kernel_roofline(step, run_once):
    """Roofline of the dominant kernel, igemm_fwd_kernel<128,false> (fp32 MFMA implicit-GEMM
    convolution: forward, backward-data and transposed-conv passes of every layer with Co > 64).
    One step is traced at the Python level, then every launch of that kernel is re-issued on the
    stream it runs on and timed with HIP events; `achieved` = algorithmic FLOPs of those launches
    (2*M*N*K of each implicit GEMM) / their summed duration."""
    from dsf_amd import nn_conv
    nn_conv.RECORD = []
    if step.grad_sync is not None:
        step.grad_sync.enabled = False          # rank-0-only diagnostic step: no collectives
    run_once()
    if step.grad_sync is not None and dist.is_initialized():
        step.grad_sync.enabled = True
    torch.cuda.synchronize()
    recs, nn_conv.RECORD = nn_conv.RECORD, None
    per_kernel = {}
    for r in recs:
        us, fl, nb = nn_conv.replay(r)
        k = per_kernel.setdefault(nn_conv.kernel_name(r), [0, 0.0, 0.0, 0.0])
        k[0] += 1; k[1] += us; k[2] += fl; k[3] += nb
    dom = max(per_kernel, key=lambda n: per_kernel[n][1])
    table = {name: {"launches_per_step": v[0], "avg_launch_us": round(v[1] / v[0], 1), "TFLOP/s": round(v[2] / (v[1] * 1e-6) / 1e12, 1),
                    "ms_per_step": round(v[1] / 1e3, 2), "algorithmic_MB_per_launch": round(v[3] / v[0] / 1e6, 2)} for name, v in per_kernel.items()}
    def entry(name, extra_note=""):
        n_, us_, fl_, nb_ = per_kernel[name]
        tf_ = fl_ / (us_ * 1e-6) / 1e12
        peak, note = matrix_peak(name)
        return {"kernel": name, "bound": "mfma", "achieved": round(tf_, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(tf_ / peak, 4), "launches_per_step": n_,
                "avg_launch_us": round(us_ / n_, 1), "ms_per_step": round(us_ / 1e3, 2), "flops_per_launch": fl_ / n_,
                "algorithmic_bytes_per_launch": nb_ / n_, **pmc_traffic(name), "note": note + extra_note}
    # the weight-gradient launches (igemm_wrw_*, conv_c1_wrw_*) run on a second stream beside the forward / backward-data chain and
    # hide behind it almost entirely; what bounds the step is the largest kernel of the MAIN queue
    main_q = [k for k in per_kernel if "wrw" not in k]
    crit = max(main_q, key=lambda k_: per_kernel[k_][1]) if main_q else dom
    critical = entry(crit, "; largest share of the MAIN queue (forward + backward-data convolutions): the weight-gradient kernels "
                           "run beside it on a second stream")
    critical["main_queue_conv_ms_per_step"] = round(sum(per_kernel[k][1] for k in main_q) / 1e3, 2)
    critical["weight_gradient_queue_ms_per_step"] = round(sum(v[1] for k, v in per_kernel.items() if "wrw" in k) / 1e3, 2)
    dom_entry = entry(dom)
    if "igemm_wrw_x6_kernel" in dom and "DSF_X6_WRW_WGS" not in os.environ:
        # Round 6: the launcher splits these launches' pixels towards ONE workgroup per CU, which is the better choice INSIDE a step (they
        # run beside the main queue's kernels: config 2 -2.7 %, config 3 -3.7 % per step, profiles/r06_wrw_split_target.txt) and the slower
        # one for the launch ALONE, which is what this entry times.  The same launches at the isolated optimum (two per CU), for reference:
        os.environ["DSF_X6_WRW_WGS"] = "512"
        try:
            us2 = fl2 = 0.0
            for r in recs:
                if nn_conv.kernel_name(r) == dom:
                    u, f, _ = nn_conv.replay(r)
                    us2 += u; fl2 += f
        finally:
            del os.environ["DSF_X6_WRW_WGS"]
        if us2 > 0:
            tf2 = fl2 / (us2 * 1e-6) / 1e12
            dom_entry["alone_at_two_workgroups_per_cu"] = {
                "avg_launch_us": round(us2 / per_kernel[dom][0], 1), "achieved": round(tf2, 2), "frac": round(tf2 / dom_entry["peak"], 4),
                "note": "the same launches with DSF_X6_WRW_WGS=512 (the split of rounds 3-5, the optimum of the launch alone); the step "
                        "runs them at one workgroup per CU because the STEP is faster that way (profiles/r06_wrw_split_target.txt)"}
    return dom_entry, table, critical


def measured_mfma_ceiling():
    """Live: TFLOP/s a bare bf16 MFMA loop (random operands, no memory traffic, 2 workgroups per CU) sustains on this GPU,
    and the same in fp32-equivalent terms (/ 6) -- the practical ceiling of the split-operand convolution kernels."""
    import ctypes
    from dsf_amd import _lib as L
    ops_ = (torch.randint(0, 256, (4096,), device="cuda", dtype=torch.int32) | 0x3f00)
    ops_ = (ops_ | (ops_.roll(1) << 16)).contiguous()                       # bf16 pairs in [0.5, 1)
    wgs, iters = 512, 3000
    out = torch.empty(wgs * 256, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda n: L.lib().dsf_mfma_bf16_probe(ctypes.c_void_p(ops_.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_int(wgs),
                                                 ctypes.c_int(n), st)
    call(100)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = sum(call(iters) for _ in range(4))
    e1.record()
    torch.cuda.synchronize()
    tf = n * 2.0 * 32 * 32 * 16 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    return {"bf16_TFLOP/s": round(tf, 1), "fp32_equivalent_TFLOP/s": round(tf / 6.0, 1),
            "note": "bare v_mfma_f32_32x32x16_bf16 loop, 2 workgroups per CU, timed here: what the matrix pipe sustains at the "
                    "clock the chip holds under it (nominal 2500 assumes 2.4 GHz)"}


def same_step_on_fp32_mfma(step, tgt, B, steps=10, warmup=3):
    """Rank-0 diagnostic beside the headline: the same training step with every convolution on the plain fp32 MFMA
    (v_mfma_f32_32x32x2_f32, conv.hip) instead of the split-operand kernels -- what DSF_CONV_MATH=f32 would measure."""
    from dsf_amd import nn_conv
    if nn_conv.MATH != "x6":
        return None
    if step.grad_sync is not None:
        step.grad_sync.enabled = False
    nn_conv.MATH = "f32"
    try:
        for _ in range(warmup):
            step(tgt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(tgt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    finally:
        nn_conv.MATH = "x6"
        if step.grad_sync is not None and dist.is_initialized():
            step.grad_sync.enabled = True
    return {"value": round(B / dt, 2), "unit": "images/s per GPU", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
            "note": "same step, convolutions on the fp32 MFMA kernels (peak 157.3 TFLOP/s); both paths agree with float64 to "
                    "5e-7 of the largest output (tests/test_gpu_conv.py)"}


def gpu_time_per_call_us(fn, launches, warmup=5, capture=True):
    """GPU time per call of ``fn`` (every kernel it issues, layout copies included) WITHOUT the host: ``launches`` calls are
    captured into one HIP graph on a side stream and the replay is bracketed by events.  Rounds 1-4 timed eager loops, which for
    a 20-50 us operation measure the Python call (torch.empty, ctypes, autograd.Function: 30-70 us) -- the crop rasteriser's
    "25 us floor" and the 71 us soft-argmax decode of the round-4 tables were that.  Falls back to the eager loop for an
    operation that cannot be captured (a host synchronisation inside)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    try:
        if not capture:                                  # (an autograd backward pass inside a capture crashes this runtime: eager)
            raise RuntimeError("eager")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(launches):
                    fn()
            graph.replay()
            side.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            graph.replay()
            e1.record(side)
            side.synchronize()
        torch.cuda.current_stream().wait_stream(side)
        us = e0.elapsed_time(e1) * 1e3 / launches
        del graph
        return us, "graph replay of %d calls" % launches
    except Exception:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / launches, "eager loop of %d calls (host time included)" % launches


def crop_kernel_roofline(render, B, launches=200):
    """Live timing of the fused crop rasteriser (dsf_render_crop_forward) with HIP events on the stream
    it is launched on.  Algorithmic bytes per image (SURVEY 8d, K1 crop mode): read verts 779*12 = 9,348 B,
    write depth crop 128*128*4 = 65,536 B + face index 65,536 B."""
    from dsf_amd import ops
    from dsf_amd.train_step import synthetic_batch
    p, c, cube = synthetic_batch(B, "cuda", seed=123)
    mano = render.mano_layer
    with torch.no_grad():
        v, _ = mano.get_mano_vertices(p[:, :3], p[:, 3:48], p[:, 48:58], p[:, 58:62], 1 / 125)
        verts = (v * cube.unsqueeze(1) / 2 + c.unsqueeze(1)).contiguous()
        c2, M, _, _ = ops.crop_setup(c, cube, render.cam, 128)
        minv = torch.linalg.inv_ex(M)[0].contiguous()
        cz, cbz = c2[:, 2].contiguous(), cube[:, 2].contiguous()
        run = lambda: ops.RenderCropFunction.apply(verts, mano.faces_i32, minv, render.resize_rowmap, cz, cbz,
                                                   render.cam, 640, 128)
        us, how = gpu_time_per_call_us(run, launches, warmup=10)
    bytes_per_launch = B * (779 * 12 + 128 * 128 * 4 + 128 * 128 * 4)
    achieved = bytes_per_launch / (us * 1e-6) / 1e9
    # VALU roofline (what actually bounds it): the kernel's work is one coverage evaluation per (crop pixel, face) pair whose
    # raster pixel lies inside the face's bounding box -- counted here exactly, with the kernel's own pixel map and boxes --
    # at ~25 lane-instructions for a rejected pair (3 edge functions, sign tests, loop) and ~65 for a covered one (3 IEEE
    # divisions, depth, key, LDS atomic); peak = 256 CUs x 4 SIMD x 32 lanes x 2.4 GHz lane-instructions per second.
    with torch.no_grad():
        W, H, S = 640.0, 480.0, 640
        fxn, fyn = render.cam.fx / (W / 2), render.cam.fy / (H / 2)
        X, Y, Z = verts.unbind(-1)
        xn, yn = (-X * fxn) / Z, (-Y * fyn) / Z                                  # px' = py' = 0 for this camera (A.1)
        fv = mano.faces_i32.long()
        fx3, fy3 = xn[:, fv], yn[:, fv]                                          # (B,F,3)
        box = lambda lo, hi: (((0.5 * (S * (1 - hi) - 1)).clamp(min=-4).ceil() - 1).clamp(min=0),
                              ((0.5 * (S * (1 - lo) - 1)).clamp(max=S + 4).floor() + 1).clamp(max=S - 1))
        xlo, xhi = box(fx3.amin(-1), fx3.amax(-1))
        ylo, yhi = box(fy3.amin(-1), fy3.amax(-1))
        jj, ii = torch.meshgrid(torch.arange(128.0, device=verts.device), torch.arange(128.0, device=verts.device), indexing="xy")
        mi = minv.float()
        sx = (mi[:, 0, 0, None, None] * jj + mi[:, 0, 1, None, None] * ii) + mi[:, 0, 2, None, None]
        sy = (mi[:, 1, 0, None, None] * jj + mi[:, 1, 1, None, None] * ii) + mi[:, 1, 2, None, None]
        fxp = torch.round(((sx / W) * 2 - 1 + 1) * (W / 2) - 0.5)
        fyp = torch.round(((sy / H) * 2 - 1 + 1) * (H / 2) - 0.5)
        ok = (fxp >= 0) & (fxp < W) & (fyp >= 0) & (fyp < H)
        ry = render.resize_rowmap.long()[fyp.clamp(0, H - 1).long()].float()
        evals = 0
        for b in range(B):                                                       # (F, 16384) comparisons per sample
            px, py, m = fxp[b].reshape(1, -1), ry[b].reshape(1, -1), ok[b].reshape(1, -1)
            inside = m & (px >= xlo[b, :, None]) & (px <= xhi[b, :, None]) & (py >= ylo[b, :, None]) & (py <= yhi[b, :, None])
            evals += int(inside.sum())
        covered = int((run()[0] < 0.99).sum())
    lane_ops = 25.0 * evals + 40.0 * covered
    valu_peak = 256 * 4 * 32 * 2.4e9
    return {"kernel": "render_crop_fwd_kernel", "bound": "valu", "avg_launch_us": round(us, 2),
            "achieved": round(lane_ops / (us * 1e-6) / 1e12, 3), "peak": round(valu_peak / 1e12, 1), "unit": "T lane-instructions/s",
            "frac": round(lane_ops / (us * 1e-6) / valu_peak, 4),
            "coverage_evaluations_per_launch": evals, "covered_pixels_per_launch": covered,
            "images_per_s": round(B / (us * 1e-6), 1), "hbm_achieved": round(achieved, 2), "hbm_peak": HBM_PEAK_GBS,
            "hbm_frac": round(achieved / HBM_PEAK_GBS, 5), **pmc_traffic("render_crop_fwd_kernel"),
            "bytes_per_launch": bytes_per_launch,
            "timed_as": how,
            "note": "VALU / latency-bound, not a bandwidth kernel: 4.5 MB of algorithmic traffic per launch.  `frac` is a MODEL "
                    "count (coverage evaluations x instructions per evaluation) over the VALU issue peak.  Round 5: faces binned "
                    "into per-tile LDS lists by the whole workgroup, heavy tiles shared by four waves, tiles Morton-interleaved "
                    "over the workgroups (78 -> 48 us at B = 32; tools/crop_stamps.py shows where a launch's cycles go)"}


def geometry_rooflines(render, B, launches=50):
    """"report each fraction separately" -- the geometry kernels of the path timed alone with HIP events on the stream
    they are launched on (torch's current stream), each against the bound that applies to it.  Sizes = one GPU's share of the
    BASELINE configuration that uses the kernel (B samples, 2048-point clouds, 1554 faces, 21 joints, 84-channel 64x64 maps).
    Timed host-free (gpu_time_per_call_us): a row is the GPU time of the whole API call, layout copies and small helper kernels included."""
    from dsf_amd.metric.meshLoss import ICPLoss, JointICPLoss
    from dsf_amd.train_step import synthetic_batch
    from dsf_amd.util.generateFeature import GFM
    mano = render.mano_layer
    gfm = GFM()
    p, c, cube = synthetic_batch(B, "cuda", seed=321)
    P = 2048

    def timed(fn, capture=True):
        return gpu_time_per_call_us(fn, launches, capture=capture)[0]
    rows = []
    valu_peak = 256 * 4 * 32 * 2.4e9                       # lane-instructions per second (as in roofline_raster)

    def hbm_row(kid, kernel, us, nbytes, note):
        gbs = nbytes / (us * 1e-6) / 1e9
        rows.append({"id": kid, "kernel": kernel, "avg_launch_us": round(us, 2), "bound": "hbm", "achieved": round(gbs, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "bytes_per_launch": int(nbytes),
                     "note": note})
    with torch.no_grad():
        jx, mesh = render.get_mesh_xyz(p)
        g = torch.Generator(device="cuda").manual_seed(5)
        idx = torch.randint(0, 779, (B, P), device="cuda", generator=g)
        pcl = (torch.gather(mesh, 1, idx[..., None].expand(-1, -1, 3)) + 0.02 * torch.randn(B, P, 3, device="cuda", generator=g)).contiguous()
        seg = mano.seg_pcl(jx, jx, mesh, pcl)
        # K3: point-to-triangle distance, every point against every triangle (ICPLoss) and against its part (JointICPLoss)
        us = timed(lambda: ICPLoss(mesh, pcl, mano.faces))
        pairs = B * P * 1554
        per_pair = 18.0                                     # the cheapest a pair can be: seed compare (9) + sphere test (9)
        rows.append({"id": "K3", "kernel": "mesh_point_fwd_kernel (ICPLoss: %d points x 1554 triangles per sample)" % P,
                     "avg_launch_us": round(us, 2), "bound": "valu", "achieved": round(pairs / (us * 1e-6) / 1e12, 4),
                     "peak": round(valu_peak / per_pair / 1e12, 3), "unit": "T pair-tests/s",
                     "frac": round(pairs / (us * 1e-6) / (valu_peak / per_pair), 4), "pairs_per_launch": pairs,
                     "note": "algorithmic pair-tests (every point x every triangle of its sample) per second; peak = VALU issue "
                             "peak / 18 lane-instructions, the least a pair costs in the culled kernel (nearest-centre seed "
                             "compare + bounding-sphere test); a pair that survives the cull costs ~120 more (the oracle's "
                             "arithmetic incl. 5 IEEE divisions).  Round 3's brute-force kernel: 712 us at B = 64 = 0.29 T/s"})
        us = timed(lambda: JointICPLoss(mesh, pcl, mano.joint_faces, seg))
        rows.append({"id": "K3p", "kernel": "mesh_point_fwd_kernel (JointICPLoss: points x the triangles of their part, 15 parts)",
                     "avg_launch_us": round(us, 2), "bound": "valu", "note": "eight workgroups per (sample, part) deal the part's "
                     "64-point groups out between them (round 5); includes the masked per-part means around the kernel; no separate model"})
        # K5: MANO layer (two launches each way since round 4)
        q = p.clone().requires_grad_(True)
    from dsf_amd import ops
    with torch.no_grad():
        us = timed(lambda: ops.ManoPackedFunction.apply(mano._native(), p, 1000.0, 1.0))
    const_bytes = (135 + 10 + 1) * 2334 * 4 + 778 * 16 * 4
    hbm_row("K5f", "mano_blend_kernel + mano_skin_kernel (forward)", us, B * (62 * 4 + 779 * 12 + 21 * 12 + 5248 * 4) + const_bytes,
            "latency-bound: 2 dependent launches, the second one a 15-step kinematic chain per sample; bytes = parameters in, "
            "vertices / joints / saved state out per sample + the 1.4 MB of model constants once")
    v, j = ops.ManoPackedFunction.apply(mano._native(), q, 1000.0, 1.0)
    gv, gj = torch.randn_like(v), torch.randn_like(j)
    us = timed(lambda: torch.autograd.grad([v, j], q, [gv, gj], retain_graph=True), capture=False)
    hbm_row("K5b", "mano_skin_bwd_kernel + mano_blend_bwd_kernel (backward)", us,
            B * (779 * 12 + 21 * 12 + 5248 * 4 + 2 * 2560 * 4 + 62 * 4 + const_bytes),
            "latency-bound: 256-thread workgroups that fit beside two convolution workgroups per CU (78 us as one 1024-thread "
            "workgroup per sample in round 3, which then waited 700 us for a drained CU inside the step); the constants are "
            "L2 / Infinity Cache resident")
    with torch.no_grad():
        # K6 / K7
        us = timed(lambda: mano.calculate_coll(jx, mesh))
        hbm_row("K6", "sphere_set + collision_fwd_kernel (calculate_coll)", us, B * (779 * 12 + 21 * 12 + 66 * 16 + 264),
                "latency-bound (one workgroup per sample, iterative top-10 per wave); bytes are negligible")
        us = timed(lambda: mano.seg_pcl(jx, jx, mesh, pcl))
        hbm_row("K7", "sphere_set + seg_pcl_kernel (%d points per sample)" % P, us, B * (P * 12 + P * 8 + 66 * 16),
                "VALU-bound by model (66 sphere tests per point), tiny; reported against HBM as SURVEY 8d lists it")
        # K10: offset-map encode / decode (84 channels x 64 x 64 per sample)
        img = torch.rand(B, 1, 128, 128, device="cuda")
        juvd = (torch.rand(B, 21, 3, device="cuda") - 0.5)
        us = timed(lambda: gfm.joint2offset(juvd, img, 0.8, 64))
        hbm_row("K10e", "joint2offset_fwd_cl_kernel (encode)", us, B * (84 * 64 * 64 * 4 + 128 * 128 * 4 + 21 * 12),
                "streaming write of the 84-channel map (channels-last)")
        maps = gfm.joint2offset(juvd, img, 0.8, 64)
        us = timed(lambda: gfm.offset2joint_softmax(maps, img, 0.8))
        hbm_row("K10d", "offset2joint_* (soft-argmax decode)", us, B * (84 * 64 * 64 * 4 + 128 * 128 * 4 + 21 * 12),
                "the 84-channel channels-last map: per-chunk maxima over the 21 heat channels, then one read of the map; pixel-chunk "
                "workgroups (round 6), three small launches")
    return rows


def self_launch(n):
    """One torch.distributed.run child with n ranks on 127.0.0.1; returns its exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this driver
    return relay_one_line(cmd, env)


def relay_one_line(cmd, env):
    """Runs the launcher child; of what its ranks write to stdout only the bench line reaches this process's stdout (the
    contract is ONE JSON line), everything else -- a backend's connection banner, a stray print -- goes to stderr."""
    import subprocess
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in child.stdout:
        (sys.stdout if line.startswith('{"metric"') else sys.stderr).write(line)
    sys.stdout.flush()
    return child.wait()


def rccl_debug_file():
    """Asks RCCL for its init / graph log in a per-process file (unless the caller configured NCCL_DEBUG already), so
    that the JSON line can carry what the library itself reports: ranks of the communicator, channels, rings / trees."""
    if "NCCL_DEBUG" in os.environ:
        return os.environ.get("NCCL_DEBUG_FILE")
    import tempfile
    path = os.path.join(tempfile.gettempdir(), "dsf_rccl_%d.log" % os.getpid())
    os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH", NCCL_DEBUG_FILE=path)
    return path


def rccl_summary(path):
    """Counts of the lines RCCL wrote about this rank's communicator (best effort: the wording is RCCL's)."""
    import re
    out = {"log": None}
    try:
        txt = open(path).read() if path else ""
    except OSError:
        return out
    lines = txt.splitlines()
    out["log"] = {"lines": len(lines),
                  "ring_lines": sum(1 for l in lines if re.search(r"\bRing \d+", l)),
                  "tree_lines": sum(1 for l in lines if re.search(r"\bTrees? ", l)),
                  "channels": max([int(m.group(1)) for l in lines for m in [re.search(r"(\d+) coll channels", l)] if m] or [0]),
                  "nranks_reported": sorted({int(m.group(1)) for l in lines for m in [re.search(r"nranks (\d+)", l)] if m}),
                  "version": next((l.split("version", 1)[1].strip(" :") for l in lines if "NCCL version" in l or "RCCL version" in l), None)}
    return out


def distributed_facts(world, dev, rccl_log):
    """What the process group itself reports (all ranks call this): backend, observed world size, one row per rank
    (host pid, device index, device name, PCI bus id) and an all-reduce of ones as a live check that the collective
    spans that many ranks."""
    if world == 1:
        return {"backend": None, "world_size_observed": 1}
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": dist.get_rank(), "pid": os.getpid(), "device": dev.index, "name": props.name,
            "pci_bus_id": getattr(props, "pci_bus_id", None)}
    rows = [None] * world
    dist.all_gather_object(rows, mine)
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    facts = {"backend": dist.get_backend(), "world_size_observed": dist.get_world_size(), "allreduce_of_ones": float(ones.item()),
             "distinct_devices": len({(r["device"], r["pci_bus_id"]) for r in rows}), "ranks": rows}
    if dist.get_backend() == "nccl":
        try:
            facts["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
        facts.update(rccl_summary(rccl_log))
    return facts


def build_workload(args, dev, rank, world):
    """The per-GPU share of a BASELINE config as: the step object, a zero-argument ``run`` (one optimizer step on
    resident synthetic inputs), units per step and what a unit is, network FLOPs per unit, and the CPU-oracle leg."""
    import numpy as np
    from dsf_amd.parallel import GradAllReducer
    from dsf_amd.render_model.mano_layer import Render
    from dsf_amd.model.backbone import MANO_OCR_stage
    from dsf_amd import ops
    from dsf_amd.train_step import (RenderSupervisedStep, MeshLossStep, PretrainStep, FinetuneStageStep, synthetic_batch, Config)
    cfg = args.config
    B = args.batch or {2: 32, 3: 64, 4: 64, 5: 64}[cfg]
    torch.manual_seed(0)                              # identical initial weights on every rank
    render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).to(dev)
    p, c, cube = synthetic_batch(B, dev, seed=0 + rank)            # per-rank shard of the global batch
    w = {"render": render, "units_per_step": B, "tgt": None}
    fit = None
    if getattr(args, "init", "fresh") == "fitted":
        # one pose, B small perturbations of it (tests/test_gpu_steps.py::_selfsup_setup's fitted case at batch size)
        fit = synthetic_batch(1, "cpu", seed=23)[0][0]
        noise = 0.003 * torch.randn(B, 62, generator=torch.Generator().manual_seed(2 + rank))
        noise[:, 58:] = 0.0
        p = (fit[None] + noise).to(dev)

    def fit_heads(net):
        if fit is None:
            return
        with torch.no_grad():
            for name in ("mano_regress", "mano_regress_s2"):
                head = getattr(net, name, None)
                if head is not None:
                    head[2].bias.copy_(fit.to(head[2].bias.device))
                    head[2].weight.mul_(0.05)              # the head stays within ~0.01 of the pose for any input

    def oracle_bits():
        from dsf_amd.assets import build_synthetic_mano
        from oracle import step_ref, nets                 # CPU oracle: the reported baseline leg only
        return step_ref, nets, step_ref.OracleRender(build_synthetic_mano(0))

    def timed_cpu(loss_fn, net_cpu, n_steps, units):
        opt = torch.optim.AdamW(net_cpu.parameters(), lr=1e-3, weight_decay=0.01)
        t0 = None
        for it in range(1 + n_steps):
            if it == 1:
                t0 = time.perf_counter()
            opt.zero_grad()
            loss_fn().backward()
            opt.step()
        secs = time.perf_counter() - t0
        return units * n_steps / secs, secs, units * n_steps

    if cfg == 2:
        backbone = args.backbone or "ResNet_stage_18"
        net = MANO_OCR_stage(backbone, 21, True).to(dev)
        fit_heads(net)
        sync = GradAllReducer(net.parameters()) if world > 1 else None
        step = RenderSupervisedStep(net, render, Config, grad_sync=sync)
        tgt = step.make_targets(p, c, cube, seed=1 + rank)
        fl = 3 * 25.42e9 if backbone.endswith("18") else 3 * 37.84e9
        w.update(step=step, tgt=tgt, run=lambda: step(tgt), run_diag=lambda: step(tgt), flops_per_unit=fl,
                 flops_note="3 x %.2f GFLOP per image, fwd + bwd" % (fl / 3e9), unit_note="one image through one optimizer step",
                 workload="BASELINE configs[1]: batch=%d/GPU %s 2-stage + MANO + depth rasteriser, single view" % (B, backbone))

        def cpu(n_steps):
            from dsf_amd.assets import build_synthetic_mano
            from oracle import step_ref
            n_steps = n_steps or 60
            ips, secs, n = step_ref.timed_steps(build_synthetic_mano(0), B=2, steps=n_steps, warmup=1, backbone=backbone)
            return ips, secs, n, "the same step (same net, losses, AdamW) at B=2 x %d steps" % n_steps
        w["cpu"] = cpu
    elif cfg == 3:
        from dsf_amd.model.hourglass import PoseNetMANO
        net = PoseNetMANO(2, 21).to(dev)
        fit_heads(net)
        sync = GradAllReducer(net.parameters()) if world > 1 else None
        step = MeshLossStep(net, render, Config, grad_sync=sync)
        tgt = step.make_targets(p, c, cube, seed=1 + rank)
        w.update(step=step, tgt=tgt, run=lambda: step(tgt), run_diag=lambda: step(tgt), flops_per_unit=3 * 4.58e9,
                 flops_note="3 x 4.58 GFLOP per image, fwd + bwd (the geometry losses are not MFMA work)",
                 unit_note="one image through one optimizer step",
                 workload="BASELINE configs[2]: batch=%d/GPU hourglass-2-stack + MANO head, m2d + ICP + part ICP + sphere collision" % B)

        def cpu(n_steps):
            step_ref, nets, orender = oracle_bits()
            n_steps = n_steps or 20
            torch.manual_seed(0)
            net_cpu = nets.build(PoseNetMANO, 2, 21)
            pc, cc, cubec = synthetic_batch(2, "cpu", seed=0)
            g = torch.Generator().manual_seed(1)
            keys = [torch.randint(0, 2 ** 31 - 1, (2, 128 * 128), dtype=torch.int32, generator=g) for _ in range(2)]
            tc = step_ref.mesh_targets(orender, pc, cc, cubec, keys[0], keys[1])
            ips, secs, n = timed_cpu(lambda: step_ref.mesh_step_loss(net_cpu, orender, tc, Config)[0], net_cpu, n_steps, 2)
            return ips, secs, n, "the same step (hourglass-2 + mesh losses, AdamW) at B=2 x %d steps" % n_steps
        w["cpu"] = cpu
    elif cfg == 4:
        net = MANO_OCR_stage("ResNet_stage_50", 21, True).to(dev)
        fit_heads(net)
        sync = GradAllReducer(net.parameters()) if world > 1 else None
        step = PretrainStep(net, render, None, Config, grad_sync=sync, views=3)
        d = step.draw(B, dev, torch.Generator(device=dev).manual_seed(4 + rank), np.random.default_rng(4 + rank))
        w.update(step=step, run=lambda: step(p, cube, d), run_diag=lambda: step(p, cube, d), flops_per_unit=3 * 3 * 37.84e9,
                 flops_note="3 views x 3 x 37.84 GFLOP per sample, fwd + bwd",
                 unit_note="one SAMPLE (rendered from 3 augmentView rotations = 3 images through the network) through one optimizer step",
                 workload="BASELINE configs[3], per-GPU share: %d samples x 3 synthetic camera views per GPU, ResNet-50 2-stage, "
                          "Trainer.Pretrain step (Render.forward with view / shape / centre / size augmentation + occluders)" % B)

        def cpu(n_steps):
            step_ref, nets, orender = oracle_bits()
            n_steps = n_steps or 4
            torch.manual_seed(0)
            net_cpu = nets.build(MANO_OCR_stage, "ResNet_stage_50", 21, True)
            pc, _, cubec = synthetic_batch(2, "cpu", seed=0)
            dc = step.draw(2, "cpu", torch.Generator().manual_seed(4), np.random.default_rng(4))
            ips, secs, n = timed_cpu(lambda: step_ref.pretrain_loss(net_cpu, orender, None, pc, cubec, dc, Config, views=3), net_cpu, n_steps, 2)
            return ips, secs, n, "the same step (ResNet-50 2-stage, 3 views per sample, AdamW) at 2 samples x %d steps" % n_steps
        w["cpu"] = cpu
    else:
        from dsf_amd.render_model.transfer import define_G
        net = MANO_OCR_stage("ResNet_stage_18", 21, True).to(dev)
        with torch.no_grad():
            for head in (net.mano_regress[2], net.mano_regress_s2[2]):
                head.bias[58] = 1.0                    # unit global scale: non-degenerate hands from random-init heads
        fit_heads(net)
        gen = define_G(1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier').to(dev)
        sync = GradAllReducer(net.parameters()) if world > 1 else None
        step = FinetuneStageStep(net, render, gen, Config, grad_sync=sync)
        pr, cr, cube_r = synthetic_batch(B, dev, seed=100 + rank)
        if fit is not None:                            # the real images show the pose the heads reproduce
            pr = (fit[None] + 0.003 * torch.randn(B, 62, generator=torch.Generator().manual_seed(7 + rank))).to(dev)
            pr[:, 58:] = fit[58:].to(dev)
        with torch.no_grad():
            img_r = render.render(pr, cr, cube_r)[0]
            _, M_r, _, _ = ops.crop_setup(cr, cube_r, render.cam, 128)
        d = step.draw(B, dev, torch.Generator(device=dev).manual_seed(5 + rank), np.random.default_rng(5 + rank))
        run = lambda: step(p, cube, img_r, cr, cube_r, M_r, draws=d)
        w.update(step=step, run=run, run_diag=run, flops_per_unit=2 * 3 * 25.42e9 + 24.36e9,
                 flops_note="2 images (synthetic + real) x 3 x 25.42 GFLOP + 24.36 GFLOP of the frozen transfer generator per pair",
                 unit_note="one PAIR (synthetic + real image, the real batch counts: main_loader = trainLoader) through one optimizer step",
                 workload="BASELINE configs[4], per-GPU share: %d synthetic + %d real images per GPU, full dual-branch self-boosting "
                          "Trainer.FinetuneStage step with the frozen Consis-CycleGAN generator, ResNet-18 2-stage" % (B, B))

        def cpu(n_steps):
            step_ref, nets, orender = oracle_bits()
            n_steps = n_steps or 8
            torch.manual_seed(0)
            net_cpu = nets.build(MANO_OCR_stage, "ResNet_stage_18", 21, True)
            with torch.no_grad():
                for head in (net_cpu.mano_regress[2], net_cpu.mano_regress_s2[2]):
                    head.bias[58] = 1.0
            gen_cpu = nets.build(define_G, 1, 1, 64, 'resnet_9blocks', 'instance', False, 'xavier').eval()
            pc, _, cubec = synthetic_batch(2, "cpu", seed=0)
            prc, crc, cube_rc = synthetic_batch(2, "cpu", seed=100)
            with torch.no_grad():
                img_rc = orender.render(prc, crc, cube_rc)[0]
            dc = step.draw(2, "cpu", torch.Generator().manual_seed(5), np.random.default_rng(5))
            fn = lambda: step_ref.finetune_stage_loss(net_cpu, orender, gen_cpu, pc, cubec, img_rc, crc, cube_rc, dc, Config)[0]
            ips, secs, n = timed_cpu(fn, net_cpu, n_steps, 2)
            return ips, secs, n, "the same step (FinetuneStage incl. the transfer generator, AdamW) at 2 pairs x %d steps" % n_steps
        w["cpu"] = cpu
    return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5),
                    help="BASELINE.json configs[n-1]: 2 (default, the headline) ResNet-18 two-stage + MANO + rasteriser, B=32; 3 hourglass-2 "
                         "+ mesh losses, B=64; 4 ResNet-50 two-stage x 3 views, 64 samples per GPU; 5 FinetuneStage + transfer net, 64 pairs per GPU")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (weak scaling); 0 = the config's own (32 / 64 / 64 / 64)")
    ap.add_argument("--backbone", default="", help="config 2 only: ResNet_stage_18 (default) / ResNet_stage_50")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--init", default="fresh", choices=("fresh", "fitted"),
                    help="fresh: random-init network (the default and the contract's workload).  fitted: the steady state a trained "
                         "estimate is in (train_render.py:735-739, 777-781 run the mesh losses on one): every sample of the batch is a "
                         "small perturbation of one pose and the MANO heads reproduce that pose, so that the predicted mesh lies on "
                         "the data -- the state in which the point-to-triangle cull works; reported as config.init")
    ap.add_argument("--graph", action="store_true", help="replay forward+backward from a HIP graph (train_step.GraphedStep; "
                    "single GPU): same kernels, no host issue -- pays below batch 16, where the step is host-bound")
    ap.add_argument("--no-graph", action="store_true", help="config 3 only: the eager step instead of its default HIP-graph replay")
    ap.add_argument("--cpu-steps", type=int, default=0, help="steps of the CPU baseline leg (0: sized per config to ~10-30 s)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (one per GPU) before this
        # process has made any GPU call, relay rank 0's JSON line (the children inherit stdout) and exit with their code.
        # Never re-exec: a process that has initialised the GPU must not be replaced on this pool.
        sys.exit(self_launch(args.gpus))

    from dsf_amd.parallel import init_distributed, GradAllReducer
    rccl_log = rccl_debug_file() if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None
    rank, local, world = init_distributed()
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s); launch with torch.distributed.run "
                         "--nproc-per-node %d (or run `python bench.py --gpus %d` bare: it launches itself)\n"
                         % (args.gpus, world, args.gpus, args.gpus))
        sys.exit(2)
    local = local % max(torch.cuda.device_count(), 1)          # (ranks may share a device in single-GPU flow tests)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # neither convolutions (dsf_conv_x6_* / dsf_conv_igemm_*) nor BatchNorm (dsf_bn_*) reach MIOpen; the switch only matters for
    # DSF_FUSED_BN=0 runs
    torch.backends.cudnn.benchmark = os.environ.get("DSF_MIOPEN_FIND", "0") == "1"   # reference :87 uses find mode; gfx950 ships no MIOpen find-db, find mode JIT-compiles every solver (hours)

    PMC_CONFIG[0] = args.config
    w = build_workload(args, dev, rank, world)
    step, run = w["step"], w["run"]
    # config 3 is host-bound in eager mode (1,400+ launches of 10-60 us kernels per step): its default is the HIP-graph replay
    if args.config == 3 and not args.no_graph:
        args.graph = True
    if args.graph:
        if args.config not in (2, 3):
            raise SystemExit("--graph: configs 2 and 3 (steps without a host decision); with --gpus N the bucket all-reduces follow each replay")
        from dsf_amd.train_step import GraphedStep
        g = GraphedStep(step, w["tgt"])
        run = lambda: g(w["tgt"])
    # DSF_MAIN_PRIORITY=-1 (A/B aid): the training loop on a HIGH-priority stream, so that the weight-gradient stream is the lower one
    import contextlib
    prio = int(os.environ.get("DSF_MAIN_PRIORITY", "0"))
    loop_stream = torch.cuda.stream(torch.cuda.Stream(priority=prio)) if prio else contextlib.nullcontext()
    torch.cuda.synchronize()
    with loop_stream:
        for _ in range(args.warmup):
            run()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss, _ = run()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss_val = float(loss)
    facts = distributed_facts(world, dev, rccl_log)
    if world > 1:
        # every rank leaves the process group HERE, together: rank 0's single-rank diagnostics below (kernel replays, the fp32-MFMA
        # comparison step: ~10 s, no collectives, the reducer switched off) must not run while the other ranks wait inside a
        # collective teardown under RCCL's watchdog
        if step.grad_sync is not None:
            step.grad_sync.enabled = False
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        B = w["units_per_step"]
        images = B * world * args.steps
        out = {
            "metric": "images/sec (fwd+bwd, 128x128 depth, MANO+render loss)",
            "value": round(images / dt, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "conv_math": os.environ.get("DSF_CONV_MATH", "x6") + (": fp32 products as 6 bf16 MFMAs on exact 3-way operand splits, fp32 accumulate"
                                                                if os.environ.get("DSF_CONV_MATH", "x6") == "x6" else ": fp32 MFMA"),
            "config": {"workload": w["workload"], "baseline_config": args.config, "unit_of_value": w["unit_note"],
                       "global_batch": B * world, "hip_graph": bool(args.graph), "init": args.init,
                       "weight_gradients_on_second_stream": os.environ.get("DSF_WRW_STREAM", "1") != "0", "crop": 128, "raster": 640, "parallelism": "dp%d" % world,
                       "mano_asset": "synthetic MANO-shaped hand (real MANO_RIGHT.pkl is license-gated)"},
            "final_loss": round(loss_val, 5),
            "distributed": facts,
        }
        out["roofline"], out["conv_kernels"], out["roofline_critical"] = conv_kernel_roofline(step, w["run_diag"])
        if args.config == 2:
            out["roofline_raster"] = crop_kernel_roofline(w["render"], B)
        if args.config in (2, 3):
            out["roofline_geometry"] = geometry_rooflines(w["render"], B)
            out["fp32_mfma_path"] = same_step_on_fp32_mfma(step, w["tgt"], B)
        if "x6" in out["roofline"]["kernel"]:
            out["roofline"]["measured_mfma_ceiling"] = measured_mfma_ceiling()
            out["roofline"]["frac_of_measured_ceiling"] = round(
                out["roofline"]["achieved"] / out["roofline"]["measured_mfma_ceiling"]["fp32_equivalent_TFLOP/s"], 4)
        tf = w["flops_per_unit"] * images / dt / 1e12
        from dsf_amd import nn_conv as _nc
        step_peak, _ = matrix_peak("x6" if _nc.MATH == "x6" else "fp32")         # the matrix pipe the step's convolutions ran on
        out["whole_step_mfma"] = {"achieved": round(tf, 2), "peak": step_peak, "unit": "TFLOP/s", "frac": round(tf / step_peak, 4),
                                  "frac_of_fp32_mfma_peak": round(tf / FP32_MATRIX_PEAK_TFLOPS, 4),
                                  "note": "network FLOPs (%s) / whole step time, every kernel of the step included; peak = that of "
                                          "the active convolution math (DSF_CONV_MATH=%s): 2500 / 6 = 416.7 fp32-equivalent TFLOP/s for "
                                          "the split (x6) kernels, 157.3 for the fp32 MFMA (rounds 1-2 quoted frac against 157.3)"
                                          % (w["flops_note"], _nc.MATH)}
        if world == 1 and not args.no_cpu_baseline:
            cores = min(usable_cpus(), 32)
            torch.set_num_threads(cores)
            os.environ["OMP_NUM_THREADS"] = str(cores)
            ips, secs, n, what = w["cpu"](args.cpu_steps)
            out["cpu_baseline"] = {"value": round(ips, 3), "unit": "images/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
                                   "sample": "%d images: %s through the CPU oracle (torch-CPU trunk, C rasteriser / point-face "
                                             "distances, numpy crop chain), %.1f s" % (n, what, secs)}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
