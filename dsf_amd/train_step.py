"""Per-batch training steps of the hot path (counterparts of ``Trainer.Pretrain`` /
``Trainer.FinetuneStage``, /root/reference/train_render.py:415-488, 622-823) with every
host synchronisation removed: no ``.item()``, no ``.cpu()``, no data-dependent Python branch,
so a whole step can be enqueued asynchronously (and captured in a HIP graph).

Loss lists, weights, detach points and thresholds follow the reference; each block cites its
lines.  TensorBoard / cv2 drawing / ``xyz2error`` host metrics are out of scope.
"""
import contextlib
import math
import os

import numpy as np
import torch

from . import ops
from .metric.losses import SmoothL1Loss
from .metric.meshLoss import ICPLoss, JointICPLoss
from .render_model.render_loss import m2d_loss
from . import streams
from .streams import fork
from .util.generateFeature import GFM
from .data.render_loader import loader as TensorUtils


class Config:
    """Loss weights / feature settings of /root/reference/config.py:31-98."""
    stage_num = 2
    deconv_weight = 1
    coord_weight = 100
    model_weight = 1
    partICP_weight = 1
    M2P_weight = 1
    coll_weight = 1
    feature_type = ['offset']
    feature_para = [0.8]
    batch_size = 32
    input_size = 128
    lr = 0.001
    weight_decay = 0.01
    cube_size = [250, 250, 250]
    net = 'ResNet_stage_18'


def synthetic_batch(B, device, seed=0, dtype=torch.float32):
    """Synthetic NYU-shape inputs of SURVEY.md 8(d): 62-d MANO parameters
    [rot3|pose45|shape10|scale1|trans3], centre (U(+-40), U(+-40), U(500,1200)) mm, cube 250^3."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    p = torch.zeros(B, 62)
    p[:, :3] = (torch.rand(B, 3, generator=g) * 2 - 1) * math.pi
    p[:, 3:48] = torch.randn(B, 45, generator=g) * 0.5
    p[:, 48:58] = torch.randn(B, 10, generator=g) * 0.5
    p[:, 58] = 1.0
    center = torch.stack([torch.rand(B, generator=g) * 80 - 40, torch.rand(B, generator=g) * 80 - 40,
                          torch.rand(B, generator=g) * 700 + 500], 1)
    cube = torch.full((B, 3), 250.0)
    return p.to(device, dtype), center.to(device, dtype), cube.to(device, dtype)


def _mano_regularisers(mano_pd, w_beta, w_scale):
    """(mean(beta^2) * w_beta, mean(|min(scale, 0)|) * w_scale) of the packed MANO rows (train_render.py:463-464): one launch each
    way on the GPU (csrc/step_ops.hip: the backward writes the whole gradient row), the reference's expression elsewhere."""
    if mano_pd.is_cuda and mano_pd.dtype == torch.float32 and mano_pd.dim() == 2 and mano_pd.size(1) >= 59:
        both = ops.ManoReg.apply(mano_pd, 48, 58, float(w_beta), float(w_scale))
        return both[0], both[1]
    return (torch.mean(torch.pow(mano_pd[:, 48:58], 2)) * w_beta, torch.mean(torch.abs(torch.clamp(mano_pd[:, 58], max=0.0))) * w_scale)


def _stat_pool(step, net, applications=1):
    """The per-step pools around one forward + backward: zeroed BatchNorm accumulation rows (nn_norm.stat_pool: one zero fill
    per step lets every fused BatchNorm of ``net`` run without finalise launches; sized once per step object) and the zeroed
    outputs of the small split-K convolutions (nn_conv.zero_pool: sized by the previous step's demand)."""
    from . import nn_norm, nn_conv
    if not hasattr(step, "_stat_floats"):
        dev = next(net.parameters()).device
        step._stat_floats = nn_norm.stat_floats(net, applications) if dev.type == "cuda" else 0
        step._stat_dev = dev
    stack = contextlib.ExitStack()
    stack.enter_context(nn_norm.stat_pool(step._stat_floats, step._stat_dev))
    stack.enter_context(nn_conv.zero_pool(step, step._stat_dev))
    return stack


def _default_adamw(params, lr, weight_decay):
    """AdamW as the reference builds it (train_render.py:131-139): the one-launch HIP version
    (dsf_amd.optim.FusedAdamW; DSF_FUSED_ADAMW=0 selects torch.optim.AdamW's multi-tensor kernels for A/B runs)."""
    import os
    if os.environ.get("DSF_FUSED_ADAMW", "1") == "1":
        from .optim import FusedAdamW
        return FusedAdamW(params, lr=lr, weight_decay=weight_decay)
    return torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay)


S1_FORK = [os.environ.get("DSF_FT_S1_FORK", "1") == "1"]        # FinetuneStageStep.loss: the stage-1 student's loss chain beside stage 2's
SYN_FORK = [os.environ.get("DSF_FT_SYN_FORK", "1") == "1"]      # FinetuneStageStep.loss: the synthetic batch's loss chains beside the real batch's forward
FT_STREAMS = [os.environ.get("DSF_FT_STREAMS", "1") == "1"]      # FinetuneStageStep: forked chains on (see its __call__)


class RenderSupervisedStep:
    """BASELINE config 2: backbone forward (incl. the stage-2 re-render bridge) -> per stage the
    pixel-branch losses and the MANO-branch losses of ``Pretrain`` (train_render.py:444-466), plus
    ``Render.render`` of the final MANO estimate and the model-to-data depth term
    (:719-732) against the target depth image; backward; AdamW step."""

    def __init__(self, net, render, config=Config, optimizer=None, grad_sync=None):
        self.net, self.render, self.cfg = net, render, config
        self.L1 = SmoothL1Loss()
        self.gfm = GFM()
        self.opt = optimizer if optimizer is not None else _default_adamw(net.parameters(), lr=config.lr,
                                                                             weight_decay=config.weight_decay)
        self.grad_sync = grad_sync                    # dsf_amd.parallel.GradAllReducer or None
        self.utils = TensorUtils(img_size=config.input_size)

    @torch.no_grad()
    def make_targets(self, params_gt, center, cube, noise_sigma=0.02, seed=1):
        """'real' depth = render of the ground-truth parameters + Gaussian depth noise on the
        foreground; also the GT joints / mesh / offset maps."""
        img, juvd, jxyz, mesh = self.render.render(params_gt, center, cube)
        g = torch.Generator(device=img.device).manual_seed(seed)
        noise = torch.randn(img.shape, device=img.device, generator=g) * noise_sigma
        img = torch.where(img < 0.99, (img + noise).clamp(-1, 0.98), img)
        return {"img": img.contiguous(), "joint_uvd": juvd, "joint_xyz": jxyz, "mesh_xyz": mesh,
                "center": center, "cube": cube}

    def loss(self, tgt):
        cfg = self.cfg
        img, center, cube = tgt["img"], tgt["center"], tgt["cube"]
        outputs = self.net(img, self.render, center, cube)
        terms = {}
        # the model branch (MANO layer, rasteriser: short latency-bound launches, forward and backward) beside the pixel branch:
        # the decoder's backward pass only waits for the pixel terms (streams.py)
        f = fork(img.device)
        with f.branch(0, *[mano_pd for _, mano_pd in outputs]):
            for s, (_, mano_pd) in enumerate(outputs):
                # model branch (:459-466)
                jxyz_pd, mesh_pd = self.render.get_mesh_xyz(mano_pd)
                terms["joint%d" % s] = self.L1(jxyz_pd, tgt["joint_xyz"], weight=cfg.coord_weight)
                terms["vert%d" % s] = self.L1(mesh_pd, tgt["mesh_xyz"], weight=cfg.coord_weight)
                terms["beta%d" % s], terms["scale%d" % s] = _mano_regularisers(mano_pd, cfg.coord_weight * 10, 0.1)
            # render loss on the final estimate (:719, :728-732, :745)
            img_pd, _, _, _ = self.render.render(outputs[-1][1], center, cube)
            terms["m2d"] = m2d_loss(img, img_pd) * cfg.model_weight
        for s, (pixel_pd, _) in enumerate(outputs):
            S = pixel_pd.size(-1)
            # pixel-wise branch (:451-456); loss weights are folded into the fused Huber kernels
            pixel_gt = self.gfm.joint2feature(tgt["joint_uvd"], img, cfg.feature_para, S, cfg.feature_type)
            juvd_pd = self.gfm.feature2joint(img, pixel_pd, cfg.feature_type, cfg.feature_para)
            terms["pix%d" % s] = self.L1(pixel_pd, pixel_gt, weight=cfg.deconv_weight)
            terms["coord%d" % s] = self.L1(juvd_pd, tgt["joint_uvd"], weight=cfg.coord_weight)
        f.join()
        order = [k % s for s in range(len(outputs)) for k in ("pix%d", "coord%d", "joint%d", "vert%d", "beta%d", "scale%d")] + ["m2d"]
        terms = {k: terms[k] for k in order}                 # (the reference's order: the sum below is order-sensitive in its last bits)
        # one stack + one sum instead of a chain of scalar adds (and their backward kernels)
        return torch.stack(list(terms.values())).sum(), terms

    def forward_backward(self, tgt):
        """Everything of the step that runs on the device without host decisions (the part ``GraphedStep`` captures)."""
        from . import nn_conv
        self.opt.zero_grad(set_to_none=True)
        self.render.mano_layer.clear_cache()                 # results of the previous step must not outlive its graph
        if not hasattr(self, "_pool_floats"):
            self._pool_floats = nn_conv.weight_grad_floats(self.net) + 256          # fused heads re-lay one merged weight
            self._pool_dev = next(self.net.parameters()).device
        with _stat_pool(self, self.net):
            loss, terms = self.loss(tgt)
            with nn_conv.grad_pool(self._pool_floats if self._pool_dev.type == "cuda" else 0, self._pool_dev, reducer=self.grad_sync):
                loss.backward()
        return loss.detach(), terms

    def __call__(self, tgt):
        out = self.forward_backward(tgt)
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.opt.step()
        return out


class MeshLossStep:
    """BASELINE config 3: pixel network with a MANO head (``PoseNetMANO``) + the self-supervised
    geometry terms of FinetuneStage on a target depth map: m2d (:728-732), whole-hand ICP (:739),
    part-aware ICP (:735) with sphere segmentation (:695-701), sphere collision (:725), plus the
    supervised joint/vertex terms that keep the synthetic problem well-posed."""

    def __init__(self, net, render, config=Config, optimizer=None, grad_sync=None, n_points=2048):
        self.net, self.render, self.cfg, self.n_points = net, render, config, n_points
        self.L1 = SmoothL1Loss()
        self.opt = optimizer if optimizer is not None else _default_adamw(net.parameters(), lr=config.lr,
                                                                             weight_decay=config.weight_decay)
        self.grad_sync = grad_sync
        self.utils = TensorUtils(img_size=config.input_size)

    @torch.no_grad()
    def make_targets(self, params_gt, center, cube, seed=1, keys=None):
        """``keys``: the two (B, 128*128) int32 sampling-key images of Img2pcl as explicit inputs (else drawn from ``seed``)."""
        img, juvd, jxyz, mesh = self.render.render(params_gt, center, cube)
        _, M, _, _ = ops.crop_setup(center, cube, self.render.cam, 128)
        u = self.utils
        crop = u.crop_hand(img, jxyz, center, M, cube)
        mano = self.render.mano_layer
        B = img.size(0)
        g = torch.Generator(device=img.device).manual_seed(seed)
        given = iter(keys) if keys is not None else None
        keys = (lambda: next(given)) if given is not None else \
            (lambda: torch.randint(0, 2 ** 31 - 1, (B, 128 * 128), device=img.device, dtype=torch.int32, generator=g))
        _, pts = u.crop_hand(crop, jxyz, center, M, cube, return_points=True)
        seg_img = mano.seg_pcl(jxyz, jxyz, mesh, pts)                                    # (:695)
        seg_img = torch.where(crop.lt(0.99).reshape(B, -1), seg_img, torch.zeros_like(seg_img)).reshape(B, 1, 128, 128)
        joint_img = torch.where(seg_img.gt(0), crop, torch.ones_like(crop))              # (:698)
        joint_pcl = u.Img2pcl(joint_img, 128, center, M, cube, self.n_points, rand_keys=keys())   # (:699)
        seg = mano.seg_pcl(jxyz, jxyz, mesh, joint_pcl)                                  # (:700)
        pcl = u.Img2pcl(crop, 128, center, M, cube, self.n_points, rand_keys=keys())     # (:701)
        return {"img": img, "crop": crop, "joint_xyz": jxyz, "mesh_xyz": mesh, "joint_pcl": joint_pcl, "seg": seg,
                "pcl": pcl, "center": center, "cube": cube, "M": M}

    def loss(self, tgt):
        cfg = self.cfg
        mano_layer = self.render.mano_layer
        _, mano_pd = self.net(tgt["img"])
        img_pd, juvd, jxyz, mesh = self.render.render(mano_pd, tgt["center"], tgt["cube"])
        # four independent chains behind the MANO layer (streams.py): the point-to-triangle search fills the chip, the others
        # are a few short launches each -- beside it instead of behind it, forward and backward
        f = fork(mesh.device)
        with f.branch(0, img_pd):
            crop_pd = self.utils.crop_hand(img_pd, tgt["joint_xyz"], tgt["center"], tgt["M"], tgt["cube"])
            l_m2d = m2d_loss(tgt["crop"], crop_pd) * cfg.model_weight
        with f.branch(1, mesh):
            l_part = JointICPLoss(mesh, tgt["joint_pcl"], mano_layer.joint_faces, tgt["seg"]).mean(-1).mean(-1) * cfg.partICP_weight
        with f.branch(2, mesh, jxyz):
            l_coll = mano_layer.calculate_coll(jxyz, mesh.detach()) * cfg.coll_weight
            l_sup = (self.L1(jxyz, tgt["joint_xyz"]) + self.L1(mesh, tgt["mesh_xyz"])) * cfg.coord_weight
        l_icp = ICPLoss(mesh, tgt["pcl"], mano_layer.faces).mean(-1) * cfg.model_weight
        f.join()
        terms = {"m2d": l_m2d, "pd2m": l_part, "d2m": l_icp, "coll": l_coll, "sup": l_sup}
        return l_m2d + l_part + l_icp + l_coll + l_sup, terms

    def forward_backward(self, tgt):
        from . import nn_conv
        self.opt.zero_grad(set_to_none=True)
        self.render.mano_layer.clear_cache()                 # results of the previous step must not outlive its graph
        if not hasattr(self, "_pool_floats"):                # one zero fill per step for the ~90 weight gradients of the
            self._pool_floats = nn_conv.weight_grad_floats(self.net) + 256          # hourglass instead of one launch each
            self._pool_dev = next(self.net.parameters()).device
        with _stat_pool(self, self.net):
            loss, terms = self.loss(tgt)
            with nn_conv.grad_pool(self._pool_floats if self._pool_dev.type == "cuda" else 0, self._pool_dev, reducer=self.grad_sync):
                loss.backward()
        return loss.detach(), terms

    def __call__(self, tgt):
        out = self.forward_backward(tgt)
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.opt.step()
        return out


def _graph_node_types(raw_graph):
    """{hipGraphNodeType: count} of a captured graph (0 kernel, 1 memcpy, 2 memset, ...) through the HIP runtime."""
    import collections
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_size_t(0)
    if hip.hipGraphGetNodes(ctypes.c_void_p(raw_graph), None, ctypes.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    nodes = (ctypes.c_void_p * max(n.value, 1))()
    if n.value and hip.hipGraphGetNodes(ctypes.c_void_p(raw_graph), nodes, ctypes.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    types = collections.Counter()
    for i in range(n.value):
        ty = ctypes.c_int(-1)
        if hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(ty)) != 0:
            raise RuntimeError("hipGraphNodeGetType failed")
        types[ty.value] += 1
    return dict(types)


class GraphedStep:
    """``forward_backward`` of a step (RenderSupervisedStep, MeshLossStep) captured ONCE in a HIP graph and replayed per
    batch: the ~700 launches of a step are issued by the driver from one graph launch instead of by the Python host, which
    removes the host-issue gaps between the many small kernels (GPU busy 92 % -> 99 %).  The same
    kernels run on the same data in the same order -- results are those of the eager step (tests/test_gpu_steps.py).
    Building the wrapper does not train: the warm-up and validation passes run forward + backward only and restore the
    BatchNorm running statistics.  What stays eager, after each replay: the optimizer step, whose learning rate, bias corrections and per-parameter step
    counts are host state that changes from step to step -- and, with a GradAllReducer, the bucketed gradient all-reduce
    (launched after the replay, not overlapped with the backward pass).
    Requirements, as for any stream capture: static shapes (one graph per batch shape), no host decision inside the step
    (the occluder count of ``FinetuneStageStep`` is one -- that step is not graphable), batches are COPIED into the static
    input buffers the graph reads.  The reference has no counterpart (PyTorch eager, train_render.py:636-823)."""

    def __init__(self, step, tgt, warmup=2, validate=True):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs the GPU (HIP graph capture)")
        # data parallel: the graph holds forward + backward only (captured with the reducer's hooks switched off); the bucket
        # all-reduces are launched eagerly after each replay (GradAllReducer.reduce_now), then the optimizer
        # every refusal below is checked BEFORE the reducer is touched: a caller that catches the error and steps eagerly must find
        # its GradAllReducer as it left it (hooks on), else the replicas would train without any all-reduce
        if not hasattr(step, "forward_backward"):
            raise TypeError("GraphedStep needs a step with forward_backward(tgt) (RenderSupervisedStep, MeshLossStep); the "
                            "steps that draw their occluder count on the host cannot be captured")
        self.sync = getattr(step, "grad_sync", None)
        if self.sync is not None:
            from .nn_norm import FusedSyncBatchNorm2d
            if any(isinstance(m, FusedSyncBatchNorm2d) for m in step.net.modules()):
                raise RuntimeError("GraphedStep: a cross-replica BatchNorm all-reduces inside the forward pass; collectives are "
                                   "not captured -- run this step eagerly")
        self.step = step
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in tgt.items()}
        from .nn_norm import FusedBatchNorm2d
        self._bns = [m for m in step.net.modules() if isinstance(m, FusedBatchNorm2d)]
        # eager warm-up on a side stream (allocator, lazy tables, weight images): forward + backward only -- no optimizer step --
        # and the BatchNorm running statistics / batch counts are put back, so that building the wrapper does not train
        with self._hooks_off():
            self._build(step, warmup, validate)

    @contextlib.contextmanager
    def _hooks_off(self):
        """GradAllReducer hooks off for GraphedStep's own passes (warm-up, capture, validation: forward + backward without a
        collective), whatever happens inside; the reducer's ``enabled`` flag is the caller's everywhere else."""
        if self.sync is None:
            yield
            return
        was = self.sync.enabled
        self.sync.enabled = False
        try:
            yield
        finally:
            self.sync.enabled = was

    def _build(self, step, warmup, validate):
        bufs = [(b, b.clone()) for b in step.net.buffers()]
        pend = [m._pending_batches for m in self._bns]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step.forward_backward(self.static)
            with torch.no_grad():
                for b, c in bufs:
                    b.copy_(c)
        torch.cuda.current_stream().wait_stream(side)
        for m, n in zip(self._bns, pend):
            m._pending_batches = n
            m.__dict__["_stats_epoch"] = m.__dict__.get("_stats_epoch", 0) + 1
        del bufs
        before = [m._pending_batches for m in self._bns]
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(self.graph):
            self.loss, self.terms = step.forward_backward(self.static)
        self.node_types = _graph_node_types(self.graph.raw_cuda_graph())
        if self.node_types.get(2, 0):
            # hipGraphNodeTypeMemset: does not replay correctly on ROCm 7.2 (tools/graph_memset.py).  csrc/ has none
            # (tests/test_library_abi.py); torch's multi-block reductions (a large .mean()/.sum() to few outputs) zero
            # their semaphores with one -- such a step captures fine and then replays wrong numbers at some later step.
            raise RuntimeError("GraphedStep: the captured step holds %d memset node(s), which this ROCm runtime does not "
                               "replay correctly (a torch multi-block reduction issues one); run this step eagerly or "
                               "reduce in two stages (model/hourglass.py::GlobalAvgPool2d)" % self.node_types[2])
        self.graph.instantiate()
        # host-side bookkeeping the captured region did once (BatchNorm's deferred num_batches_tracked) is repeated per replay
        self._bn_calls = [m._pending_batches - b for m, b in zip(self._bns, before)]
        for m, b in zip(self._bns, before):
            m._pending_batches = b                          # the capture itself ran no kernel
        self._grads = [(p, p.grad) for p in step.net.parameters() if p.grad is not None]   # the tensors the graph writes
        if validate:
            self._validate()

    def _validate(self):
        """Two replays from the current state must reproduce the eager step (loss terms and gradients; bitwise in
        deterministic mode), else the graph is refused.  Why this is not paranoia: on ROCm 7.2 a memset node does not
        replay correctly (tools/graph_memset.py) -- this library has none (tests/test_library_abi.py), but torch's own
        multi-block reductions zero their semaphores with hipMemsetAsync, and a step that contains one captures fine and
        then replays wrong numbers.  The network state (BatchNorm statistics) is restored afterwards."""
        from . import _lib as L
        net = self.step.net
        bufs = [(b, b.clone()) for b in net.buffers()]
        pend = [m._pending_batches for m in self._bns]

        def restore():
            with torch.no_grad():
                for b, c in bufs:
                    b.copy_(c)
            for m, n in zip(self._bns, pend):
                m._pending_batches = n
        _, terms = self.step.forward_backward(self.static)                    # eager, same state, same inputs
        ref_t = torch.stack([v.detach().float().reshape(()) for v in terms.values()])
        ref_g = {p: p.grad.detach().clone() for p, _ in self._grads if p.grad is not None}
        restore()
        exact = L.deterministic()
        for p, g in self._grads:
            p.grad = g                                                         # back to the tensors the graph writes
        try:
            for r in range(2):
                self.graph.replay()
                got_t = torch.stack([v.detach().float().reshape(()) for v in self.terms.values()])
                if exact:
                    ok = torch.equal(got_t, ref_t) and all(torch.equal(p.grad, ref_g[p]) for p, _ in self._grads if p in ref_g)
                else:
                    num = sum(((p.grad - ref_g[p]).double() ** 2).sum() for p, _ in self._grads if p in ref_g)
                    den = sum((ref_g[p].double() ** 2).sum() for p, _ in self._grads if p in ref_g)
                    ok = bool(((got_t - ref_t).abs() <= 2e-3 * ref_t.abs() + 1e-6).all()) and float(num) <= (2e-2 ** 2) * float(den)
                restore()
                if not ok:
                    raise RuntimeError(
                        "GraphedStep: replay %d of the captured step does not reproduce the eager step (terms %s vs %s). A "
                        "captured hipMemsetAsync does not replay on this ROCm (torch's multi-block reductions issue one); run "
                        "this step eagerly." % (r + 1, [round(float(v), 6) for v in got_t], [round(float(v), 6) for v in ref_t]))
        finally:
            restore()

    def __call__(self, tgt=None):
        if tgt is not None and tgt is not self.static:
            for k, v in tgt.items():
                if torch.is_tensor(v):
                    self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        for m, n in zip(self._bns, self._bn_calls):
            m._pending_batches += n
            m.__dict__["_stats_epoch"] = m.__dict__.get("_stats_epoch", 0) + 1        # the replay rewrote the running statistics
        if self.sync is not None:
            for p, g in self._grads:
                p.grad = g                                  # the tensors the graph wrote (finish() rebinds .grad to bucket views)
            was = self.sync.enabled
            self.sync.enabled = True                        # reduce_now is a no-op on a disabled reducer
            try:
                self.sync.reduce_now()
            finally:
                self.sync.enabled = was                     # the wrapped step stays usable eagerly (its hooks as the caller set them)
        self.step.opt.step()
        return self.loss, self.terms


def draw_augmentation(B, device, generator=None, host_rng=None, n_joints=21, crop=128, views=1, depth_range=(500, 1200),
                      view_scale=1.0, mask=True):
    """Every random draw one synthetic-branch pass consumes, as explicit tensors, in the order the reference
    draws them: train_render.py:628-631 (shape N(0,3^2), centre U(+-20 mm), size U(0.8,1.2), view U(0,2pi)^3 * view_scale),
    mano_layer.py:1007 (centre depth U(depth_range)), :1328-1334 (mask_img: 3..9 occluders drawn on the HOST -- numpy in
    the reference, ``host_rng`` here, so no device sync --, their joints, uvd offsets U(+-0.15) and radii U(0,0.3)), plus
    the two sampling-key images of ``Img2pcl`` (render_loader.py:1151-1154).  ``views`` > 1 draws for views*B renders."""
    g = generator if generator is not None else torch.Generator(device=device)
    rng = host_rng if host_rng is not None else np.random.default_rng()
    n = B * views
    rnd = lambda *s: torch.rand(*s, device=device, generator=g)
    d = {"aug_shape": torch.randn(n, 10, device=device, generator=g) * 3,
         "aug_center": (rnd(n, 3) - 0.5) * 40,
         "aug_size": 1 + (rnd(n, 1) - 0.5) * 0.4,
         "aug_view": rnd(n, 3) * math.pi * 2 * view_scale}
    depth = rnd(n, 1) * (depth_range[1] - depth_range[0]) + depth_range[0]
    d["center0"] = torch.cat((torch.zeros(n, 2, device=device), depth), dim=-1)
    if mask:
        k = int(rng.choice(np.arange(3, 10), 1)[0])                                   # mano_layer.py:1328 (min 3, max 10 exclusive)
        jid = torch.as_tensor(rng.choice(np.arange(0, n_joints), k, replace=False).astype(np.int64))
        d["mask_joint_id"] = jid.to(device)
        d["mask_offset"] = (rnd(n, k, 3) - 0.5) * 0.15 * 2
        d["mask_radius"] = rnd(n, k) * 0.3
    keys = lambda: torch.randint(0, 2 ** 31 - 1, (B, crop * crop), device=device, dtype=torch.int32, generator=g)
    d["keys_joint"], d["keys_pcl"] = keys(), keys()
    return d


def draws_to(d, device):
    return {k: v.to(device) for k, v in d.items()}


class _StepBase:
    """What the step classes share: Huber / GFM / loader utilities, AdamW as the reference builds it, the optional
    gradient all-reducer, and the synthetic-branch pass of ``Pretrain`` / ``Finetune`` / ``FinetuneStage``."""

    def __init__(self, net, render, transfer_net=None, config=Config, optimizer=None, grad_sync=None, mask=True):
        self.net, self.render, self.transfer, self.cfg, self.mask = net, render, transfer_net, config, mask
        self.L1 = SmoothL1Loss()
        self.gfm = GFM()
        self.opt = optimizer if optimizer is not None else _default_adamw(net.parameters(), lr=config.lr,
                                                                             weight_decay=config.weight_decay)
        self.grad_sync = grad_sync
        self.utils = TensorUtils(img_size=config.input_size)
        if transfer_net is not None:
            for p in transfer_net.parameters():
                p.requires_grad_(False)
            transfer_net.eval()
            from . import nn_conv
            nn_conv.manage_weights(transfer_net.parameters())       # frozen: its split weight images are kept across steps

    def draw(self, B, device, generator=None, host_rng=None, views=1, view_scale=1.0):
        return draw_augmentation(B, device, generator, host_rng, views=views, depth_range=self.render.depth_range,
                                 view_scale=view_scale, mask=self.mask)

    @torch.no_grad()
    def synth(self, model_para, cube, d):
        """``RenderNet(model_para, None, cube, augment..., mask)`` + the frozen transfer generator
        (train_render.py:428-435 / 502-509 / 633-639) -> dict of the 8-tuple + the transferred image."""
        R = self.render
        img, juvd, vuvd, jxyz, vxyz, center, cube_s, M = R(model_para, d["center0"], cube, augmentView=d["aug_view"],
                                                          augmentShape=d["aug_shape"], augmentCenter=d["aug_center"],
                                                          augmentSize=d["aug_size"], mask=False)
        if self.mask and "mask_joint_id" in d:
            img = R.mask_img(img, juvd, 0.15, 0.3, draws=(d["mask_joint_id"], d["mask_offset"], d["mask_radius"]))
        img_t = self.transfer(img) if self.transfer is not None else img
        return {"img": img, "img_t": img_t, "joint_uvd": juvd, "joint_xyz": jxyz, "mesh_xyz": vxyz, "center": center,
                "cube": cube_s, "M": M}

    def _real_targets(self, img_src, crop_r, jxyz_pix, jxyz_mano, mesh_mano, center_r, M_r, cube_r, d):
        """Part labels and the two point clouds of the real image (train_render.py:561-575 / 693-701): ``img_src`` is the
        image the part pixels are taken from (``img_r`` in Finetune :566, the crop in FinetuneStage :698)."""
        u, mano_layer = self.utils, self.render.mano_layer
        B = crop_r.size(0)
        _, pts = u.crop_hand(crop_r, jxyz_mano, center_r, M_r, cube_r, return_points=True)       # uvdImg2xyzImg(crop) (:561 / :693)
        seg_img = mano_layer.seg_pcl(jxyz_pix, jxyz_mano, mesh_mano, pts)
        seg_img = torch.where(crop_r.lt(0.99).reshape(B, -1), seg_img, torch.zeros_like(seg_img)).reshape(B, 1, 128, 128)
        joint_img = torch.where(seg_img.gt(0), img_src, torch.ones_like(img_src))
        joint_pcl = u.Img2pcl(joint_img, 128, center_r, M_r, cube_r, 2048, rand_keys=d["keys_joint"])
        segment = mano_layer.seg_pcl(jxyz_pix, jxyz_mano, mesh_mano, joint_pcl)
        pcl = u.Img2pcl(crop_r, 128, center_r, M_r, cube_r, 2048, rand_keys=d["keys_pcl"])
        return joint_pcl, segment, pcl, seg_img

    def _masked_huber(self, a, b, row_mask):
        """L1Loss(index_select(a, rows), index_select(b, rows)) with the reference's empty rule (:600-603 / :798-801):
        the host branch ``joint_mano_mask.sum() == 0`` tests the SUM OF THE SELECTED INDICES, evaluated here on device."""
        z = (a - b).float()
        az = z.abs()
        per_row = torch.where(az < 0.01, 0.5 * z * z, 0.01 * (az - 0.005)).mean(-1)          # (rows,)
        m = row_mask.to(per_row.dtype)
        n = m.sum()
        idx_sum = (torch.arange(m.numel(), device=m.device, dtype=per_row.dtype) * m).sum()
        val = (per_row * m).sum() / torch.clamp(n, min=1.0)
        return torch.where(idx_sum == 0, torch.zeros_like(val), val)

    def _m2p(self, juvd_pix, juvd_mano, mano_ok, pd2m_j):
        """M2P selection (:590-603 / :787-801): samples whose render and ICP agree with the data teach the pixel branch,
        joint by joint (part distance < 1e-3; wrist always; the five tips follow joints 2,5,8,11,14)."""
        B = juvd_pix.size(0)
        if juvd_pix.is_cuda and B > 0 and juvd_pix.shape[1:] == (21, 3) and pd2m_j.shape == (B, 15) and juvd_pix.dtype == torch.float32:
            # selection + masked Huber + the reference's empty rule in one launch each way (csrc/step_ops.hip)
            return ops.M2P.apply(juvd_pix, juvd_mano.detach(), mano_ok.detach(), pd2m_j.detach(), float(self.cfg.coord_weight))
        jm = pd2m_j.lt(1e-3)
        jm = torch.cat((torch.ones(B, 1, device=jm.device, dtype=torch.bool), jm, jm[:, [2, 5, 8, 11, 14]]), dim=-1)
        rows = (mano_ok.unsqueeze(-1) & jm).detach().reshape(-1)
        return self._masked_huber(juvd_pix.reshape(-1, 3), juvd_mano.detach().reshape(-1, 3), rows) * self.cfg.coord_weight

    def _optimise(self, loss):
        from . import nn_conv
        if not hasattr(self, "_pool_floats"):
            self._pool_floats = nn_conv.weight_grad_floats(self.net) + 256
            self._pool_dev = next(self.net.parameters()).device
        with nn_conv.grad_pool(self._pool_floats if self._pool_dev.type == "cuda" else 0, self._pool_dev, reducer=self.grad_sync):
            loss.backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.opt.step()

    def _begin(self):
        self.opt.zero_grad(set_to_none=True)
        self.render.mano_layer.clear_cache()                 # results of the previous step must not outlive its graph


class PretrainStep(_StepBase):
    """Counterpart of ``Trainer.Pretrain`` (train_render.py:415-488): B synthetic hands rendered through ``Render.forward``
    with shape / centre / size augmentation and random occluders, the frozen transfer generator, then per stage the
    pixel-branch and MANO-branch supervised losses.  ``views`` > 1 is BASELINE config 4: every sample is rendered from
    ``views`` random ``augmentView`` rotations (views*B meshes through the rasteriser and the backbone); the reference's
    own Pretrain multiplies its view draw by 0 (:424), ``view_scale=0`` reproduces that."""

    def __init__(self, net, render, transfer_net=None, config=Config, optimizer=None, grad_sync=None, mask=True, views=1,
                 view_scale=None):
        super().__init__(net, render, transfer_net, config, optimizer, grad_sync, mask)
        self.views = views
        self.view_scale = (0.0 if views == 1 else 1.0) if view_scale is None else view_scale

    def draw(self, B, device, generator=None, host_rng=None):
        return super().draw(B, device, generator, host_rng, views=self.views, view_scale=self.view_scale)

    def loss(self, model_para, cube, draws=None):
        cfg, R, L1, gfm = self.cfg, self.render, self.L1, self.gfm
        d = draws if draws is not None else self.draw(model_para.size(0), model_para.device)
        if self.views > 1:
            model_para = model_para.repeat_interleave(self.views, dim=0)
            cube = cube.repeat_interleave(self.views, dim=0)
        s = self.synth(model_para, cube, d)
        img = s["img"]
        outputs = self.net(s["img_t"], R, s["center"], s["cube"])
        terms = {}
        f = fork(img.device)                                  # the model branch beside the pixel branch, as in RenderSupervisedStep.loss
        with f.branch(0, *[mano_pd for _, mano_pd in outputs]):
            for i, (_, mano_pd) in enumerate(outputs):
                jxyz_pd, mesh_pd = R.get_mesh_xyz(mano_pd)
                terms["joint%d" % i] = L1(jxyz_pd, s["joint_xyz"], weight=cfg.coord_weight)
                terms["vert%d" % i] = L1(mesh_pd, s["mesh_xyz"], weight=cfg.coord_weight)
                terms["beta%d" % i], terms["scale%d" % i] = _mano_regularisers(mano_pd, cfg.coord_weight * 10, 0.1)
        for i, (pixel_pd, _) in enumerate(outputs):
            S = pixel_pd.size(-1)
            pixel_gt = gfm.joint2feature(s["joint_uvd"], img, cfg.feature_para, S, cfg.feature_type)
            juvd_pd = gfm.feature2joint(img, pixel_pd, cfg.feature_type, cfg.feature_para)
            terms["pix%d" % i] = L1(pixel_pd, pixel_gt, weight=cfg.deconv_weight)
            terms["coord%d" % i] = L1(juvd_pd, s["joint_uvd"], weight=cfg.coord_weight)
        f.join()
        order = [k % i for i in range(len(outputs)) for k in ("pix%d", "coord%d", "joint%d", "vert%d", "beta%d", "scale%d")]
        terms = {k: terms[k] for k in order}                 # (the reference's order: the sum below is order-sensitive in its last bits)
        return torch.stack(list(terms.values())).sum(), terms

    def __call__(self, model_para, cube, draws=None):
        self._begin()
        with _stat_pool(self, self.net):
            loss, terms = self.loss(model_para, cube, draws)
            self._optimise(loss)
        return loss.detach(), terms


class FinetuneStep(_StepBase):
    """Counterpart of ``Trainer.Finetune`` (train_render.py:490-620), the single-stage self-supervised step: a synthetic
    supervised pass (stage-1 outputs only) + a real-image pass whose MANO branch is fitted to the data by the model-to-data
    depth term, ICP and part-aware ICP, tied to the pixel branch by P2M / M2P, with the collision prior."""

    def loss(self, model_para, cube, img_r, center_r, cube_r, M_r, draws=None):
        cfg, R, u, gfm, L1 = self.cfg, self.render, self.utils, self.gfm, self.L1
        mano_layer = R.mano_layer
        B = img_r.size(0)
        d = draws if draws is not None else self.draw(model_para.size(0), model_para.device)
        # ---- synthetic image (:496-527): stage-1 outputs only ----
        s = self.synth(model_para, cube, d)
        img = s["img"]
        pixel_pd, mano_pd = self.net(s["img_t"], R, s["center"], s["cube"])[0]
        pixel_gt = gfm.joint2feature(s["joint_uvd"], img, cfg.feature_para, pixel_pd.size(-1), cfg.feature_type)
        juvd_pd = gfm.feature2joint(img, pixel_pd, cfg.feature_type, cfg.feature_para)
        total = L1(pixel_pd[:, :pixel_gt.size(1)], pixel_gt) * cfg.deconv_weight + L1(juvd_pd, s["joint_uvd"]) * cfg.coord_weight
        jx, mx = R.get_mesh_xyz(mano_pd)
        total = total + L1(mx, s["mesh_xyz"]) * cfg.coord_weight + L1(jx, s["joint_xyz"]) * cfg.coord_weight \
            + mano_layer.calculate_coll(jx, mx.detach()) * cfg.coll_weight
        # ---- real image (:530-610) ----
        pixel_r, mano_r = self.net(img_r, R, center_r, cube_r)[0]
        juvd_r = gfm.feature2joint(img_r, pixel_r, cfg.feature_type, cfg.feature_para)
        jxyz_r = u.uvd_nl2xyznl_tensor(juvd_r, center_r, M_r, cube_r)
        img_m, mjuvd, mjxyz, mesh = R.render(mano_r, center_r, cube_r)
        coll = mano_layer.calculate_coll(mjxyz, mesh.detach())
        crop_r = u.crop_hand(img_r, mjxyz.detach(), center_r, M_r, cube_r)                     # (:554)
        crop_m = u.crop_hand(img_m, mjxyz.detach(), center_r, M_r, cube_r)
        union = (crop_r.lt(0.99) | crop_m.lt(0.99)).float()
        m2d = (torch.abs(crop_r - crop_m).mean(-1).mean(-1) / (union.mean(-1).mean(-1) + 1e-8)).mean()      # (:556-558)
        with torch.no_grad():
            joint_pcl, segment, pcl, _ = self._real_targets(img_r, crop_r, jxyz_r, mjxyz, mesh, center_r, M_r, cube_r, d)
        pd2m_j = JointICPLoss(mesh, joint_pcl, mano_layer.joint_faces, segment)               # (:570-572)
        d2m_b = ICPLoss(mesh, pcl, mano_layer.faces)                                           # (:576-577)
        p2m = L1(mjuvd, juvd_r.detach()) * cfg.coord_weight                                    # (:583)
        both = (crop_r.lt(0.95) & img_m.lt(0.95)).float()                                      # (:586-589): the UNcropped render
        depth_b = (torch.abs(crop_r - img_m) * both).sum(-1).sum(-1) / both.sum(-1).sum(-1)   # 0/0 = NaN fails lt(): as the reference
        mano_ok = depth_b.lt(0.04).squeeze(-1) & d2m_b.lt(1e-3)
        m2p = self._m2p(juvd_r, mjuvd, mano_ok, pd2m_j)
        terms = {"m2d": m2d, "pd2m": pd2m_j.mean(-1).mean(-1), "P2M": p2m, "coll": coll, "M2P": m2p, "d2m": d2m_b.mean(-1)}
        total = total + p2m + m2d * 0.1 * cfg.model_weight + terms["d2m"] * cfg.model_weight + terms["pd2m"] * cfg.partICP_weight \
            + m2p * cfg.M2P_weight + coll * cfg.coll_weight
        return total, terms

    def __call__(self, model_para, cube, img_r, center_r, cube_r, M_r, draws=None):
        self._begin()
        with _stat_pool(self, self.net, applications=2):     # the network sees the synthetic and the real batch
            loss, terms = self.loss(model_para, cube, img_r, center_r, cube_r, M_r, draws)
            self._optimise(loss)
        return loss.detach(), terms


class FinetuneStageStep(_StepBase):
    """Counterpart of ``Trainer.FinetuneStage`` (train_render.py:622-823), the reference's default
    self-boosting step (config.py:36-38): a synthetic supervised pass through the frozen transfer
    generator + a real-image pass where the detached stage-2 outputs teach stage 1 and the geometry
    terms (m2d, ICP, part-aware ICP, collision, P2M, M2P) close the self-supervised loop.

    Differences from the reference, all forced by "no host sync inside the step":
      * random draws are explicit (``draw()`` / ``draws=``; the occluder COUNT 3..9 and the occluded joints come from a host
        RNG exactly as in the reference, which uses numpy there -- no device value is read);
      * the M2P term (:784-801) selects rows with ``nonzero()`` + ``index_select`` and branches on
        ``joint_mano_mask.sum() == 0`` on the host; here it is the algebraically identical masked mean
        (same rows, same divisor, same "sum of selected indices == 0 -> 0" rule), evaluated on device;
      * ``xyz2error`` / TensorBoard / colour LUT host round trips (:654-667, :703, :713-721) are dropped.
    """

    def __init__(self, net, render, transfer_net, config=Config, optimizer=None, grad_sync=None, mask=True):
        super().__init__(net, render, transfer_net, config, optimizer, grad_sync, mask)

    def loss(self, model_para, cube, img_r, center_r, cube_r, M_r, generator=None, draws=None):
        cfg, R, u, gfm, L1 = self.cfg, self.render, self.utils, self.gfm, self.L1
        mano_layer = R.mano_layer
        dev = model_para.device
        B = img_r.size(0)
        d = draws if draws is not None else self.draw(model_para.size(0), dev, generator)
        # ---- synthetic branch (:628-667) ----
        s = self.synth(model_para, cube, d)
        img, juvd_gt, jxyz_gt, mesh_gt = s["img"], s["joint_uvd"], s["joint_xyz"], s["mesh_xyz"]
        outputs = self.net(s["img_t"], R, center=s["center"], cube=s["cube"])
        # every loss term with its weight folded in (a weight of 1 multiplies nothing); ONE stack + sum at the end instead of a
        # chain of ~35 scalar multiplies and adds, each with a backward launch of its own
        acc = []
        w = lambda t, k: t if k == 1 else t * k
        # the synthetic batch's loss chains (encode / decode / Huber / MANO layer / collision: ~20 short launches, 64 workgroups each)
        # on the branch stream, beside the network's pass over the REAL batch that the host issues next (DSF_FT_SYN_FORK=0: in front of it)
        f_syn = fork(dev) if SYN_FORK[0] else None
        with (f_syn.branch(0, *[t for o in outputs for t in o]) if f_syn is not None else contextlib.nullcontext()):
            for pixel_pd, mano_pd in outputs:
                S = pixel_pd.size(-1)
                pixel_gt = gfm.joint2feature(juvd_gt, img, cfg.feature_para, S, cfg.feature_type)
                juvd = gfm.feature2joint(img, pixel_pd, cfg.feature_type, cfg.feature_para)
                acc += [L1(pixel_pd, pixel_gt, weight=cfg.deconv_weight), L1(juvd, juvd_gt, weight=cfg.coord_weight)]
                jx, mx = R.get_mesh_xyz(mano_pd)
                acc += [L1(jx, jxyz_gt, weight=cfg.coord_weight), L1(mx, mesh_gt, weight=cfg.coord_weight),
                        w(mano_layer.calculate_coll(jx, mx.detach()), cfg.coll_weight)]
        # ---- real branch: teacher from the detached stage-2 outputs (:671-703) ----
        outputs = self.net(img_r, R, center=center_r, cube=cube_r)
        if f_syn is not None:
            f_syn.join()
        pix_t, mano_t = outputs[1][0].detach(), outputs[1][1].detach()
        with torch.no_grad():
            juvd_t = gfm.feature2joint(img_r, pix_t, cfg.feature_type, cfg.feature_para)
            jxyz_t = u.uvd_nl2xyznl_tensor(juvd_t, center_r, M_r, cube_r)
            mj_t, mm_t = R.get_mesh_xyz(mano_t)
            crop_r = u.crop_hand(img_r, mj_t, center_r, M_r, cube_r)
            joint_pcl, segment, pcl, _ = self._real_targets(crop_r, crop_r, jxyz_t, mj_t, mm_t, center_r, M_r, cube_r, d)
        # ---- stage 1 student (:706-749) ----  (its chain of short launches on the branch stream, beside stage 2's on this one:
        # DSF_FT_S1_FORK=0 keeps them one after the other)
        pix1, mano1 = outputs[0]
        f_s1 = fork(dev) if S1_FORK[0] else None
        with (f_s1.branch(0, pix1, mano1, pix_t, juvd_t, jxyz_t, mj_t, mm_t, crop_r, joint_pcl, segment, pcl) if f_s1 is not None
              else contextlib.nullcontext()):
            juvd1 = gfm.feature2joint(img_r, pix1, cfg.feature_type, cfg.feature_para)
            acc += [L1(pix1, pix_t, weight=cfg.deconv_weight), L1(juvd1, juvd_t, weight=cfg.coord_weight)]
            img1, mjuvd1, mjxyz1, mesh1 = R.render(mano1, center_r, cube_r)
            acc += [L1(mjxyz1, jxyz_t, weight=cfg.coord_weight), L1(mesh1, mm_t, weight=cfg.coord_weight),
                    w(mano_layer.calculate_coll(mjxyz1, mesh1.detach()), cfg.coll_weight)]
            crop1 = u.crop_hand(img1, mj_t, center_r, M_r, cube_r)
            acc += [w(m2d_loss(crop_r, crop1), cfg.model_weight), w(ICPLoss(mesh1, pcl, mano_layer.faces).mean(-1), cfg.model_weight),
                    w(JointICPLoss(mesh1, joint_pcl, mano_layer.joint_faces, segment).mean(-1).mean(-1), cfg.partICP_weight)]
        # ---- stage 2 (:752-808) ----
        pix2, mano2 = outputs[1]
        juvd2 = gfm.feature2joint(img_r, pix2, cfg.feature_type, cfg.feature_para)
        img2, mjuvd2, mjxyz2, mesh2 = R.render(mano2, center_r, cube_r)
        p2m = L1(mjuvd2, juvd_t, weight=cfg.coord_weight)
        coll2 = mano_layer.calculate_coll(mjxyz2, mesh2.detach())
        crop2 = u.crop_hand(img2, mj_t, center_r, M_r, cube_r)
        pd2m_j = JointICPLoss(mesh2, joint_pcl, mano_layer.joint_faces, segment)
        d2m_b = ICPLoss(mesh2, pcl, mano_layer.faces)
        fused = ops.m2d(crop_r, crop2) if crop_r.is_cuda else None
        if fused is not None:
            # the model-to-data term and the agreement sums of the M2P gate from ONE reduction (csrc/step_ops.hip): sums =
            # {sum |d| union, sum union, sum |d| both, sum both} per sample
            m2d2, sums, _ = fused
            depth_b = sums[:, 2] / (sums[:, 1] + 1e-8)
        else:
            union = (crop_r.lt(0.99) | crop2.lt(0.99)).float()
            m2d2 = m2d_loss(crop_r, crop2)
            both = (crop_r.lt(0.99) & crop2.lt(0.99)).float()
            depth_b = (((crop_r - crop2).abs() * both).sum(-1).sum(-1) / (union.sum(-1).sum(-1) + 1e-8)).squeeze(-1)
        mano_ok = depth_b.lt(0.04) & d2m_b.lt(1e-3)                                          # (:787-789)
        m2p = self._m2p(juvd2, mjuvd2, mano_ok, pd2m_j)
        d2m, pd2m = d2m_b.mean(-1), pd2m_j.mean(-1).mean(-1)
        acc += [p2m, w(coll2, cfg.coll_weight), w(m2d2, cfg.model_weight), w(d2m, cfg.model_weight), w(pd2m, cfg.partICP_weight),
                w(m2p, cfg.M2P_weight)]
        if f_s1 is not None:
            f_s1.join()
        total = torch.stack([t.reshape(()) for t in acc]).sum()
        terms = {"P2M": p2m, "m2d": m2d2, "d2m": d2m, "pd2m": pd2m, "M2P": m2p, "coll": coll2}
        return total, terms

    def __call__(self, model_para, cube, img_r, center_r, cube_r, M_r, generator=None, draws=None):
        self._begin()
        # (the forked chains of the network -- stage-2 bridge, MANO heads beside the decoders -- did not pay in this step through round 5
        #  and most of round 6, whose network runs twice per pass beside the frozen generator: 87.3 ms with them off, 88.0 on; with the
        #  MANO head of _run_trunk on the branch stream they do: 70.89 -> 70.16 ms, same box, 8 alternating blocks of 5.  DSF_FT_STREAMS=0:
        #  one stream)
        one_stream = streams.disabled() if not FT_STREAMS[0] else contextlib.nullcontext()
        with _stat_pool(self, self.net, applications=2), one_stream:     # the network sees the synthetic and the real batch
            loss, terms = self.loss(model_para, cube, img_r, center_r, cube_r, M_r, generator, draws)
            self._optimise(loss)
        return loss.detach(), terms
