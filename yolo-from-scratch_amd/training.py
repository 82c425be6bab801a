"""Training step on the HIP path: fused trainer (no autograd, no per-step host sync), the HipAdam
optimizer facade, data-parallel gradient averaging over RCCL, and the reference's train_epoch /
eval_epoch signatures (train.py:888-1032).

One training step = NCHW->NHWC, forward op list, fused loss kernel (writes d loss / d head outputs),
backward op list (split at gradient-bucket boundaries when data-parallel so each bucket's all-reduce
runs on RCCL's stream under the remaining backward kernels), global-norm + clip + Adam over flat
buffers.  Loss scalars stay on the device; callers read them when they want to.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

from datetime import timedelta

import torch
import torch.distributed as dist

from . import _lib as L
from .functional import LOSS_IMG_SIZE, _anchors18, run_loss_kernel, yolo_loss_multiscale
from .graph import WEIGHTS_EPOCH, PARAM_GENERATION, invalidate_folded_weights
from .hostside import stack_targets
from .modules import HipModule


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def check_targets(targets, shapes, device):
    """The fused step hands raw device pointers to yh_yolo_loss with the batch / grid sizes of the plan, so the targets
    must match them exactly -- the reference raises a shape error in its loss for the same mistakes (train.py:806-830)."""
    if len(targets) != len(shapes):
        raise ValueError(f"expected {len(shapes)} target tensors (one per scale), got {len(targets)}")
    out = []
    for s, (t, want) in enumerate(zip(targets, shapes)):
        if not isinstance(t, torch.Tensor) or t.device.type != "cuda" or t.device != torch.device(device):
            raise ValueError(f"target[{s}] must live on {device} (got {getattr(t, 'device', type(t))}); no CPU fallback")
        if t.dtype != torch.float32:
            raise ValueError(f"target[{s}] must be float32, got {t.dtype}")
        if tuple(t.shape) != tuple(want):
            raise ValueError(f"target[{s}] has shape {tuple(t.shape)}, the step expects {tuple(want)} "
                             "(batch, grid, grid, 3, 5 + num_classes of the model / input batch)")
        out.append(t.contiguous())
    return out


class CollectiveError(RuntimeError):
    """A data-parallel collective failed or timed out; the rank must exit (non-zero), not retry."""


class GradBuckets:
    """Data-parallel gradient exchange over a flat gradient buffer (torch.distributed only: RCCL on the
    GPUs, gloo in the CPU tests).  The buffer is cut into contiguous buckets (parameter registration
    order); each bucket knows after how many backward ops its gradients are final, so its all-reduce
    (SUM; the 1/world factor is folded into the clip+Adam kernel) can be issued while the rest of the
    backward still runs.  `plan_segments` returns [(n_ops_done, (begin, end) | None), ...]."""

    def __init__(self, flat_g: torch.Tensor, process_group=None, n_buckets: int = 4, collectives_at_world_1: bool = False):
        """collectives_at_world_1: issue the bucketed collectives even when the group has ONE rank (an all-reduce over one
        rank is the identity, so the step must stay bitwise equal to the non-distributed one): the whole RCCL leg -- lazy
        communicator creation, side-stream ordering against the backward segments, wait -- can then be exercised on a
        one-GPU box (tests/test_gpu_dp.py)."""
        self.flat_g, self.pg, self.n_buckets = flat_g, process_group, max(1, n_buckets)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (bool(collectives_at_world_1) and dist.is_initialized())
        self._works = []

    def plan_segments(self, spans, ready, n_ops):
        """spans: [(offset, padded_numel)] per parameter in registration order; ready: op count after
        which each parameter's gradient is final; n_ops: length of the backward op list."""
        if not self.active:
            return [(n_ops, None)]
        total = self.flat_g.numel()
        target = total / self.n_buckets
        buckets, start, rdy = [], 0, 0
        for (off, n), r in zip(spans, ready):
            end = off + n
            rdy = max(rdy, r)
            if end - start >= target and len(buckets) < self.n_buckets - 1:
                buckets.append((rdy, (start, end)))
                start, rdy = end, 0
        if start < total:
            buckets.append((rdy, (start, total)))
        buckets.sort(key=lambda b: b[0])          # earliest-complete first (heads, neck, then backbone)
        if buckets[-1][0] < n_ops:
            buckets.append((n_ops, None))
        return buckets

    def launch(self, rng):
        if self.active and rng is not None:
            try:
                self._works.append(dist.all_reduce(self.flat_g[rng[0]:rng[1]], op=dist.ReduceOp.SUM, group=self.pg,
                                                   async_op=True))
            except Exception as e:                     # a communicator that is already in an error state refuses new work
                self._works = []
                raise CollectiveError(f"gradient all-reduce could not be enqueued: {e}") from e

    def wait(self, timeout_s: Optional[float] = None):
        """Join the outstanding all-reduces.  A failed or timed-out collective (a dead rank, an RCCL asynchronous error) is
        raised as CollectiveError instead of hanging the surviving ranks in wait() -- SURVEY section 5: RCCL async error -> abort.
        The caller lets it propagate: the process ends with a non-zero exit code (never a re-exec), the launcher tears the job
        down.  timeout_s: None = the process group's own timeout (init_process_group(timeout=...))."""
        works, self._works = self._works, []
        for w in works:
            try:
                ok = w.wait(timedelta(seconds=timeout_s)) if timeout_s is not None else w.wait()
            except Exception as e:
                raise CollectiveError(f"gradient all-reduce failed: {e}") from e
            if ok is False:
                raise CollectiveError(f"gradient all-reduce did not complete within {timeout_s} s")


class HipTrainer:
    """Owns the flat parameter / gradient / Adam-state buffers of a YOLO model and runs fused steps."""

    def __init__(self, model: HipModule, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm: Optional[float] = 10.0,
                 process_group=None, n_buckets: int = 4, dtype: Optional[str] = None, collectives_at_world_1: bool = False):
        """dtype: None keeps the model's compute dtype; 'bf16' / 'f32' set it (HipModule.set_compute_dtype).
        collectives_at_world_1: see GradBuckets (test hook: run broadcast + bucketed all-reduces on a 1-rank group)."""
        self.model = model
        if dtype is not None:
            model.set_compute_dtype(dtype)
        params = list(model.parameters())
        if not params or not params[0].is_cuda:
            raise RuntimeError("HipTrainer: move the model to the GPU first; there is no CPU fallback")
        L.lib()
        self.device = params[0].device
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        self.step_count = 0
        views = model._grad_views(self.device)           # flat gradient buffer, 4-float padded per tensor
        self.flat_g = model._flat_grad
        n = self.flat_g.numel()
        self.flat_p = torch.zeros(n, device=self.device, dtype=torch.float32)
        self.offsets = {}
        off = 0
        with torch.no_grad():
            for p in params:
                dst = self.flat_p[off:off + p.numel()].view_as(p)
                dst.copy_(p.data)
                p.data = dst                              # parameters now live in the flat buffer
                p.grad = views[id(p)]                     # and their .grad in the flat gradient buffer
                self.offsets[id(p)] = (off, p.numel())
                off += (p.numel() + 3) // 4 * 4
        PARAM_GENERATION[0] += 1                          # every parameter moved: plans re-trace, hipGraphs re-capture
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.norm = torch.zeros(1, device=self.device, dtype=torch.float32)
        self.norm_ws = torch.empty(int(L.lib().yh_sqnorm_ws(n)) + 2, device=self.device, dtype=torch.float64)
        self.loss_out = torch.zeros(13, device=self.device, dtype=torch.float32)
        self._loss_ws = None
        self.buckets = GradBuckets(self.flat_g, process_group, n_buckets, collectives_at_world_1)
        self.pg, self.world = process_group, self.buckets.world
        self._segments = None
        self._params = params
        if self.buckets.active:
            dist.broadcast(self.flat_p, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                           group=process_group)      # identical replicas
            invalidate_folded_weights()              # written through the flat view: torch's version counters of the parameters did not move

    @torch.no_grad()
    def load_flat_parameters(self, flat: torch.Tensor):
        """Overwrite all parameters from a flat fp32 vector laid out like `flat_p` (resume, EMA swap, parameter server):
        the one sanctioned way to write through the flat buffer -- it also invalidates folded inference weights."""
        if flat.numel() != self.flat_p.numel():
            raise ValueError(f"expected {self.flat_p.numel()} values, got {flat.numel()}")
        self.flat_p.copy_(flat.reshape(-1).to(self.flat_p))
        invalidate_folded_weights()

    def _plan_segments(self, plan):
        spans = [(self.offsets[id(p)][0], (self.offsets[id(p)][1] + 3) // 4 * 4) for p in self._params]
        ready = [plan.grad_ready[id(p)] for p in self._params]
        return self.buckets.plan_segments(spans, ready, plan.bwd_ops[1])

    # ---- one step -------------------------------------------------------------------------------
    def step(self, imgs: torch.Tensor, targets: Sequence[torch.Tensor]) -> torch.Tensor:
        """imgs (B,3,S,S) fp32 on the device, targets = three (B,G,G,3,5+nc) device tensors.
        Returns the device tensor [total, box, obj, cls, ...per-scale] of this step (no host sync)."""
        model = self.model
        if not model.training:
            model.train()
        st = _stream(self.device)
        plan = model._plan_for(imgs)
        if self._segments is None or self._segments[0] is not plan:
            self._segments = (plan, self._plan_segments(plan))
            grids = [v.H for v, _ in plan.outputs]
            self._loss_ws = torch.empty(int(L.lib().yh_loss_ws(L.int3(grids), plan.B)) + 8, device=self.device,
                                        dtype=torch.float32)
        heads = [v for v, _ in plan.outputs]
        nc = model.num_classes
        targets = check_targets(targets, [(plan.B, v.H, v.W, 3, 5 + nc) for v in heads], self.device)
        model._load_input(plan, imgs)
        plan.run_forward(st)
        run_loss_kernel([v.buf.data for v in heads], targets, [v.buf.grad for v in heads],
                        _anchors18(model.anchors), [v.H for v in heads], plan.B, nc, None, None, self.loss_out,
                        self._loss_ws, st, dpred_bf16=plan.bf16, dpred_ld=[v.ldg for v in heads])
        begin = 0
        for end, rng in self._segments[1]:
            plan.run_backward(st, begin, end)       # yh_run joins its side stream before returning
            begin = end
            self.buckets.launch(rng)                # RCCL runs on its own stream, under the next segment
        self.buckets.wait()
        self.apply_update()
        return self.loss_out

    def apply_update(self):
        """Global-norm clip (max_norm) + Adam over the flat buffers; averages over ranks when DP."""
        st = _stream(self.device)
        lib = L.lib()
        n = self.flat_g.numel()
        gscale = 1.0 / self.world
        self.step_count += 1
        WEIGHTS_EPOCH[0] += 1               # the fused kernel rewrites the parameters in place (folded eval weights go stale)
        clip = self.max_norm is not None and self.max_norm > 0
        if clip:
            L.check(lib.yh_grad_sqnorm(self.flat_g.data_ptr(), n, gscale, self.norm.data_ptr(), self.norm_ws.data_ptr(), st),
                    "grad_sqnorm")
        L.check(lib.yh_adam_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), n,
                                 float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                 self.step_count, float(self.max_norm) if clip else 0.0,
                                 self.norm.data_ptr() if clip else None, gscale, st), "adam_step")


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Optimizer facade over HipTrainer: `HipAdam(model, lr=...)` behaves like
    `optim.Adam(model.parameters(), lr=...)` (train.py:1506) but steps through the fused HIP kernel
    on flat buffers; `max_norm` folds clip_grad_norm_(…, 10.0) (train.py:916) into the same launch."""

    def __init__(self, model: HipModule, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm: Optional[float] = None,
                 process_group=None):
        self.trainer = HipTrainer(model, lr, betas, eps, max_norm, process_group)
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps))

    def zero_grad(self, set_to_none: bool = True):
        self.trainer.flat_g.zero_()          # gradients live in one flat buffer; keep the views alive

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        self.trainer.lr, self.trainer.betas, self.trainer.eps = g["lr"], g["betas"], g["eps"]
        self.trainer.apply_update()

    def fused_step(self, imgs, targets):
        g = self.param_groups[0]
        self.trainer.lr, self.trainer.betas, self.trainer.eps = g["lr"], g["betas"], g["eps"]
        return self.trainer.step(imgs, targets)


def train_epoch(model, loader, optimizer, device, num_classes=1):
    """One epoch; returns mean (total, box, obj, cls) over batches (train.py:888-926).  With a HipAdam
    optimizer every batch is one fused HIP step (clip 10.0 + Adam included) and the loss scalars are
    read back once per epoch; with any other torch optimizer the same kernels run under autograd and
    the reference's clip_grad_norm_(10.0) / optimizer.step() sequence is followed literally."""
    model.train()
    anchors_list = model.anchors
    n = 0
    if isinstance(optimizer, HipAdam):
        if optimizer.trainer.max_norm is None:
            optimizer.trainer.max_norm = 10.0
        acc = torch.zeros(4, device=device, dtype=torch.float64)
        for imgs, targets in loader:
            out = optimizer.fused_step(imgs.to(device, non_blocking=True), stack_targets(targets, device))
            acc += out[:4].double()
            n += 1
        vals = (acc / max(n, 1)).tolist()
        return vals[0], vals[1], vals[2], vals[3]
    tot = [0.0, 0.0, 0.0, 0.0]
    for imgs, targets in loader:
        imgs = imgs.to(device)
        tb = stack_targets(targets, device)
        optimizer.zero_grad()
        preds = model(imgs)
        loss, lb, lo, lc = yolo_loss_multiscale(preds, tb, anchors_list, num_classes)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=10.0)
        optimizer.step()
        for k, v in enumerate((loss, lb, lo, lc)):
            tot[k] += v.item()
        n += 1
    return tuple(t / n for t in tot)


def eval_epoch(model, loader, device, num_classes=1, iou_threshold=0.5, conf_threshold=0.5):
    """Validation loss + same-cell/same-anchor precision, recall, F1 in percent (train.py:960-1032).
    The reference walks every cell in Python with two `.item()` per cell; here one HIP kernel per batch
    (yh_eval_counts) applies the identical counting rule and the three counters are read once per epoch."""
    model.eval()
    anchors_list = model.anchors
    a18 = _anchors18(anchors_list)
    counts = torch.zeros(3, device=device, dtype=torch.int64)
    loss_acc = torch.zeros((), device=device, dtype=torch.float64)
    nb = 0
    with torch.no_grad():
        for imgs, targets in loader:
            imgs = imgs.to(device)
            tb = [t.contiguous() for t in stack_targets(targets, device)]
            preds = [p.contiguous() for p in model(imgs)]
            loss_acc += yolo_loss_multiscale(preds, tb, anchors_list, num_classes)[0].double()
            nb += 1
            L.check(L.lib().yh_eval_counts(L.ptr3(preds), L.ptr3(tb), L.floats(a18), L.int3([p.shape[1] for p in preds]),
                                           preds[0].shape[0], num_classes, LOSS_IMG_SIZE, float(conf_threshold),
                                           float(iou_threshold), counts.data_ptr(), _stream(device)), "eval_counts")
    tp, fp, fn = (int(v) for v in counts.tolist())
    prec = tp / (tp + fp) if tp + fp > 0 else 0
    rec = tp / (tp + fn) if tp + fn > 0 else 0
    f1 = 2 * prec * rec / (prec + rec) if prec + rec > 0 else 0
    return float(loss_acc) / max(nb, 1), prec * 100, rec * 100, f1 * 100
