"""Host-side mirror of the reference's model classes (train.py:224-632) on the HIP path.

The classes keep the reference's constructor signatures, attribute names, child registration order
and therefore its state-dict keys and its seeded initial weights (nn.Conv2d / nn.BatchNorm2d objects
are used as parameter containers only -- their own forward is never called).  `forward` traces the
module into a static NHWC plan (graph.py) once per input shape / mode and then replays it through
libyolohip; gradients flow back through one autograd node per call.

Interface parity: NCHW fp32 in; ConvBlock/C3/Bottleneck/SPPF return NCHW, YOLO returns the three
(B,G,G,3,5+nc) tensors (train.py:632).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L
from .graph import Plan, View, PARAM_GENERATION, invalidate_folded_weights

DEFAULT_ANCHORS = [[[10, 13], [16, 30], [33, 23]], [[30, 61], [62, 45], [59, 119]],
                   [[116, 90], [156, 198], [373, 326]]]          # train.py:372-374


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class _PlanRunner(torch.autograd.Function):
    """One autograd node for a whole traced module: forward = yh_run(fwd list), backward = yh_run(bwd list)."""

    @staticmethod
    def forward(ctx, host, plan: Plan, x: torch.Tensor, *params):
        host._load_input(plan, x)
        plan.run_forward(_stream(x.device))
        ctx.host, ctx.plan, ctx.gen = host, plan, plan.generation
        ctx.x_needs_grad = x.requires_grad
        outs = host._collect_outputs(plan)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        plan: Plan = ctx.plan
        if plan.generation != ctx.gen:
            raise RuntimeError("backward through a HIP plan whose activations were overwritten by a later forward "
                               "(call backward before the next forward of the same module/shape)")
        host = ctx.host
        host._load_output_grads(plan, gouts)
        plan.run_backward(_stream(plan.device))
        gx = host._input_grad(plan) if ctx.x_needs_grad else None
        grads = []
        for p in plan.params():
            v = host._flat_views[id(p)]
            # a trainer may have pointed p.grad at the flat buffer: the ops already wrote it in place
            aliased = p.grad is not None and p.grad.data_ptr() == v.data_ptr()
            grads.append(None if aliased else v.clone())
        return (None, None, gx, *grads)


class HipModule(nn.Module):
    """Base class: plan cache, flat gradient buffer, NCHW <-> NHWC boundary conversions."""

    # Plans kept per module, least recently used first out.  Every plan owns a full set of activation + gradient buffers (~10 GB at
    # batch 64, 640x640): a ragged last batch, a second resolution or an eval pass must not pin another one forever.
    # An evicted plan is only DROPPED from the cache: an autograd node that still needs it for its backward keeps it (and its
    # buffers) alive until then.  (Plans do not share an activation arena: a training plan's buffers live from its forward to
    # its backward, and another plan may run in between.)
    PLAN_CACHE_MAX = 3

    def _hip_init(self):
        self._plans: "OrderedDict[tuple, Plan]" = OrderedDict()
        self._flat_grad: Optional[torch.Tensor] = None
        self._flat_views: Dict[int, torch.Tensor] = {}
        self._compute_dtype = "f32"

    def set_compute_dtype(self, dtype: str):
        """'f32' (default: the reference's arithmetic) or 'bf16' (BASELINE configs 3-4): TRAINING plans then keep
        activations, activation gradients and per-step weight packs in bf16 and run the convolutions on the bf16 MFMA;
        master weights, accumulators, BatchNorm statistics, the loss and all parameter gradients stay fp32.  Eval-mode
        forwards always run the fp32 fused kernels.  The module interface (fp32 NCHW in / fp32 out) does not change."""
        if dtype not in ("f32", "bf16"):
            raise ValueError(f"compute dtype must be 'f32' or 'bf16', got {dtype!r}")
        self._compute_dtype = dtype
        return self

    # -- weight-state bookkeeping (plans and captured hipGraphs hold raw addresses and folded copies) ---
    def invalidate_folded_weights(self):
        """Call after writing parameter / BatchNorm-buffer storage in a way torch's version counters of the registered
        tensors do not see (`p.data.mul_()`, writes through `trainer.flat_p`, in-place collectives): eval plans and
        InferenceSessions then re-fold BatchNorm before their next forward.  See graph.invalidate_folded_weights."""
        invalidate_folded_weights()
        return self

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)        # .to() / .cuda() / .float(): storage may have moved
        PARAM_GENERATION[0] += 1
        invalidate_folded_weights()
        return out

    def load_state_dict(self, state_dict, strict=True, assign=False):
        out = super().load_state_dict(state_dict, strict=strict, assign=assign)
        PARAM_GENERATION[0] += 1                 # assign=True replaces the tensors; values changed either way
        invalidate_folded_weights()
        return out

    # -- to be provided by subclasses ------------------------------------------------------------
    def _emit(self, g: Plan, x: View, out: Optional[View] = None) -> View:
        raise NotImplementedError

    def _trace(self, g: Plan):
        y = self._emit(g, g.input.view())
        g.mark_output(y, "nchw")

    # -- plan management ---------------------------------------------------------------------------
    def _grad_views(self, device) -> Dict[int, torch.Tensor]:
        params = list(self.parameters())
        n = sum((p.numel() + 3) // 4 * 4 for p in params)
        if self._flat_grad is None or self._flat_grad.device != device or self._flat_grad.numel() != n \
                or set(self._flat_views) != {id(p) for p in params}:
            self._flat_grad = torch.zeros(n, device=device, dtype=torch.float32)
            self._flat_views, off = {}, 0
            for p in params:
                self._flat_views[id(p)] = self._flat_grad[off:off + p.numel()].view_as(p)
                off += (p.numel() + 3) // 4 * 4
            self._plans.clear()
        return self._flat_views

    def _plan_for(self, x: torch.Tensor) -> Plan:
        if not x.is_cuda:
            raise RuntimeError(f"{type(self).__name__}: the HIP path needs a GPU tensor (got {x.device}); "
                               "there is no CPU fallback in this package")
        if x.dim() != 4 or x.dtype not in (torch.float32, torch.uint8):
            raise ValueError("expected an NCHW float32 batch (or a (B,H,W,3) uint8 image batch)")
        L.lib()   # fail loudly if the extension is missing
        need_dx = bool(x.requires_grad and torch.is_grad_enabled())
        training = bool(self.training)
        shape = tuple(x.shape) if x.dtype == torch.float32 else (x.shape[0], x.shape[3], x.shape[1], x.shape[2])
        dtype = self._compute_dtype if training else "f32"
        key = (shape, training, need_dx, x.device.index, dtype)
        views = self._grad_views(x.device)
        plan = self._plans.get(key)
        if plan is not None and plan.params_moved():
            plan = None
        if plan is None:
            self._plans.pop(key, None)
            while len(self._plans) >= self.PLAN_CACHE_MAX:
                self._plans.popitem(last=False)          # least recently used
            plan = Plan(x.device, shape, training, need_dx, dtype)
            self._trace(plan)
            plan.compile(views if training else None)
            self._plans[key] = plan
        else:
            self._plans.move_to_end(key)
        return plan

    def forward(self, x: torch.Tensor):
        plan = self._plan_for(x)
        if torch.is_grad_enabled() and plan.training:
            outs = _PlanRunner.apply(self, plan, x, *plan.params())
        else:
            self._load_input(plan, x)
            plan.run_forward(_stream(x.device))
            outs = self._collect_outputs(plan)
        return self._package(list(outs))

    def _package(self, outs: List[torch.Tensor]):
        return outs[0]

    # -- boundary conversions ----------------------------------------------------------------------
    def _load_input(self, plan: Plan, x: torch.Tensor):
        x = x.contiguous()
        buf = plan.input
        lib = L.lib()
        if x.dtype == torch.uint8:      # (B,H,W,3) image bytes straight from the loader: /255 on the device
            B, H, W, C = x.shape
            fn = lib.yh_bf16_u8hwc_to_nhwc if plan.bf16 else lib.yh_u8hwc_to_nhwc
            L.check(fn(x.data_ptr(), buf.data.data_ptr(), B, H, W, C, buf.C, buf.C, _stream(x.device)), "u8hwc_to_nhwc")
            return
        B, C, H, W = x.shape
        fn = lib.yh_bf16_nchw_to_nhwc if plan.bf16 else lib.yh_nchw_to_nhwc
        L.check(fn(x.data_ptr(), buf.data.data_ptr(), B, C, H, W, buf.C, buf.C, _stream(x.device)), "nchw_to_nhwc")

    def _collect_outputs(self, plan: Plan) -> List[torch.Tensor]:
        outs = []
        for v, kind in plan.outputs:
            if kind == "nchw":
                t = torch.empty(v.B, v.C, v.H, v.W, device=plan.device, dtype=torch.float32)
                fn = L.lib().yh_bf16_nhwc_to_nchw if v.buf.dtype == torch.bfloat16 else L.lib().yh_nhwc_to_nchw
                L.check(fn(v.ptr(), t.data_ptr(), v.B, v.C, v.H, v.W, v.ld, 0, _stream(plan.device)), "nhwc_to_nchw")
            else:   # head output: the NHWC buffer *is* (B,G,G,3,5+nc)
                t = v.buf.data.view(v.B, v.H, v.W, 3, v.C // 3).clone()
            outs.append(t)
        return outs

    def _load_output_grads(self, plan: Plan, gouts):
        st = _stream(plan.device)
        for (v, kind), g in zip(plan.outputs, gouts):
            if g is None:
                L.check(L.lib().yh_memset(v.buf.grad.data_ptr(), 0, v.buf.grad.numel() * v.buf.grad.element_size(), st), "memset")
                continue
            g = g.contiguous().float()
            if kind == "nchw":
                if v.off != 0 or v.ld != v.C:
                    raise NotImplementedError("output gradient into a channel slice")
                fn = L.lib().yh_bf16_nchw_to_nhwc if plan.bf16 else L.lib().yh_nchw_to_nhwc
                L.check(fn(g.data_ptr(), v.buf.grad.data_ptr(), v.B, v.C, v.H, v.W, v.ldg, v.C, st), "nchw_to_nhwc(grad)")
            else:       # head gradient: (B,G,G,3,5+nc) -> the buffer's (possibly padded, possibly bf16) pixel rows
                v.buf.grad[..., :v.C].copy_(g.reshape(v.B, v.H, v.W, v.C))

    def _input_grad(self, plan: Plan) -> torch.Tensor:
        buf = plan.input
        gx = torch.empty(plan.B, plan.Cimg, plan.Himg, plan.Wimg, device=plan.device, dtype=torch.float32)
        fn = L.lib().yh_bf16_nhwc_to_nchw if plan.bf16 else L.lib().yh_nhwc_to_nchw
        L.check(fn(buf.grad.data_ptr(), gx.data_ptr(), plan.B, plan.Cimg, plan.Himg, plan.Wimg, buf.grad_C, 0,
                   _stream(plan.device)), "nhwc_to_nchw(grad)")
        return gx


# ------------------------------------------------------------------------------------------------
class ConvBlock(HipModule):
    """Conv2d(bias=False) -> BatchNorm2d -> SiLU (train.py:253-265)."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0):
        super().__init__()
        self._hip_init()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = nn.SiLU()

    def _emit(self, g, x, out=None, residual=None, upsample=False):
        return g.conv(x, self.conv, self.bn, out=out, residual=residual, upsample=upsample)


class Bottleneck(HipModule):
    """x + ConvBlock3x3(ConvBlock3x3(x)) when shortcut and channels match (train.py:295-306)."""

    def __init__(self, in_channels, out_channels, shortcut=True):
        super().__init__()
        self._hip_init()
        self.conv1 = ConvBlock(in_channels, out_channels, 3, 1, 1)
        self.conv2 = ConvBlock(out_channels, out_channels, 3, 1, 1)
        self.shortcut = shortcut and in_channels == out_channels

    def _emit(self, g, x, out=None):
        h = self.conv1._emit(g, x)
        return self.conv2._emit(g, h, out=out, residual=x if self.shortcut else None)   # add fused in the BN pass


class C3(HipModule):
    """conv3(cat[bottlenecks(conv1 x), conv2 x]) with hidden = out//2 (train.py:267-293)."""

    def __init__(self, in_channels, out_channels, n=1, shortcut=True):
        super().__init__()
        self._hip_init()
        hidden = out_channels // 2
        self.conv1 = ConvBlock(in_channels, hidden, 1, 1, 0)
        self.conv2 = ConvBlock(in_channels, hidden, 1, 1, 0)
        self.conv3 = ConvBlock(2 * hidden, out_channels, 1, 1, 0)
        self.bottlenecks = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut) for _ in range(n)])
        self._hidden = hidden

    def _emit(self, g, x, out=None, side_branch=False):
        """side_branch: run the conv2 branch on the side lane next to conv1 -> bottlenecks (only where no other
        long-running side-lane work is pending, since the join waits for everything on that lane)."""
        hid = self._hidden
        if hid % 4:
            raise NotImplementedError("C3 hidden channels must be a multiple of 4 on the HIP path")
        Ho, Wo = x.H, x.W
        cat = g.new_buffer(x.B, Ho, Wo, 2 * hid, "c3cat")
        nb = len(self.bottlenecks)
        if side_branch:
            # conv2 only needs x: trace it first on the side lane, then the main chain, then join
            with g.side_lane():
                self.conv2._emit(g, x, out=cat.view(hid, hid))
            a = self.conv1._emit(g, x, out=cat.view(0, hid) if nb == 0 else None)
            for i, b in enumerate(self.bottlenecks):
                a = b._emit(g, a, out=cat.view(0, hid) if i == nb - 1 else None)
            g.join()
            return self.conv3._emit(g, cat.view(), out=out)
        a = self.conv1._emit(g, x, out=cat.view(0, hid) if nb == 0 else None)
        for i, b in enumerate(self.bottlenecks):
            a = b._emit(g, a, out=cat.view(0, hid) if i == nb - 1 else None)
        self.conv2._emit(g, x, out=cat.view(hid, hid))
        return self.conv3._emit(g, cat.view(), out=out)


class SPPF(HipModule):
    """1x1 conv+BN+SiLU, three chained 5x5 max-pools, concat, 1x1 conv+BN+SiLU (train.py:224-251)."""

    def __init__(self, in_channels, out_channels, kernel_size=5):
        super().__init__()
        self._hip_init()
        if kernel_size != 5:
            raise NotImplementedError("the HIP SPPF path implements the reference's 5x5 pooling")
        hidden = in_channels // 2
        self.conv1 = nn.Conv2d(in_channels, hidden, 1, 1)
        self.bn1 = nn.BatchNorm2d(hidden)
        self.act = nn.SiLU()
        self.maxpool = nn.MaxPool2d(kernel_size=kernel_size, stride=1, padding=kernel_size // 2)
        self.conv2 = nn.Conv2d(hidden * 4, out_channels, 1, 1)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self._hidden = hidden

    def _emit(self, g, x, out=None):
        h = self._hidden
        if h % 4:
            raise NotImplementedError("SPPF hidden channels must be a multiple of 4 on the HIP path")
        cat = g.new_buffer(x.B, x.H, x.W, 4 * h, "sppfcat")
        g.conv(x, self.conv1, self.bn1, out=cat.view(0, h))
        for i in range(3):
            g.pool5(cat.view(i * h, h), cat.view((i + 1) * h, h))
        return g.conv(cat.view(), self.conv2, self.bn2, out=out)


def _seq_c3(g, seq_item, x, out=None):
    return seq_item._emit(g, x, out=out)


class YOLO(HipModule):
    """Backbone + FPN + PANet + three heads (train.py:308-632); same buffers, keys and defaults."""

    def __init__(self, num_classes=1, anchors=None, img_size=640, width_mult=0.50, depth_mult=0.33):
        super().__init__()
        self._hip_init()
        self.num_classes, self.img_size = num_classes, img_size
        self.width_mult, self.depth_mult = width_mult, depth_mult
        ch = lambda base: int(np.ceil(base * width_mult / 8) * 8)             # train.py:345-347
        rep = lambda n: max(round(n * depth_mult), 1) if n > 1 else n           # train.py:349-351
        c1, c3, c4, c5 = ch(64), ch(128), ch(256), ch(512)
        self.grid_size_p3, self.grid_size_p4, self.grid_size_p5 = img_size // 8, img_size // 16, img_size // 32
        self.grid_size = self.grid_size_p5
        self.register_buffer("strides", torch.tensor([8, 16, 32], dtype=torch.float32))
        if anchors is None:
            sets = DEFAULT_ANCHORS
        else:
            sets = anchors if isinstance(anchors[0][0], list) else [anchors] * 3
        for i, a in enumerate(sets):
            self.register_buffer(f"anchors_p{i + 3}", torch.tensor(a, dtype=torch.float32))
        self.num_anchors = 3
        for lvl, gsz in ((3, self.grid_size_p3), (4, self.grid_size_p4), (5, self.grid_size_p5)):
            ar = torch.arange(gsz, dtype=torch.float32)
            # contiguous copies (the reference registers stride-0 views, which break load_state_dict: quirk Q5)
            self.register_buffer(f"grid_x_p{lvl}", ar.view(1, 1, gsz, 1).expand(1, gsz, gsz, 1).contiguous())
            self.register_buffer(f"grid_y_p{lvl}", ar.view(1, gsz, 1, 1).expand(1, gsz, gsz, 1).contiguous())
        self.output_channels = self.num_anchors * (5 + num_classes)

        def down(cin, cout):   # inline Conv2d(bias=True), BatchNorm2d, SiLU triple
            return [nn.Conv2d(cin, cout, 3, 2, 1), nn.BatchNorm2d(cout), nn.SiLU()]

        self.stem = nn.Sequential(*down(3, c1 // 2), *down(c1 // 2, c1))
        self.backbone_p3 = nn.Sequential(C3(c1, c1, n=rep(1)), *down(c1, c3), C3(c3, c3, n=rep(2)))
        self.backbone_p4 = nn.Sequential(*down(c3, c4), C3(c4, c4, n=rep(2)))
        self.backbone_p5 = nn.Sequential(*down(c4, c5), C3(c5, c5, n=rep(1)))
        self.sppf = SPPF(c5, c5)
        self.lateral_p4 = ConvBlock(c4, c4, 1, 1, 0)
        self.lateral_p3 = ConvBlock(c3, c3, 1, 1, 0)
        self.upsample_p5_to_p4 = nn.Upsample(scale_factor=2, mode="nearest")
        self.reduce_p5_for_p4 = ConvBlock(c5, c4, 1, 1, 0)
        self.merge_p4 = C3(c4 * 2, c4, n=rep(1))
        self.upsample_p4_to_p3 = nn.Upsample(scale_factor=2, mode="nearest")
        self.reduce_p4_for_p3 = ConvBlock(c4, c3, 1, 1, 0)
        self.merge_p3 = C3(c3 * 2, c3, n=rep(1))
        self.downsample_p3_to_p4 = ConvBlock(c3, c3, 3, 2, 1)
        self.panet_merge_p4 = C3(c3 + c4, c4, n=rep(1))
        self.downsample_p4_to_p5 = ConvBlock(c4, c4, 3, 2, 1)
        self.panet_merge_p5 = C3(c4 + c5, c5, n=rep(1))

        def head(c):
            return nn.Sequential(ConvBlock(c, c, 3, 1, 1), ConvBlock(c, c, 3, 1, 1),
                                 nn.Conv2d(c, self.output_channels, 1, bias=True))

        self.head_p3, self.head_p4, self.head_p5 = head(c3), head(c4), head(c5)
        self._c = (c1, c3, c4, c5)
        self.initialize_detection_biases()

    @property
    def anchors(self):
        return [self.anchors_p3, self.anchors_p4, self.anchors_p5]

    def initialize_detection_biases(self, prior=0.01):
        """Objectness bias = -log((1-prior)/prior), class biases 0 (train.py:519-566)."""
        obj = -math.log((1 - prior) / prior)
        for hd in (self.head_p3, self.head_p4, self.head_p5):
            last = hd[-1]
            if last.bias is None:
                last.bias = nn.Parameter(torch.zeros(last.out_channels, device=last.weight.device))
            with torch.no_grad():
                b = last.bias.view(self.num_anchors, 5 + self.num_classes)
                b[:, 4].fill_(obj)
                if self.num_classes > 0:
                    b[:, 5:].fill_(0.0)

    # ---- graph ---------------------------------------------------------------------------------
    def _trace(self, g: Plan):
        if g.Himg % 32 or g.Wimg % 32:
            raise AssertionError(f"input {g.Himg}x{g.Wimg} must be a multiple of 32 (train.py:606,616,626)")
        c1, c3, c4, c5 = self._c
        B = g.B
        H3, W3, H4, W4, H5, W5 = g.Himg // 8, g.Wimg // 8, g.Himg // 16, g.Wimg // 16, g.Himg // 32, g.Wimg // 32
        x = g.input.view()
        x = g.conv(x, self.stem[0], self.stem[1])
        x = g.conv(x, self.stem[3], self.stem[4])
        bp3, bp4, bp5 = self.backbone_p3, self.backbone_p4, self.backbone_p5
        sb = True
        x = bp3[0]._emit(g, x, side_branch=sb)
        x = g.conv(x, bp3[1], bp3[2])
        p3 = bp3[4]._emit(g, x, side_branch=sb)
        x = g.conv(p3, bp4[0], bp4[1])
        p4 = bp4[3]._emit(g, x, side_branch=sb)
        x = g.conv(p4, bp5[0], bp5[1])
        x = bp5[3]._emit(g, x, side_branch=sb)
        # concat buffers: [p4_down | p5_backbone], [p5_up | p4_lateral], [p4_up | p3_lateral], [p3_down | p4_fpn]
        cat_pan5 = g.new_buffer(B, H5, W5, c4 + c5, "cat_pan5")
        cat_fpn4 = g.new_buffer(B, H4, W4, 2 * c4, "cat_fpn4")
        cat_fpn3 = g.new_buffer(B, H3, W3, 2 * c3, "cat_fpn3")
        cat_pan4 = g.new_buffer(B, H4, W4, c3 + c4, "cat_pan4")
        p5 = self.sppf._emit(g, x, out=cat_pan5.view(c4, c5))
        self.lateral_p4._emit(g, p4, out=cat_fpn4.view(c4, c4))
        self.lateral_p3._emit(g, p3, out=cat_fpn3.view(c3, c3))
        self.reduce_p5_for_p4._emit(g, p5, out=cat_fpn4.view(0, c4), upsample=True)
        p4f = self.merge_p4._emit(g, cat_fpn4.view(), out=cat_pan4.view(c3, c4), side_branch=sb)
        self.reduce_p4_for_p3._emit(g, p4f, out=cat_fpn3.view(0, c3), upsample=True)
        p3f = self.merge_p3._emit(g, cat_fpn3.view(), side_branch=sb)

        def head(feat, hd):
            h = hd[1]._emit(g, hd[0]._emit(g, feat))
            return g.conv(h, hd[2], None)

        # The P3 and P4 heads (one third of the forward FLOPs) depend only on p3_fpn / p4_panet: they run on the
        # side lane next to the PANet chain, so their HBM-bound BatchNorm passes and tail rounds overlap the
        # neck's MFMA work and vice versa.  Outputs are registered in the reference's order (P3, P4, P5).
        with g.side_lane():
            out3 = head(p3f, self.head_p3)
        self.downsample_p3_to_p4._emit(g, p3f, out=cat_pan4.view(0, c3))
        p4n = self.panet_merge_p4._emit(g, cat_pan4.view())
        with g.side_lane():
            out4 = head(p4n, self.head_p4)
        self.downsample_p4_to_p5._emit(g, p4n, out=cat_pan5.view(0, c4))
        p5n = self.panet_merge_p5._emit(g, cat_pan5.view())
        out5 = head(p5n, self.head_p5)
        g.join()
        for o in (out3, out4, out5):
            g.mark_output(o, "nhwc_heads")

    def _package(self, outs):
        return outs
