"""Loss-side mirror of the reference's public functions (train.py:634-886) on the HIP path.

Same names, argument meaning and return conventions as the reference: 0-dim fp32 tensors that support
`.backward()` and `.item()`.  Each function is one autograd node around a libyolohip call; the fused
three-scale kernel evaluates decode + CIoU + both BCE terms in a single pass per step.
"""
from __future__ import annotations

from typing import List, Sequence

import torch

from . import _lib as L

_OBJ_W = (4.0, 1.0, 0.4)          # train.py:865
_W_BOX, _W_CLS = 0.05, 0.5        # train.py:879
LOSS_IMG_SIZE = 640.0             # yolo_loss decodes with the default img_size (train.py:796, quirk Q1)

def _host_anchors(a) -> List[float]:
    """(3,2) anchor tensor -> 6 host floats.  The kernels take anchors by value (HOST array), so a device
    tensor costs one read-back; the copy is cached ON THE TENSOR OBJECT (model buffers live as long as the
    model) together with its version counter -- never keyed by address, which the allocator recycles."""
    if isinstance(a, torch.Tensor):
        hit = getattr(a, "_yh_host", None)
        if hit is None or hit[0] != a._version:
            hit = (a._version, [float(v) for v in a.detach().reshape(-1).tolist()])
            try:
                a._yh_host = hit
            except AttributeError:
                pass
        return hit[1]
    return [float(v) for row in a for v in row]


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_gpu(t: torch.Tensor, who: str):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: the HIP path needs GPU tensors (got {t.device}); no CPU fallback in this package")
    L.lib()


# ------------------------------------------------------------------------------------------------
class _Decode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, anc6, img_size):
        raw = raw.contiguous()
        B, GH, GW, A, CH = raw.shape
        out = torch.empty_like(raw)
        L.check(L.lib().yh_decode(raw.data_ptr(), out.data_ptr(), L.floats(anc6), B, GH, GW, CH - 5, float(img_size),
                                  _stream(raw)), "decode")
        ctx.save_for_backward(raw)
        ctx.anc6, ctx.img = anc6, float(img_size)
        return out

    @staticmethod
    def backward(ctx, gout):
        (raw,) = ctx.saved_tensors
        B, GH, GW, A, CH = raw.shape
        gout = gout.contiguous()
        graw = torch.empty_like(raw)
        L.check(L.lib().yh_decode_bwd(raw.data_ptr(), gout.data_ptr(), graw.data_ptr(), L.floats(ctx.anc6), B, GH, GW,
                                      CH - 5, ctx.img, _stream(raw)), "decode_bwd")
        return graw, None, None


def decode_predictions(raw_preds, anchors, img_size=640):
    """(B,GH,GW,3,5+nc) raw head output -> same shape with channels 0..3 = (bx,by,bw,bh) normalised;
    channels 4: untouched (train.py:712-779)."""
    _need_gpu(raw_preds, "decode_predictions")
    if raw_preds.dim() != 5 or raw_preds.shape[3] != 3:
        raise ValueError("expected (B, GH, GW, 3, 5+nc)")
    return _Decode.apply(raw_preds.float(), _host_anchors(anchors), img_size)


# ------------------------------------------------------------------------------------------------
class _CIoU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, tgt, eps):
        pred, tgt = pred.contiguous().float(), tgt.contiguous().float()
        N = pred.shape[0]
        dpred = torch.empty_like(pred)
        loss = torch.empty((), device=pred.device, dtype=torch.float32)
        ws = torch.empty(2 * ((N + 255) // 256) + 2, device=pred.device, dtype=torch.float64)
        L.check(L.lib().yh_ciou(pred.data_ptr(), tgt.data_ptr(), dpred.data_ptr(), N, float(eps), 1.0, loss.data_ptr(),
                                ws.data_ptr(), _stream(pred)), "ciou")
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None


def ciou_loss(pred_boxes, target_boxes, eps=1e-7):
    """mean over N of 1 - CIoU, boxes (N,4) = (x,y,w,h) (train.py:634-710)."""
    _need_gpu(pred_boxes, "ciou_loss")
    if pred_boxes.shape[0] == 0:
        return pred_boxes.sum() * float("nan")      # the reference's mean() over an empty set
    return _CIoU.apply(pred_boxes, target_boxes.to(pred_boxes.device), eps)


# ------------------------------------------------------------------------------------------------
def run_loss_kernel(preds: Sequence[torch.Tensor], targets: Sequence[torch.Tensor], dpreds, anchors18, grids, B, nc,
                    loss_w, grad_w, out: torch.Tensor, ws: torch.Tensor, stream: int, dpred_bf16: bool = False, dpred_ld=None):
    """Thin typed call of yh_yolo_loss(_ex); preds/targets/dpreds are 3-slot lists (None = absent).  dpred_bf16 /
    dpred_ld: the bf16 path's head gradient (bf16, pixel rows padded to a multiple of 8 channels)."""
    L.check(L.lib().yh_yolo_loss_ex(L.ptr3(preds), L.ptr3(targets), L.ptr3(dpreds) if dpreds is not None else None,
                                    int(bool(dpred_bf16)), L.int3(dpred_ld) if dpred_ld is not None else None,
                                    L.floats(anchors18), L.int3(grids), B, nc, LOSS_IMG_SIZE,
                                    L.floats(loss_w) if loss_w is not None else None,
                                    L.floats(grad_w) if grad_w is not None else None,
                                    out.data_ptr(), ws.data_ptr(), stream), "yolo_loss")


class _YoloLoss(torch.autograd.Function):
    """Outputs (total, sum box, sum obj, sum cls); all four are differentiable."""

    @staticmethod
    def forward(ctx, nc, anchors18, loss_w, n_scales, *tensors):
        preds = [t.contiguous().float() for t in tensors[:n_scales]]
        targets = [t.contiguous().float() for t in tensors[n_scales:]]
        dev = preds[0].device
        B = preds[0].shape[0]
        grids = [p.shape[1] for p in preds] + [0] * (3 - n_scales)
        for p, t in zip(preds, targets):
            if p.shape != t.shape or p.shape[1] != p.shape[2] or p.shape[3] != 3 or p.shape[4] != 5 + nc:
                raise ValueError(f"prediction {tuple(p.shape)} / target {tuple(t.shape)} are not (B,G,G,3,{5 + nc})")
        pad = [None] * (3 - n_scales)
        out = torch.empty(13, device=dev, dtype=torch.float32)
        ws = torch.empty(int(L.lib().yh_loss_ws(L.int3(grids), B)) + 8, device=dev, dtype=torch.float32)
        run_loss_kernel(preds + pad, targets + pad, None, anchors18, grids, B, nc, loss_w, None, out, ws, _stream(preds[0]))
        ctx.save_for_backward(*preds, *targets)
        ctx.meta = (nc, anchors18, loss_w, n_scales, grids, B)
        return tuple(out[k].clone() for k in range(4))

    @staticmethod
    def backward(ctx, g_total, g_box, g_obj, g_cls):
        nc, anchors18, loss_w, n_scales, grids, B = ctx.meta
        saved = ctx.saved_tensors
        preds, targets = list(saved[:n_scales]), list(saved[n_scales:])
        dev = preds[0].device
        # upstream gradients are four scalars; fold them into per-scale gradient weights (one host read)
        gt, gb, go, gc = torch.stack([g.reshape(()).float() for g in (g_total, g_box, g_obj, g_cls)]).tolist()
        grad_w = []
        for s in range(3):
            grad_w += [gt * loss_w[3 * s] + gb, gt * loss_w[3 * s + 1] + go, gt * loss_w[3 * s + 2] + gc]
        dpreds = [torch.empty_like(p) for p in preds]
        pad = [None] * (3 - n_scales)
        out = torch.empty(13, device=dev, dtype=torch.float32)
        ws = torch.empty(int(L.lib().yh_loss_ws(L.int3(grids), B)) + 8, device=dev, dtype=torch.float32)
        run_loss_kernel(preds + pad, targets + pad, dpreds + pad, anchors18, grids, B, nc, loss_w, grad_w, out, ws,
                        _stream(preds[0]))
        return (None, None, None, None, *dpreds, *([None] * n_scales))


def _anchors18(anchors_list) -> List[float]:
    flat: List[float] = []
    for a in anchors_list:
        flat += _host_anchors(a)
    return flat + [1.0] * (18 - len(flat))


def yolo_loss(predictions, targets, anchors, num_classes=1):
    """Single-scale loss -> (0.05*box + 1.0*obj + 0.5*cls, box, obj, cls) (train.py:781-838)."""
    _need_gpu(predictions, "yolo_loss")
    w = [_W_BOX, 1.0, _W_CLS] + [0.0] * 6
    return _YoloLoss.apply(num_classes, _anchors18([anchors]), w, 1, predictions, targets.to(predictions.device))


def yolo_loss_multiscale(predictions, targets, anchors_list, num_classes=1):
    """Sum over P3,P4,P5 of 0.05*box + w_s*obj + 0.5*cls, w = [4.0,1.0,0.4]; also returns the
    unweighted component sums (train.py:840-886)."""
    _need_gpu(predictions[0], "yolo_loss_multiscale")
    if len(predictions) != 3 or len(targets) != 3 or len(anchors_list) != 3:
        raise ValueError("expected three scales")
    w: List[float] = []
    for s in range(3):
        w += [_W_BOX, _OBJ_W[s], _W_CLS]
    dev = predictions[0].device
    return _YoloLoss.apply(num_classes, _anchors18(anchors_list), w, 3, *predictions, *[t.to(dev) for t in targets])
