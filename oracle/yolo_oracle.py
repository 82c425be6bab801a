"""CPU oracle for the YOLO hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU with stock torch / numpy ops, the algorithm of the reference's
training / inference hot path so the HIP kernels can be checked against it on a box where the
reference itself is absent.  It is *not* part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product
package (``yolo-from-scratch_amd/``) never does.

Pinning: every function here is checked against outputs of the reference itself (generated in the
build container by ``tests/golden/make_golden.py`` and committed as ``tests/golden/*.npz``) in
``tests/test_oracle_pinned.py``.  One piece cannot be pinned that way: the class-aware NMS
(`nms_batched`), because the reference delegates it to ``torchvision.ops.batched_nms``
(train.py:1232-1233), a third-party dependency that is not installed here and that the reference
pins no version of (README.md:25).  Its published algorithm is restated (BOTH branches and the size
rule that picks one, see `nms_batched`); no output of torchvision itself is available in this image,
so for that step "parity unpinned" applies; it is anchored on the reference's own python ``nms``
(train.py:1086-1112) known-answer tests (single class, no threshold ties => identical selections).

The model is written functionally: it walks the network graph reading tensors out of a flat
``{state_dict key: tensor}`` mapping, so the same code evaluates reference weights, product weights
or freshly seeded weights.  Citations are ``train.py:line`` of the reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.BatchNorm2d default (train.py:261 uses defaults)
BN_MOMENTUM = 0.1
DEFAULT_ANCHORS = (    # train.py:372-374 (model) == train.py:81-83 (dataset)
    ((10, 13), (16, 30), (33, 23)),
    ((30, 61), (62, 45), (59, 119)),
    ((116, 90), (156, 198), (373, 326)),
)
OBJ_SCALE_WEIGHTS = (4.0, 1.0, 0.4)   # train.py:865
W_BOX, W_CLS = 0.05, 0.5              # train.py:879


# --------------------------------------------------------------------------------------------
# model graph
# --------------------------------------------------------------------------------------------
class _RoundBf16(torch.autograd.Function):
    """Storage emulation of the bf16 path (no counterpart in the reference, which is fp32 only): the value is rounded
    to bf16 on the way forward, its gradient on the way back; all arithmetic stays fp32."""

    @staticmethod
    def forward(ctx, t, fwd, bwd):
        ctx.bwd = bwd
        return t.to(torch.bfloat16).to(torch.float32) if fwd else t.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(torch.float32) if ctx.bwd else g), None, None


class _Net:
    """Evaluates blocks against a parameter dict.  `training` selects batch statistics
    (and in-place running-stat updates, as nn.BatchNorm2d does) or running statistics.

    storage="bf16" models WHERE the product's bf16 path (BASELINE configs 3-4) rounds: the image, the per-step weight
    packs, every stored conv output / activation and every stored activation gradient are rounded to bf16, while
    convolution sums, BatchNorm statistics, SiLU, the loss and the parameter gradients stay fp32.  It is an error MODEL
    used to derive the bf16 tests' tolerances (how far a correct bf16-storage pipeline drifts from fp32); the fp32 mode
    is the pinned restatement of the reference."""

    SITES = ("img", "w", "y", "dy", "a", "da", "dhead")     # the places where the bf16 path stores a rounded value

    def __init__(self, P: Dict[str, torch.Tensor], training: bool, storage="f32"):
        """storage: "f32", "bf16" (every site) or a collection of SITES names (ablations: which rounding costs what)."""
        self.P, self.training = P, training
        self.sites = frozenset(self.SITES) if storage == "bf16" else frozenset() if storage == "f32" else frozenset(storage)
        assert self.sites <= frozenset(self.SITES), storage
        self.bf16 = bool(self.sites)

    def q(self, t, fwd=None, bwd=None):
        """fwd / bwd: site names (or None) of the forward value / of its gradient."""
        f, b = fwd in self.sites, bwd in self.sites
        return _RoundBf16.apply(t, f, b) if (f or b) else t

    def _w(self, key):                       # the bf16 weight pack: rounded copy, fp32 master gradient
        return self.q(self.P[key], "w", None)

    def _bn_silu(self, y, bn, residual=None):
        P = self.P
        y = self.q(y, "y", "dy")             # stored pre-BN output (statistics are those of the stored values); dY stored
        out = F.batch_norm(y, P[f"{bn}.running_mean"], P[f"{bn}.running_var"],
                           P[f"{bn}.weight"], P[f"{bn}.bias"], self.training, BN_MOMENTUM, BN_EPS)
        if self.training and f"{bn}.num_batches_tracked" in P:
            P[f"{bn}.num_batches_tracked"] += 1
        a = F.silu(out)
        if residual is not None:
            a = residual + a
        return self.q(a, "a", "da")          # stored activation; its gradient (sum over consumers) stored

    def cbs(self, x, name, stride=1, residual=None):
        """ConvBlock = bias-free conv -> BN -> SiLU (train.py:253-265); padding = k//2."""
        w = self._w(f"{name}.conv.weight")
        return self._bn_silu(F.conv2d(x, w, None, stride, w.shape[-1] // 2), f"{name}.bn", residual)

    def inline(self, x, conv, bn, stride):
        """Conv2d(bias=True) -> BN -> SiLU written inline in the reference (train.py:401-404,
        408, 413, 418 and SPPF 236-241)."""
        w = self._w(f"{conv}.weight")
        return self._bn_silu(F.conv2d(x, w, self.P[f"{conv}.bias"], stride, w.shape[-1] // 2), bn)

    def bottleneck(self, x, name):
        """x + CB3x3(CB3x3(x)) (train.py:295-306); shortcut always active in this net."""
        return self.cbs(self.cbs(x, f"{name}.conv1"), f"{name}.conv2", residual=x)

    def c3(self, x, name):
        """conv3(cat[bottlenecks(conv1 x), conv2 x]) (train.py:288-293)."""
        a = self.cbs(x, f"{name}.conv1")
        i = 0
        while f"{name}.bottlenecks.{i}.conv1.conv.weight" in self.P:
            a = self.bottleneck(a, f"{name}.bottlenecks.{i}")
            i += 1
        return self.cbs(torch.cat([a, self.cbs(x, f"{name}.conv2")], 1), f"{name}.conv3")

    def sppf(self, x, name):
        """1x1 -> three chained 5x5/s1/p2 max-pools -> cat -> 1x1 (train.py:243-251)."""
        x = self.inline(x, f"{name}.conv1", f"{name}.bn1", 1)
        y1 = F.max_pool2d(x, 5, 1, 2)
        y2 = F.max_pool2d(y1, 5, 1, 2)
        y3 = F.max_pool2d(y2, 5, 1, 2)
        return self.inline(torch.cat([x, y1, y2, y3], 1), f"{name}.conv2", f"{name}.bn2", 1)

    def head(self, x, name):
        """CB3x3, CB3x3, Conv1x1(+bias) (train.py:452-466)."""
        x = self.cbs(self.cbs(x, f"{name}.0"), f"{name}.1")
        o = F.conv2d(x, self._w(f"{name}.2.weight"), self.P[f"{name}.2.bias"])
        return self.q(o, None, "dhead")      # fp32 head output, bf16 head gradient


def forward(P: Dict[str, torch.Tensor], x: torch.Tensor, num_classes: int,
            training: bool = True, storage: str = "f32") -> List[torch.Tensor]:
    """YOLO.forward (train.py:568-632): NCHW image batch -> three (B,G,G,3,5+nc) tensors."""
    n = _Net(P, training, storage)
    x = n.q(x, "img", None)
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    # backbone (train.py:572-576)
    s = n.inline(n.inline(x, "stem.0", "stem.1", 2), "stem.3", "stem.4", 2)
    p3 = n.c3(n.inline(n.c3(s, "backbone_p3.0"), "backbone_p3.1", "backbone_p3.2", 2), "backbone_p3.4")
    p4 = n.c3(n.inline(p3, "backbone_p4.0", "backbone_p4.1", 2), "backbone_p4.3")
    p5 = n.c3(n.inline(p4, "backbone_p5.0", "backbone_p5.1", 2), "backbone_p5.3")
    p5 = n.sppf(p5, "sppf")
    # FPN top-down (train.py:580-589)
    lat4, lat3 = n.cbs(p4, "lateral_p4"), n.cbs(p3, "lateral_p3")
    p4f = n.c3(torch.cat([up(n.cbs(p5, "reduce_p5_for_p4")), lat4], 1), "merge_p4")
    p3f = n.c3(torch.cat([up(n.cbs(p4f, "reduce_p4_for_p3")), lat3], 1), "merge_p3")
    # PANet bottom-up (train.py:593-598)
    p4n = n.c3(torch.cat([n.cbs(p3f, "downsample_p3_to_p4", 2), p4f], 1), "panet_merge_p4")
    p5n = n.c3(torch.cat([n.cbs(p4n, "downsample_p4_to_p5", 2), p5], 1), "panet_merge_p5")
    # heads; note P3 reads the FPN output, P4/P5 the PANet outputs (train.py:602, 612, 622)
    outs = []
    for feat, name in ((p3f, "head_p3"), (p4n, "head_p4"), (p5n, "head_p5")):
        o = n.head(feat, name)
        b, _, h, w = o.shape
        outs.append(o.view(b, 3, 5 + num_classes, h, w).permute(0, 3, 4, 1, 2).contiguous())
    return outs


def anchors_of(P) -> List[torch.Tensor]:
    return [P["anchors_p3"], P["anchors_p4"], P["anchors_p5"]]


# --------------------------------------------------------------------------------------------
# decode + loss (Appendix A of SURVEY.md; train.py:634-886)
# --------------------------------------------------------------------------------------------
def decode(raw: torch.Tensor, anchors: torch.Tensor, img_size: int = 640) -> torch.Tensor:
    """decode_predictions (train.py:736-779).  Channels 4: pass through untouched."""
    _, gh, gw, na, _ = raw.shape
    jj = torch.arange(gw, dtype=raw.dtype).view(1, 1, gw, 1)
    ii = torch.arange(gh, dtype=raw.dtype).view(1, gh, 1, 1)
    sg = torch.sigmoid(raw[..., :4])
    bx = ((sg[..., 0] * 2.0 - 0.5) + jj) / gw
    by = ((sg[..., 1] * 2.0 - 0.5) + ii) / gh
    aw = (anchors[:, 0] / img_size).view(1, 1, 1, na)
    ah = (anchors[:, 1] / img_size).view(1, 1, 1, na)
    bw = aw * torch.pow(2.0 * sg[..., 2], 2)
    bh = ah * torch.pow(2.0 * sg[..., 3], 2)
    return torch.cat([torch.stack([bx, by, bw, bh], -1), raw[..., 4:]], -1)


def ciou(pred: torch.Tensor, tgt: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """ciou_loss (train.py:646-710): mean over N of 1 - (IoU - rho^2/c^2 - alpha*v); alpha detached."""
    px, py, pw, ph = pred.unbind(-1)
    tx, ty, tw, th = tgt.unbind(-1)
    px1, px2, py1, py2 = px - pw / 2, px + pw / 2, py - ph / 2, py + ph / 2
    tx1, tx2, ty1, ty2 = tx - tw / 2, tx + tw / 2, ty - th / 2, ty + th / 2
    iw = torch.clamp(torch.min(px2, tx2) - torch.max(px1, tx1), min=0)
    ih = torch.clamp(torch.min(py2, ty2) - torch.max(py1, ty1), min=0)
    inter = iw * ih
    iou = inter / (pw * ph + tw * th - inter + eps)
    rho2 = (px - tx) ** 2 + (py - ty) ** 2
    cw = torch.max(px2, tx2) - torch.min(px1, tx1)
    ch = torch.max(py2, ty2) - torch.min(py1, ty1)
    c2 = cw ** 2 + ch ** 2 + eps
    v = (4 / (math.pi ** 2)) * torch.pow(torch.atan(pw / (ph + eps)) - torch.atan(tw / (th + eps)), 2)
    with torch.no_grad():
        alpha = v / (1 - iou + v + eps)
    return (1 - (iou - rho2 / c2 - alpha * v)).mean()


def loss_one_scale(pred, target, anchors, num_classes: int = 1):
    """yolo_loss (train.py:796-838).  Loss decode always uses img_size=640 (quirk Q1)."""
    dec = decode(pred, anchors)
    pos = target[..., 4] > 0.5
    n_pos = int(pos.sum())
    zero = torch.tensor(0.0)
    box = ciou(dec[..., :4][pos], target[..., :4][pos]) if n_pos else zero
    obj = F.binary_cross_entropy_with_logits(pred[..., 4:5], target[..., 4:5])
    if n_pos and num_classes > 0:
        cls = F.binary_cross_entropy_with_logits(pred[..., 5:][pos].reshape(-1),
                                                 target[..., 5:][pos].reshape(-1))
    else:
        cls = zero
    return W_BOX * box + 1.0 * obj + W_CLS * cls, box, obj, cls


def loss_multiscale(preds, targets, anchors_list, num_classes: int = 1):
    """yolo_loss_multiscale (train.py:865-886)."""
    tot = tb = to = tc = 0.0
    for p, t, a, w in zip(preds, targets, anchors_list, OBJ_SCALE_WEIGHTS):
        _, b, o, c = loss_one_scale(p, t, a, num_classes)
        tot = tot + (W_BOX * b + w * o + W_CLS * c)
        tb, to, tc = tb + b, to + o, tc + c
    return tot, tb, to, tc


# --------------------------------------------------------------------------------------------
# clip + Adam (train.py:916-918)
# --------------------------------------------------------------------------------------------
def clip_coef(grads: Sequence[torch.Tensor], max_norm: float = 10.0) -> Tuple[float, float]:
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    return float(total), float(torch.clamp(max_norm / (total + 1e-6), max=1.0))


def adam_step(p, g, m, v, step: int, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    """One torch.optim.Adam update (defaults, no weight decay / amsgrad), in place; step is 1-based."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = v.sqrt() / math.sqrt(1 - b2 ** step) + eps
    p.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))


# --------------------------------------------------------------------------------------------
# inference post-process (train.py:1152-1233) and NMS
# --------------------------------------------------------------------------------------------
def candidates(preds, anchors_list, img_size: int, num_classes: int, conf_threshold: float,
               pad_left: float = 0.0, pad_top: float = 0.0, scale: float = 1.0):
    """Candidate extraction of `predict` (train.py:1152-1229) for a batch of ONE image.
    Order: scale-major P3,P4,P5 then row-major (i, j, a).  Returns fp32 boxes (M,4) corners in
    original-image pixels, fp32 scores (M,), int64 classes (M,)."""
    boxes, scores, classes = [], [], []
    for pred, anc in zip(preds, anchors_list):
        d = decode(pred, anc, img_size)[0]
        obj = torch.sigmoid(pred[0, ..., 4])
        keep = obj > conf_threshold                              # strict >, objectness only
        if not bool(keep.any()):
            continue
        det = d[keep]
        o = obj[keep]
        if num_classes == 1:
            cp = torch.sigmoid(det[:, 5])
            cid = torch.zeros(len(det), dtype=torch.long)
        else:
            cp, cid = torch.sigmoid(det[:, 5:]).max(dim=1)       # first max wins
        xc, yc, w, h = (det[:, k] * img_size for k in range(4))
        x1, y1, x2, y2 = xc - w / 2, yc - h / 2, xc + w / 2, yc + h / 2
        x1, y1, x2, y2 = ((x1 - pad_left) / scale, (y1 - pad_top) / scale,
                          (x2 - pad_left) / scale, (y2 - pad_top) / scale)
        boxes.append(torch.stack([x1, y1, x2, y2], 1))
        scores.append(o * cp)
        classes.append(cid)
    if not boxes:
        return torch.zeros(0, 4), torch.zeros(0), torch.zeros(0, dtype=torch.long)
    return torch.cat(boxes), torch.cat(scores), torch.cat(classes)


# torchvision.ops.batched_nms, the call of train.py:1232-1233.  torchvision is a third-party dependency that is absent
# from this image and that the reference pins no version of (README.md:25: "pip install torch torchvision"); what follows
# restates its PUBLISHED algorithm (torchvision/ops/boxes.py: batched_nms, _batched_nms_coordinate_trick,
# _batched_nms_vanilla; torchvision/csrc/ops/cpu/nms_kernel.cpp: nms_kernel_impl), unchanged since torchvision 0.9:
#
#   batched_nms(boxes, scores, idxs, thr):
#       if boxes.numel() > (4000 if boxes.device.type == "cpu" else 20000): _batched_nms_vanilla
#       else:                                                               _batched_nms_coordinate_trick
#   _batched_nms_coordinate_trick: max_coordinate = boxes.max(); offsets = idxs.to(boxes) * (max_coordinate + 1);
#       nms(boxes + offsets[:, None], scores, thr)            -- ONE class-agnostic nms over shifted boxes
#   _batched_nms_vanilla: nms() per unique class; the union of the kept indices, sorted by score descending
#   nms (CPU kernel): areas = (x2-x1)*(y2-y1); order = scores.sort(stable, descending); greedy scan, j is suppressed by
#       a kept i when  inter / (area_i + area_j - inter) > thr  with inter = max(0, xx2-xx1) * max(0, yy2-yy1);
#       boxes are scalar_t = float, every operation rounds to fp32, and `thr` is a C double: the fp32 IoU is PROMOTED
#       to double for the comparison (so with thr = 0.4 an IoU of exactly float32(0.4) = 0.4000000059... IS suppressed).
#
# The reference's CPU path (the parity target) takes the coordinate trick for M <= 1000 candidates and the per-class
# form above that.  The two differ for nc > 1: the shifted coordinates round on a coarser grid (ulp 0.004 at 5e4), and
# boxes with negative coordinates of one class can reach into the previous class's band and suppress across classes.
NMS_MODES = ("vanilla", "trick", "cpu", "cuda")   # "cpu"/"cuda" = torchvision's size rule for a tensor on that device


def nms_plain(boxes: np.ndarray, scores: np.ndarray, thr: float, float_threshold: bool = False) -> np.ndarray:
    """torchvision.ops.nms, fp32 boxes: kept indices in descending-score order (ties: lower index first).  The CPU kernel takes
    the threshold as a C double and promotes the fp32 IoU for the comparison; the CUDA kernel (float_threshold) compares
    float against float -- an IoU exactly equal to float(thr) is suppressed by the first and kept by the second when
    float(thr) > thr."""
    boxes = np.asarray(boxes, np.float32).reshape(-1, 4)
    scores = np.asarray(scores, np.float32).reshape(-1)
    thr = float(np.float32(thr)) if float_threshold else float(thr)
    order = np.argsort(-scores, kind="stable")
    x1, y1, x2, y2 = (boxes[:, k] for k in range(4))
    with np.errstate(invalid="ignore", over="ignore"):
        area = ((x2 - x1).astype(np.float32) * (y2 - y1).astype(np.float32)).astype(np.float32)
    dead = np.zeros(len(order), bool)
    kept = []
    for a, i in enumerate(order):
        if dead[a]:
            continue
        kept.append(int(i))
        rest = order[a + 1:]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest])).astype(np.float32)
            h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest])).astype(np.float32)
            inter = (w * h).astype(np.float32)
            iou = (inter / ((area[i] + area[rest]).astype(np.float32) - inter).astype(np.float32)).astype(np.float32)
        dead[a + 1:] |= iou.astype(np.float64) > thr
    return np.asarray(kept, np.int64)


def nms_shifted_boxes(boxes: np.ndarray, classes: np.ndarray) -> np.ndarray:
    """boxes_for_nms of _batched_nms_coordinate_trick: three fp32 operations, one rounding each."""
    boxes = np.asarray(boxes, np.float32).reshape(-1, 4)
    with np.errstate(invalid="ignore", over="ignore"):
        unit = np.float32(boxes.max() + np.float32(1))                                  # max_coordinate + tensor(1).to(boxes)
        offsets = (np.asarray(classes).reshape(-1).astype(np.float32) * unit).astype(np.float32)   # idxs.to(boxes) * (...)
        return (boxes + offsets[:, None]).astype(np.float32)


def nms_uses_trick(num_boxes: int, mode: str) -> bool:
    """torchvision's branch for M boxes: numel = 4 M against 4000 (CPU tensors) / 20000 (GPU tensors)."""
    assert mode in NMS_MODES, mode
    if mode in ("vanilla", "trick"):
        return mode == "trick"
    return not (4 * num_boxes > (4000 if mode == "cpu" else 20000))


def nms_batched(boxes: np.ndarray, scores: np.ndarray, classes: np.ndarray, thr: float, mode: str = "cpu") -> np.ndarray:
    """torchvision.ops.batched_nms as the reference calls it (train.py:1232-1233).  mode "cpu" (default:
    the reference's CPU path) / "cuda" apply torchvision's size rule, "vanilla" / "trick" force one branch.
    Returns kept candidate indices in descending-score order (int64)."""
    boxes = np.asarray(boxes, np.float32).reshape(-1, 4)
    scores = np.asarray(scores, np.float32).reshape(-1)
    classes = np.asarray(classes).reshape(-1).astype(np.int64)
    if len(scores) == 0:
        return np.zeros(0, np.int64)
    f32thr = mode == "cuda"                                          # the device kernel's float-vs-float comparison
    if nms_uses_trick(len(scores), mode):
        return nms_plain(nms_shifted_boxes(boxes, classes), scores, thr, f32thr)
    keep_mask = np.zeros(len(scores), bool)
    for c in np.unique(classes):
        cur = np.nonzero(classes == c)[0]
        keep_mask[cur[nms_plain(boxes[cur], scores[cur], thr, f32thr)]] = True
    keep = np.nonzero(keep_mask)[0]
    return keep[np.argsort(-scores[keep], kind="stable")].astype(np.int64)          # CPU torch.sort is a stable sort


def iou_corners(a, b) -> float:
    """compute_iou_corners (train.py:1064-1084), python floats."""
    iw = max(0, min(a[2], b[2]) - max(a[0], b[0]))
    ih = max(0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iw * ih
    union = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
    return inter / union if union > 0 else 0


def nms_python(dets: list, thr: float) -> list:
    """nms (train.py:1086-1112): class-agnostic, keeps a later box only if IoU < thr with every
    kept one (so IoU == thr suppresses), python-float arithmetic, stable descending sort."""
    todo = sorted(dets, key=lambda d: d[4], reverse=True)
    out = []
    while todo:
        top, todo = todo[0], todo[1:]
        out.append(top)
        todo = [d for d in todo if iou_corners(top, d) < thr]
    return out


# --------------------------------------------------------------------------------------------
# synthetic targets (assignment rule of YOLODataset.__getitem__, train.py:164-205)
# --------------------------------------------------------------------------------------------
def assign_targets(boxes: Sequence[Sequence[Tuple[int, float, float, float, float]]], img_size: int,
                   num_classes: int, anchors=DEFAULT_ANCHORS) -> List[torch.Tensor]:
    """boxes[b] = [(class, xc, yc, w, h) normalised to the padded square image].  Returns the three
    dense (B,G,G,3,5+nc) target tensors: best shape-IoU anchor over all 9 (first max wins across
    scales because the comparison is strict `>`), cell = min(int(c*G), G-1), first writer wins."""
    grids = [img_size // 8, img_size // 16, img_size // 32]
    out = [torch.zeros(len(boxes), g, g, 3, 5 + num_classes) for g in grids]
    anc = [torch.tensor(a, dtype=torch.float32) for a in anchors]
    for b, blist in enumerate(boxes):
        for (cid, xc, yc, w, h) in blist:
            wh = torch.tensor([w * img_size, h * img_size])
            best, bs, ba = -1.0, 0, 0
            for s in range(3):
                iw = torch.min(wh[0], anc[s][:, 0])
                ih = torch.min(wh[1], anc[s][:, 1])
                inter = iw * ih
                iou = inter / (wh[0] * wh[1] + anc[s][:, 0] * anc[s][:, 1] - inter + 1e-16)
                if iou.max().item() > best:
                    best, bs, ba = iou.max().item(), s, int(iou.argmax())
            g = grids[bs]
            gx, gy = min(int(xc * g), g - 1), min(int(yc * g), g - 1)
            t = out[bs][b]
            if t[gy, gx, ba, 4] == 0:
                t[gy, gx, ba, 0:4] = torch.tensor([xc, yc, w, h])
                t[gy, gx, ba, 4] = 1.0
                t[gy, gx, ba, 5 if num_classes == 1 else 5 + cid] = 1.0
    return out


def synthetic_boxes(batch: int, num_classes: int, img_size: int, n_obj: int = 8, seed: int = 2000):
    """Seeded synthetic labels of SURVEY section 8(d): centres U(0.05,0.95)^2, log-uniform w,h in
    [8,320] px, uniform class."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(batch):
        c = rng.uniform(0.05, 0.95, size=(n_obj, 2))
        wh = np.exp(rng.uniform(math.log(8), math.log(320), size=(n_obj, 2))) / img_size
        k = rng.integers(0, max(num_classes, 1), size=n_obj)
        out.append([(int(k[i]), float(c[i, 0]), float(c[i, 1]), float(wh[i, 0]), float(wh[i, 1]))
                    for i in range(n_obj)])
    return out
