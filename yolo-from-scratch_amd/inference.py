"""Inference post-process on the HIP path: candidate extraction + class-aware global NMS
(train.py:1114-1250), plus `predict` with the reference's signature.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from .functional import _anchors18
from .graph import state_token
from .hostside import letterbox_resize


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


# torchvision.ops.batched_nms has two branches (include/yolohip.h, yh_nms).  "cpu" = torchvision's size rule for CPU
# tensors, i.e. what the reference's CPU path -- the parity target -- executes: coordinate trick for M <= 1000 candidates,
# per class above; "cuda" = the rule torchvision applies to GPU tensors (limit 20000 elements); the other two force a branch.
NMS_MODES = {"per_class": 0, "vanilla": 0, "coordinate_trick": 1, "trick": 1, "cpu": 2, "cuda": 3}
DEFAULT_NMS_MODE = "cpu"


def _nms_mode(mode) -> int:
    try:
        return NMS_MODES[mode]
    except KeyError:
        raise ValueError(f"nms mode {mode!r}: expected one of {sorted(NMS_MODES)}") from None


class Detector:
    """Fixed-capacity device buffers + the two kernels of the post-process.  Everything stays on the
    device until `fetch`; capacity = every cell of the three grids, so nothing is ever truncated."""

    def __init__(self, grids: Sequence[int], num_classes: int, device, nms_workspace: torch.Tensor = None,
                 max_fetch: int = 2048):
        """nms_workspace: a suppression-matrix scratch shared with other Detectors whose calls run in order on
        one stream (predict_batch): it is cap * ceil(cap/64) * 8 bytes (79 MB at 640x640, 1.27 GB at 1280)."""
        L.lib()
        self.grids, self.nc, self.device = [int(g) for g in grids], int(num_classes), device
        self.cap = sum(3 * g * g for g in self.grids)
        i32 = dict(device=device, dtype=torch.int32)
        self.boxes = torch.empty(self.cap, 4, device=device, dtype=torch.float32)
        self.scores = torch.empty(self.cap, device=device, dtype=torch.float32)
        self.classes = torch.empty(self.cap, **i32)
        self.count = torch.zeros(1, **i32)
        self.keep = torch.empty(self.cap, **i32)
        self.nkeep = torch.zeros(1, **i32)
        self.ws_c = torch.empty(int(L.lib().yh_candidates_ws(L.int3(self.grids))) + 4, **i32)
        need = int(L.lib().yh_nms_ws(self.cap)) + 256
        if nms_workspace is not None and (nms_workspace.numel() < need or nms_workspace.device != torch.device(device)
                                          or nms_workspace.dtype != torch.uint8):
            raise ValueError("nms_workspace too small / wrong device or dtype")
        self.ws_n = nms_workspace if nms_workspace is not None else torch.empty(need, device=device, dtype=torch.uint8)
        # the result table of yh_gather_detections and its pinned host mirror: header + the first `max_fetch` rows arrive
        # with ONE device->host copy (a longer list takes a second copy of the remaining rows)
        self.max_fetch = min(self.cap, int(max_fetch))
        self.out = torch.zeros(8 + 6 * self.cap, device=device, dtype=torch.float32)
        self.out_pin = torch.zeros(8 + 6 * self.max_fetch, dtype=torch.float32).pin_memory()
        self._out_np = self.out_pin.numpy()

    def candidates(self, preds: Sequence[torch.Tensor], anchors_list, img_size, conf_threshold, pad_left=0.0,
                   pad_top=0.0, scale=1.0, letterbox_dev: torch.Tensor = None):
        for p, g in zip(preds, self.grids):
            if tuple(p.shape) != (1, g, g, 3, 5 + self.nc) or not p.is_contiguous():
                raise ValueError(f"expected contiguous (1,{g},{g},3,{5 + self.nc}) predictions, got {tuple(p.shape)}")
        L.check(L.lib().yh_candidates(L.ptr3(preds), L.floats(_anchors18(anchors_list)), L.int3(self.grids), self.nc,
                                      float(img_size), float(conf_threshold), float(pad_left), float(pad_top),
                                      float(scale), self.boxes.data_ptr(), self.scores.data_ptr(),
                                      self.classes.data_ptr(), self.count.data_ptr(), self.cap, self.ws_c.data_ptr(),
                                      letterbox_dev.data_ptr() if letterbox_dev is not None else None,
                                      _stream(self.device)), "candidates")

    def nms(self, iou_threshold: float, mode: str = DEFAULT_NMS_MODE):
        L.check(L.lib().yh_nms(self.boxes.data_ptr(), self.scores.data_ptr(), self.classes.data_ptr(),
                               self.count.data_ptr(), self.cap, float(iou_threshold), _nms_mode(mode), self.keep.data_ptr(),
                               self.nkeep.data_ptr(), self.ws_n.data_ptr(), _stream(self.device)), "nms")

    def gather(self):
        """Enqueue the result table (yh_gather_detections) and its copy into the pinned mirror; no host sync."""
        L.check(L.lib().yh_gather_detections(self.boxes.data_ptr(), self.scores.data_ptr(), self.classes.data_ptr(),
                                             self.count.data_ptr(), self.keep.data_ptr(), self.nkeep.data_ptr(), self.cap,
                                             self.out.data_ptr(), _stream(self.device)), "gather_detections")
        self.out_pin.copy_(self.out[: self.out_pin.numel()], non_blocking=True)

    def read(self) -> List[Tuple[float, float, float, float, float, int]]:
        """Wait for the stream and turn the pinned table into the reference's list of 6-tuples (train.py:1242-1246)."""
        torch.cuda.current_stream(self.device).synchronize()
        hdr = self._out_np[:2].view(np.int32)
        m, k = int(hdr[0]), int(hdr[1])
        if m > self.cap:
            raise RuntimeError("candidate capacity exceeded")
        if k <= self.max_fetch:
            rows = self._out_np[8: 8 + 6 * k].copy()
        else:
            rows = self.out[8: 8 + 6 * k].cpu().numpy()
        cols = np.ascontiguousarray(rows.reshape(k, 6).T)          # six contiguous columns -> python lists -> zip (all in C)
        return list(zip(cols[0].tolist(), cols[1].tolist(), cols[2].tolist(), cols[3].tolist(), cols[4].tolist(),
                        cols[5].view(np.int32).tolist()))

    def fetch(self) -> List[Tuple[float, float, float, float, float, int]]:
        self.gather()
        return self.read()


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float,
                mode: str = DEFAULT_NMS_MODE) -> torch.Tensor:
    """Same contract as torchvision.ops.batched_nms (the reference's call, train.py:1232-1233): kept indices (int64) in
    descending-score order, IoU > threshold suppresses (threshold compared as a double, like torchvision's CPU kernel).
    `mode` picks torchvision's branch: "cpu" (default) applies its size rule for CPU tensors -- the reference's CPU path:
    coordinate trick for boxes.numel() <= 4000, per class above --, "cuda" its rule for GPU tensors (20000),
    "trick" / "vanilla" force one."""
    if not boxes.is_cuda:
        raise RuntimeError("batched_nms: the HIP path needs GPU tensors; no CPU fallback in this package")
    M = int(boxes.shape[0])
    if M == 0:
        return torch.zeros(0, dtype=torch.int64, device=boxes.device)
    dev = boxes.device
    b = boxes.contiguous().float()
    s = scores.contiguous().float()
    c = idxs.contiguous().to(torch.int32)
    count = torch.tensor([M], device=dev, dtype=torch.int32)
    keep = torch.empty(M, device=dev, dtype=torch.int32)
    nkeep = torch.zeros(1, device=dev, dtype=torch.int32)
    ws = torch.empty(int(L.lib().yh_nms_ws(M)) + 256, device=dev, dtype=torch.uint8)
    L.check(L.lib().yh_nms(b.data_ptr(), s.data_ptr(), c.data_ptr(), count.data_ptr(), M, float(iou_threshold),
                           _nms_mode(mode), keep.data_ptr(), nkeep.data_ptr(), ws.data_ptr(), _stream(dev)), "nms")
    return keep[: int(nkeep.item())].long()


def predict(model, image_path, device, num_classes=1, conf_threshold=0.5, iou_threshold=0.4, nms_mode=DEFAULT_NMS_MODE):
    """Letterbox -> model (eval) -> decode/threshold -> un-letterbox -> global class-aware NMS; returns
    [(x1,y1,x2,y2,conf,class_id)] in original-image pixels (train.py:1114-1250).  The NMS takes the branch of
    torchvision.ops.batched_nms that the reference's CPU path takes for the same number of candidates (`nms_mode`)."""
    from PIL import Image
    model.eval()
    pil = Image.open(image_path).convert("RGB")
    img_size = model.img_size
    pil, scale, pad_top, pad_left = letterbox_resize(pil, img_size)
    x = torch.from_numpy(np.array(pil)).unsqueeze(0).to(device)     # (1,S,S,3) uint8: the /255 of train.py:1137 runs on the device
    with torch.no_grad():
        preds = model(x)
    det = getattr(model, "_detector", None)
    grids = [p.shape[1] for p in preds]
    if det is None or det.grids != grids or det.nc != num_classes or det.device != preds[0].device:
        det = Detector(grids, num_classes, preds[0].device)
        model._detector = det
    det.candidates(preds, model.anchors, img_size, conf_threshold, pad_left, pad_top, scale)
    det.nms(iou_threshold, nms_mode)
    return det.fetch()


def predict_batch(model, images, device, num_classes=1, conf_threshold=0.5, iou_threshold=0.4, nms_mode=DEFAULT_NMS_MODE):
    """`predict` for several images at once: ONE batched forward, then candidate extraction and
    class-aware NMS per image segment (each image keeps its own letterbox parameters and its own candidate set, so
    nothing is suppressed across images); one host synchronisation at the end.  `images`: paths or PIL images.
    Returns a list with, per image, exactly what `predict(model, image, ...)` returns (same boxes, scores, order)."""
    from PIL import Image
    model.eval()
    S = model.img_size
    xs, meta = [], []
    for im in images:
        pil = im if isinstance(im, Image.Image) else Image.open(im)
        pil, scale, pad_top, pad_left = letterbox_resize(pil.convert("RGB"), S)
        xs.append(torch.from_numpy(np.array(pil)))                   # (S,S,3) uint8; /255 on the device (true division)
        meta.append((pad_left, pad_top, scale))
    if not xs:
        return []
    x = torch.stack(xs).to(device)
    with torch.no_grad():
        preds = model(x)
    grids = [p.shape[1] for p in preds]
    dets = getattr(model, "_detectors", None)
    if dets is None or len(dets) < len(xs) or dets[0].grids != grids or dets[0].nc != num_classes or dets[0].device != preds[0].device:
        first = Detector(grids, num_classes, preds[0].device)      # per image: boxes/scores/keep; ONE suppression matrix
        dets = [first] + [Detector(grids, num_classes, preds[0].device, nms_workspace=first.ws_n) for _ in xs[1:]]
        model._detectors = dets
    for b, (pad_left, pad_top, scale) in enumerate(meta):
        dets[b].candidates([p[b:b + 1] for p in preds], model.anchors, S, conf_threshold, pad_left, pad_top, scale)
        dets[b].nms(iou_threshold, nms_mode)
    return [dets[b].fetch() for b in range(len(xs))]


class InferenceSession:
    """bs=1 end-to-end inference (BASELINE config 5): image load (NCHW float -> NHWC, or HWC uint8 -> NHWC with the
    reference's /255), BN-folded fused convs, candidate extraction, global NMS, the result table and its device->host
    copy -- enqueued eagerly per image (the lowest end-to-end latency: the device starts on the first kernel while the host
    enqueues the rest), or with use_graph=True captured ONCE into a hipGraph (torch.cuda.CUDAGraph on the launch stream) and
    replayed per image (less host time per image, better back-to-back throughput, ~0.08 ms more latency: hipGraphLaunch's
    host-side work precedes the first node).  Everything has a fixed address either way: the input image, the letterbox
    parameters {pad_left, pad_top, scale}, all outputs and the pinned host mirror of the result table.

    Per image the host does: one cheap weight-state check, one host->device copy of the image, the launches (or one graph
    launch), one stream synchronisation, and reads the kept rows out of pinned memory.

    Weight changes: anything done through this package (optimizer steps, training forwards, load_state_dict, .to()) or
    through in-place torch ops on the registered parameters / buffers is picked up automatically.  After writes torch cannot
    see on those tensors (`p.data.mul_()`, `trainer.flat_p` views, in-place collectives) call
    `model.invalidate_folded_weights()`.  Re-assigned parameter / buffer attributes (`conv.weight = nn.Parameter(...)`,
    load_state_dict(assign=True) through any parent module) are seen through torch's global registration hooks
    (graph._on_reregistration): the session re-traces and re-captures by itself."""

    def __init__(self, model, conf_threshold=0.5, iou_threshold=0.4, use_graph=False, nms_mode=DEFAULT_NMS_MODE):
        self.model = model.eval()
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise RuntimeError("InferenceSession: the HIP path needs the model on the GPU; no CPU fallback in this package")
        self.device, self.S, self.nc = p0.device, model.img_size, model.num_classes
        self.conf, self.iou, self.nms_mode = float(conf_threshold), float(iou_threshold), nms_mode
        _nms_mode(nms_mode)
        self.x = torch.zeros(1, 3, self.S, self.S, device=self.device)
        self.x_u8 = torch.zeros(1, self.S, self.S, 3, device=self.device, dtype=torch.uint8)
        self.lb = torch.tensor([0.0, 0.0, 1.0], device=self.device)
        self._lb_host = (0.0, 0.0, 1.0)
        self.use_graph = bool(use_graph)
        self.det = None
        self._build()

    def _build(self):
        """Trace the eval plan; hipGraphs are (re)captured lazily per input kind.  Called again when a parameter, BatchNorm
        buffer or the model's mode moved under the plan (HipTrainer adopting the parameters, load_state_dict(assign=True),
        ...): the op lists and the captured graphs hold raw device addresses."""
        model = self.model.eval()
        with torch.no_grad():
            self.plan = model._plan_for(self.x)
        self.heads = [v for v, _ in self.plan.outputs]
        grids = [v.H for v in self.heads]
        if self.det is None or self.det.grids != grids:
            self.det = Detector(grids, self.nc, self.device)
        self.graphs = {}
        self.plan.refresh_folded_weights(_stream(self.device))
        self._fold_inputs = self.plan.fold_inputs()
        self._vsum = sum(t._version for t in self._fold_inputs)
        self._token = state_token()

    def refresh(self):
        """Full re-validation: re-trace if any tensor moved, re-fold if any value may have changed."""
        if self.plan.params_moved():
            self._build()
            return
        self._fold_inputs = self.plan.fold_inputs()
        self._vsum = sum(t._version for t in self._fold_inputs)
        self._token = state_token()
        self.plan.refresh_folded_weights(_stream(self.device))

    def _check_state(self):
        """Per image: one tuple compare + the version counters of the cached fold inputs (no module-tree walk)."""
        if state_token() != self._token:
            self.refresh()
            return
        v = 0
        for t in self._fold_inputs:
            v += t._version
        if v != self._vsum:
            self._vsum = v
            self.plan.refresh_folded_weights(_stream(self.device))

    def _capture(self, kind: str):
        """ONE hipGraph per input kind (image load + forward + post-process).  What a replay buys is HOST time per image (one
        launch call instead of ~75) and back-to-back throughput; its end-to-end latency is hipGraphLaunch's host-side work
        (which precedes the first node) above the eager path's, where the device starts on the first kernel while the host is
        still enqueueing the rest: profiles/r03_infer_latency.json 1.422 / 1.354 ms replayed against 1.340 / 1.274 ms eager
        (fp32 / uint8 image).  Round 3 split the replay into a short head graph and a tail graph to hide that prelude; the
        stored records do not show a gain, so the split is gone and `use_graph` defaults to False."""
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._enqueue(kind)            # warm-up: kernel attributes, lazy module state
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._enqueue(kind)
        self.graphs[kind] = graph
        return graph

    def _enqueue(self, kind: str):
        """Everything a replay repeats: image load, forward list, post-process.  BatchNorm folding is NOT in here: it depends on the
        weights only and runs (eagerly, before the replay) when the weight state changed."""
        ops, n = self.plan.fwd_ops
        self.model._load_input(self.plan, self.x if kind == "f32" else self.x_u8)
        L.run_ops(ops, n, _stream(self.device), self.plan._ctx())
        preds = [v.buf.data.view(1, v.H, v.W, 3, v.C // 3) for v in self.heads]
        self.det.candidates(preds, self.model.anchors, self.S, self.conf, letterbox_dev=self.lb)
        self.det.nms(self.iou, self.nms_mode)
        self.det.gather()

    def run(self, img: torch.Tensor, pad_left=0.0, pad_top=0.0, scale=1.0, fetch=True):
        """img: (1,3,S,S) / (3,S,S) float32 in [0,1], or (S,S,3) / (1,S,S,3) uint8 image bytes (divided by 255 on the
        device exactly like train.py:1137), host or device; pinned host tensors copy asynchronously.  Returns the
        detections list [(x1,y1,x2,y2,conf,class_id)] (fetch=False: enqueue only, `session.det.read()` later)."""
        self._check_state()
        kind = "u8" if img.dtype == torch.uint8 else "f32"
        dst = self.x_u8 if kind == "u8" else self.x
        dst.copy_(img.reshape(dst.shape), non_blocking=True)
        lb = (float(pad_left), float(pad_top), float(scale))
        if lb != self._lb_host:
            self.lb.copy_(torch.tensor(lb, dtype=torch.float32))
            self._lb_host = lb
        if self.use_graph:
            (self.graphs.get(kind) or self._capture(kind)).replay()
        else:
            self._enqueue(kind)
        return self.det.read() if fetch else None


def assign_targets_gpu(labels, img_size, num_classes, device, anchors=None):
    """labels[b] = [(class, xc, yc, w, h), ...] normalised to the padded square image -> the three dense
    (B,G,G,3,5+nc) target tensors, built on the device by yh_assign_targets (the rule of
    YOLODataset.__getitem__, train.py:164-205)."""
    from .modules import DEFAULT_ANCHORS
    L.lib()
    B = len(labels)
    maxn = max(1, max((len(l) for l in labels), default=1))
    lab = torch.zeros(B, maxn, 5, dtype=torch.float64)
    cnt = torch.zeros(B, dtype=torch.int32)
    for b, l in enumerate(labels):
        cnt[b] = len(l)
        if l:
            lab[b, :len(l)] = torch.tensor([[float(v) for v in row] for row in l], dtype=torch.float64)
    lab, cnt = lab.to(device), cnt.to(device)
    grids = [img_size // 8, img_size // 16, img_size // 32]
    out = [torch.empty(B, g, g, 3, 5 + num_classes, device=device, dtype=torch.float32) for g in grids]
    a18 = [float(v) for sc in (anchors or DEFAULT_ANCHORS) for pair in sc for v in pair]
    L.check(L.lib().yh_assign_targets(lab.data_ptr(), cnt.data_ptr(), B, maxn, L.floats(a18), L.int3(grids), num_classes,
                                      int(img_size), L.ptr3(out), _stream(device)), "assign_targets")
    return out
