"""Host-only pieces of the reference surface that sit on either side of the device hot path.

These are scalar / file-IO helpers the reference's callers import next to the model (letterbox,
dataset target assignment, LR schedule, python-list NMS, grid-cell evaluation metrics).  SURVEY.md
section 8 marks them "out of scope for kernels"; they are restated here so `predict`, `train_epoch`
and `eval_epoch` remain drop-in.  Nothing in this file is on the timed path.
"""
from __future__ import annotations

import glob
import math
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

from .modules import DEFAULT_ANCHORS

# width / depth multipliers per model size (train.py:1346-1352)
YOLO_SIZES = {"n": (0.25, 0.33), "s": (0.50, 0.33), "m": (0.75, 0.67), "l": (1.00, 1.00), "x": (1.25, 1.33)}


def letterbox_resize(image, target_size=640, pad_color=(114, 114, 114)):
    """Aspect-preserving bilinear resize onto a grey square; returns (image, scale, pad_top, pad_left)
    with new = int(dim*scale) and centred padding (train.py:15-58)."""
    from PIL import Image
    w0, h0 = image.size
    scale = min(target_size / w0, target_size / h0)
    nw, nh = int(w0 * scale), int(h0 * scale)
    bilinear = Image.Resampling.BILINEAR if hasattr(Image, "Resampling") else 2
    canvas = Image.new("RGB", (target_size, target_size), pad_color)
    left, top = (target_size - nw) // 2, (target_size - nh) // 2
    canvas.paste(image.resize((nw, nh), bilinear), (left, top))
    return canvas, scale, top, left


def shape_iou(box_wh, anchors):
    """IoU of a (w,h) box with (A,2) anchors, both centred at the origin (train.py:108-131)."""
    inter = torch.min(box_wh[0], anchors[:, 0]) * torch.min(box_wh[1], anchors[:, 1])
    return inter / (box_wh[0] * box_wh[1] + anchors[:, 0] * anchors[:, 1] - inter + 1e-16)


class YOLODataset(Dataset):
    """images/*.jpg|png + ../labels/<stem>.txt -> (img (3,S,S) in [0,1], [t_p3,t_p4,t_p5]) with the
    reference's assignment rule: best shape-IoU anchor over all nine, cell = min(int(c*G), G-1), first
    writer wins, one-hot class (train.py:60-207)."""

    def __init__(self, img_dir, num_classes=1, anchors=None, img_size=640, raw=False):
        # raw=True (not in the reference): __getitem__ returns (uint8 (S,S,3) image, float64 (n,5) letterboxed labels)
        # for the device-side pipeline (pipeline.DevicePrefetcher): /255 and the target assignment run on the GPU
        self.raw = raw
        self.imgs = sorted(glob.glob(f"{img_dir}/*.jpg") + glob.glob(f"{img_dir}/*.png"))
        self.labels = [str(Path(p).parent.parent / "labels" / f"{Path(p).stem}.txt") for p in self.imgs]
        self.num_classes, self.img_size = num_classes, img_size
        self.grid_size_p3, self.grid_size_p4, self.grid_size_p5 = img_size // 8, img_size // 16, img_size // 32
        self.grid_sizes = [self.grid_size_p3, self.grid_size_p4, self.grid_size_p5]
        self.strides = [8, 16, 32]
        if anchors is None:
            self.anchors = [torch.tensor(a, dtype=torch.float32) for a in DEFAULT_ANCHORS]
        elif isinstance(anchors[0][0], list):
            self.anchors = [torch.tensor(a, dtype=torch.float32) for a in anchors]
        else:
            one = anchors.clone().detach() if isinstance(anchors, torch.Tensor) else torch.tensor(anchors, dtype=torch.float32)
            self.anchors = [one] * 3
        self.num_anchors_per_scale = 3
        self.output_dim = 5 + num_classes

    def __len__(self):
        return len(self.imgs)

    def compute_anchor_iou(self, box_wh, anchors):
        return shape_iou(box_wh, anchors)

    def __getitem__(self, idx):
        from PIL import Image
        pil = Image.open(self.imgs[idx]).convert("RGB")
        w0, h0 = pil.size
        pil, scale, top, left = letterbox_resize(pil, self.img_size)
        S = self.img_size
        if self.raw:
            rows = []
            if Path(self.labels[idx]).exists():
                with open(self.labels[idx], encoding="utf-8") as fh:
                    for line in fh:
                        f = line.strip().split()
                        if len(f) != 5:
                            continue
                        xc, yc, w, h = (float(v) for v in f[1:])
                        rows.append([float(int(float(f[0]))), (xc * w0 * scale + left) / S, (yc * h0 * scale + top) / S,
                                     (w * w0 * scale) / S, (h * h0 * scale) / S])
            lab = torch.tensor(rows, dtype=torch.float64).reshape(-1, 5)
            return torch.from_numpy(np.array(pil)), lab
        img = torch.from_numpy(np.array(pil)).permute(2, 0, 1).float() / 255.0
        targets = [torch.zeros((g, g, 3, self.output_dim)) for g in self.grid_sizes]
        if Path(self.labels[idx]).exists():
            with open(self.labels[idx], encoding="utf-8") as fh:
                for line in fh:
                    f = line.strip().split()
                    if len(f) != 5:
                        continue
                    cid = int(float(f[0]))
                    xc, yc, w, h = (float(v) for v in f[1:])
                    xc, yc = (xc * w0 * scale + left) / S, (yc * h0 * scale + top) / S
                    w, h = (w * w0 * scale) / S, (h * h0 * scale) / S
                    wh = torch.tensor([w * S, h * S])
                    best, bs, ba = -1, 0, 0
                    for s in range(3):
                        iou = shape_iou(wh, self.anchors[s])
                        if iou.max().item() > best:
                            best, bs, ba = iou.max().item(), s, iou.argmax().item()
                    g = self.grid_sizes[bs]
                    gx, gy = min(int(xc * g), g - 1), min(int(yc * g), g - 1)
                    t = targets[bs]
                    if t[gy, gx, ba, 4] == 0:
                        t[gy, gx, ba, 0:4] = torch.tensor([xc, yc, w, h])
                        t[gy, gx, ba, 4] = 1.0
                        t[gy, gx, ba, 5 if self.num_classes == 1 else 5 + cid] = 1.0
        return img, targets


def yolo_collate_fn(batch):
    """Stack images; keep per-sample target lists (train.py:209-222)."""
    return torch.stack([b[0] for b in batch]), [b[1] for b in batch]


def raw_collate_fn(batch):
    """Collate for YOLODataset(raw=True): (B,S,S,3) uint8 images, (B,maxn,5) float64 labels (zero-padded), (B,) int32
    label counts -- a few KB of labels instead of 38 MB of dense targets per 64-image batch."""
    imgs = torch.stack([b[0] for b in batch])
    maxn = max(1, max(int(b[1].shape[0]) for b in batch))
    lab = torch.zeros(len(batch), maxn, 5, dtype=torch.float64)
    cnt = torch.zeros(len(batch), dtype=torch.int32)
    for i, (_, l) in enumerate(batch):
        n = int(l.shape[0])
        cnt[i] = n
        if n:
            lab[i, :n] = l
    return imgs, lab, cnt


def stack_targets(targets, device):
    """list[B][3] of (G,G,3,5+nc) -> three (B,G,G,3,5+nc) device tensors (train.py:900-903).  Three already
    stacked device tensors (what pipeline.DevicePrefetcher yields) pass through."""
    if len(targets) == 3 and all(torch.is_tensor(t) and t.dim() == 5 for t in targets):
        return [t.to(device, non_blocking=True) for t in targets]
    return [torch.stack([t[s] for t in targets]).to(device, non_blocking=True) for s in range(3)]


# ---- python-list NMS API (train.py:1064-1112): host scalars, class-agnostic, IoU >= thr suppresses ----
def compute_iou_corners(box1, box2):
    x11, y11, x12, y12 = box1[0:4]
    x21, y21, x22, y22 = box2[0:4]
    inter = max(0, min(x12, x22) - max(x11, x21)) * max(0, min(y12, y22) - max(y11, y21))
    union = (x12 - x11) * (y12 - y11) + (x22 - x21) * (y22 - y21) - inter
    return inter / union if union > 0 else 0


def nms(detections, iou_threshold):
    """Greedy NMS over a python list of (x1,y1,x2,y2,conf,cls); the reference keeps this next to the
    tensor path and pins it with known-answer tests.  Host-only by construction (list of tuples)."""
    pending = sorted(detections, key=lambda d: d[4], reverse=True)
    kept = []
    while pending:
        kept.append(pending[0])
        pending = [d for d in pending[1:] if compute_iou_corners(kept[-1], d) < iou_threshold]
    return kept


def compute_box_iou(box1, box2):
    """Centre-format IoU with eps 1e-6 in the denominator (train.py:928-958); eval metric only.
    Works on 4-vectors and on (4, N) stacks alike."""
    ax1, ax2, ay1, ay2 = box1[0] - box1[2] / 2, box1[0] + box1[2] / 2, box1[1] - box1[3] / 2, box1[1] + box1[3] / 2
    bx1, bx2, by1, by2 = box2[0] - box2[2] / 2, box2[0] + box2[2] / 2, box2[1] - box2[3] / 2, box2[1] + box2[3] / 2
    iw = torch.clamp(torch.min(ax2, bx2) - torch.max(ax1, bx1), min=0)
    ih = torch.clamp(torch.min(ay2, by2) - torch.max(ay1, by1), min=0)
    inter = iw * ih
    union = (ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter
    return inter / (union + 1e-6)


def get_lr_lambda(warmup_epochs=3, total_epochs=100, initial_lr=1e-2, min_lr=1e-4, warmup_start_lr=1e-6):
    """Per-epoch LR multiplier: linear warm-up then cosine decay to min_lr (train.py:1034-1062)."""
    def lr_lambda(epoch):
        if epoch < warmup_epochs:
            return (warmup_start_lr + (initial_lr - warmup_start_lr) * epoch / warmup_epochs) / initial_lr
        progress = (epoch - warmup_epochs) / (total_epochs - warmup_epochs)
        return (min_lr + (initial_lr - min_lr) * 0.5 * (1.0 + np.cos(np.pi * progress))) / initial_lr
    return lr_lambda


def compute_optimal_anchors(dataset_yaml, img_size=640, num_anchors=9):
    """k-means (k=9, random_state=0, n_init=10) over label w,h in pixels, sorted by area, split 3/3/3
    (train.py:1252-1343).  Returns None when sklearn or labels are missing."""
    try:
        from sklearn.cluster import KMeans
    except ImportError:
        return None
    import yaml
    with open(dataset_yaml) as fh:
        cfg = yaml.safe_load(fh)
    train_dir = Path(cfg["train"])
    wh = []
    for lab in sorted((train_dir.parent / "labels").glob("*.txt")):
        for line in open(lab, encoding="utf-8"):
            f = line.strip().split()
            if len(f) == 5:
                wh.append([float(f[3]) * img_size, float(f[4]) * img_size])
    if len(wh) < num_anchors:
        return None
    km = KMeans(n_clusters=num_anchors, random_state=0, n_init=10).fit(np.array(wh))
    c = km.cluster_centers_
    c = c[np.argsort(c[:, 0] * c[:, 1])]
    out = [[int(round(w)), int(round(h))] for w, h in c]
    return [out[0:3], out[3:6], out[6:9]]


def synthetic_targets(batch, num_classes, img_size, n_obj=8, seed=2000, anchors=None):
    """Seeded synthetic labels (centres U(0.05,0.95)^2, log-uniform w,h in [8,320] px,
    uniform class) assigned to the three target grids with YOLODataset's rule.  Host code, built once
    before the timed region; returns three (B,G,G,3,5+nc) CPU tensors."""
    rng = np.random.default_rng(seed)
    anc = [torch.tensor(a, dtype=torch.float32) for a in (anchors or DEFAULT_ANCHORS)]
    grids = [img_size // 8, img_size // 16, img_size // 32]
    out = [torch.zeros(batch, g, g, 3, 5 + num_classes) for g in grids]
    for b in range(batch):
        ctr = rng.uniform(0.05, 0.95, size=(n_obj, 2))
        wh = np.exp(rng.uniform(math.log(8), math.log(320), size=(n_obj, 2))) / img_size
        cls = rng.integers(0, max(num_classes, 1), size=n_obj)
        for o in range(n_obj):
            xc, yc, w, h = float(ctr[o, 0]), float(ctr[o, 1]), float(wh[o, 0]), float(wh[o, 1])
            box = torch.tensor([w * img_size, h * img_size])
            best, bs, ba = -1.0, 0, 0
            for s in range(3):
                iou = shape_iou(box, anc[s])
                if iou.max().item() > best:
                    best, bs, ba = iou.max().item(), s, int(iou.argmax())
            g = grids[bs]
            gx, gy = min(int(xc * g), g - 1), min(int(yc * g), g - 1)
            t = out[bs][b]
            if t[gy, gx, ba, 4] == 0:
                t[gy, gx, ba, 0:4] = torch.tensor([xc, yc, w, h])
                t[gy, gx, ba, 4] = 1.0
                t[gy, gx, ba, 5 if num_classes == 1 else 5 + int(cls[o])] = 1.0
    return out


# ---- checkpoint format of the reference (train.py:1533-1540, load paths 1410-1417 / 1431-1438) -------------------
def save_checkpoint(model, epoch, path):
    """The reference's per-epoch dict: {'model': state_dict, 'epoch', 'num_classes', 'img_size', 'width_mult',
    'depth_mult'}.  Tensors are saved on the CPU with OIHW weights, so the file loads in the reference too."""
    torch.save({"model": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "epoch": int(epoch),
                "num_classes": model.num_classes, "img_size": model.img_size, "width_mult": model.width_mult,
                "depth_mult": model.depth_mult}, path)


def load_checkpoint(path, num_classes=None, device="cpu"):
    """Rebuild YOLO from a checkpoint's metadata and load its weights (the reference's inference/eval load
    path).  Reference-written files carry stride-0 expanded grid_* buffers (which makes the
    reference's own load_state_dict fail on torch 2.x): here they are copied into contiguous buffers."""
    from .modules import YOLO
    ckpt = torch.load(path, map_location="cpu", weights_only=True)      # plain tensors + ints/floats only
    nc = num_classes if num_classes is not None else ckpt.get("num_classes", 1)
    model = YOLO(num_classes=nc, img_size=ckpt.get("img_size", 640), width_mult=ckpt.get("width_mult", 0.50),
                 depth_mult=ckpt.get("depth_mult", 0.33))
    model.load_state_dict(ckpt["model"])
    return model.to(device), ckpt.get("epoch", 0)
