"""Device-side input pipeline for the training step (the reference feeds `train_epoch` from a
single-process PIL DataLoader, train.py:1470-1474, and builds dense targets per sample in `YOLODataset.__getitem__`,
train.py:140-205).

`DevicePrefetcher` wraps any loader and keeps `depth` batches in flight: each batch is staged in pinned host memory,
copied host->device on a dedicated copy stream while the previous step computes, and handed over with an event wait
(no host sync).  With `YOLODataset(raw=True)` + `raw_collate_fn` the loader ships uint8 HWC image bytes and a few KB of
letterboxed labels; `/255` (yh_u8hwc_to_nhwc, inside the model's input load) and the dense target tensors
(yh_assign_targets, the rule of train.py:164-205) are produced on the device: 79 MB instead of 354 MB over PCIe per
64-image 640x640 batch.  Batches come out in loader order with the same values as the reference's path (bit-identical
images, bit-identical targets).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch

from . import _lib as L


class DevicePrefetcher:
    def __init__(self, loader: Iterable, device, img_size: int = 640, num_classes: int = 1, anchors=None, depth: int = 2):
        self.loader, self.device = loader, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher: the HIP path needs a GPU device; no CPU fallback in this package")
        L.lib()
        self.img_size, self.nc, self.depth = int(img_size), int(num_classes), max(1, int(depth))
        from .modules import DEFAULT_ANCHORS
        a = anchors if anchors is not None else DEFAULT_ANCHORS
        self.a18 = [float(v) for sc in a for pair in sc for v in (pair.tolist() if torch.is_tensor(pair) else pair)]
        self.copy_stream = torch.cuda.Stream(self.device)
        self._pinned: List[dict] = [dict() for _ in range(self.depth + 1)]     # staging slots, reused round-robin
        self._slot = 0

    def __len__(self):
        return len(self.loader)

    # ---- staging -----------------------------------------------------------------------------------------------
    def _pin(self, slot: dict, name: str, t: torch.Tensor) -> torch.Tensor:
        buf = slot.get(name)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            slot[name] = buf
        buf.copy_(t)
        return buf

    def _stage(self, batch):
        """Issue the H2D copies (and the target assignment) of one batch on the copy stream."""
        slot = self._pinned[self._slot]
        self._slot = (self._slot + 1) % len(self._pinned)
        ev_free = slot.get("free")
        if ev_free is not None:
            ev_free.synchronize()               # the copies that last read this slot's pinned buffers have finished
        with torch.cuda.stream(self.copy_stream):
            if len(batch) == 3:                 # raw mode: uint8 images, padded labels, counts
                imgs, lab, cnt = batch
                d_imgs = self._pin(slot, "imgs", imgs).to(self.device, non_blocking=True)
                d_lab = self._pin(slot, "lab", lab).to(self.device, non_blocking=True)
                d_cnt = self._pin(slot, "cnt", cnt).to(self.device, non_blocking=True)
                B, maxn = int(lab.shape[0]), int(lab.shape[1])
                grids = [self.img_size // 8, self.img_size // 16, self.img_size // 32]
                targets = [torch.empty(B, g, g, 3, 5 + self.nc, device=self.device, dtype=torch.float32) for g in grids]
                L.check(L.lib().yh_assign_targets(d_lab.data_ptr(), d_cnt.data_ptr(), B, maxn, L.floats(self.a18), L.int3(grids),
                                                  self.nc, self.img_size, L.ptr3(targets), self.copy_stream.cuda_stream),
                        "assign_targets")
                keep = (d_lab, d_cnt)
            else:                               # the reference's batch: float images, list[B][3] of dense targets
                imgs, tl = batch
                d_imgs = self._pin(slot, "imgs", imgs).to(self.device, non_blocking=True)
                if len(tl) == 3 and all(torch.is_tensor(t) and t.dim() == 5 for t in tl):
                    stacked = list(tl)
                else:
                    stacked = [torch.stack([t[s] for t in tl]) for s in range(3)]
                targets = [self._pin(slot, f"t{s}", stacked[s]).to(self.device, non_blocking=True) for s in range(3)]
                keep = ()
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
            slot["free"] = ready
        return d_imgs, targets, ready, keep

    def __iter__(self):
        it = iter(self.loader)
        queue = []
        for _ in range(self.depth):
            b = next(it, None)
            if b is None:
                break
            queue.append(self._stage(b))
        while queue:
            d_imgs, targets, ready, keep = queue.pop(0)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ready)
            for t in (d_imgs, *targets, *keep):          # allocator: these blocks are in use on the consumer's stream too
                t.record_stream(cur)
            nxt = next(it, None)
            if nxt is not None:
                queue.append(self._stage(nxt))
            yield d_imgs, targets
