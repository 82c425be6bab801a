"""Device-side input pipeline for the training step (the reference feeds `train_epoch` from a
single-process PIL DataLoader, train.py:1470-1474, and builds dense targets per sample in `YOLODataset.__getitem__`,
train.py:140-205).

`DevicePrefetcher` wraps any loader and keeps `depth` batches in flight: each batch is copied host->device on a
dedicated copy stream (asynchronously when it arrives in pinned memory, e.g. from DataLoader(pin_memory=True)) while the previous step computes, and handed over with an event wait (no host sync).  The
device-side tensors are STATIC per slot (depth + 1 slots, no allocator traffic in steady state): a yielded batch stays
valid until the consumer has asked for the batch after the next one -- copy it if it must live longer
(`static_buffers=False` allocates fresh tensors per batch instead).  With `YOLODataset(raw=True)` + `raw_collate_fn` the loader ships uint8 HWC image bytes and a few KB of
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
    def __init__(self, loader: Iterable, device, img_size: int = 640, num_classes: int = 1, anchors=None, depth: int = 2,
                 static_buffers: bool = True):
        self.loader, self.device = loader, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher: the HIP path needs a GPU device; no CPU fallback in this package")
        L.lib()
        self.img_size, self.nc, self.depth = int(img_size), int(num_classes), max(1, int(depth))
        self.static = bool(static_buffers)
        from .modules import DEFAULT_ANCHORS
        a = anchors if anchors is not None else DEFAULT_ANCHORS
        self.a18 = [float(v) for sc in a for pair in sc for v in (pair.tolist() if torch.is_tensor(pair) else pair)]
        self.copy_stream = torch.cuda.Stream(self.device)
        self._slots: List[dict] = [dict() for _ in range(self.depth + 1)]      # staging slots, reused round-robin
        self._next = 0

    def __len__(self):
        return len(self.loader)

    # ---- staging -----------------------------------------------------------------------------------------------
    def _h2d(self, slot: dict, name: str, t: torch.Tensor) -> torch.Tensor:
        """Host tensor -> this slot's device tensor on the copy stream.  Pinned sources (DataLoader(pin_memory=True)) are
        copied asynchronously; pageable ones are copied directly (the call blocks the host for the transfer, the compute
        stream keeps running).  Re-staging pageable data through a pinned buffer here is NOT worth it on this platform:
        pinned host memory is uncached for the CPU (measured 39 ms to fill 79 MB vs 6 ms for the direct copy)."""
        if t.is_pinned():
            slot["h_" + name] = t               # keep it alive until this slot is reused (the async copy reads it)
        if not self.static:
            return t.to(self.device, non_blocking=True)
        d = slot.get("d_" + name)
        if d is None or d.shape != t.shape or d.dtype != t.dtype:
            d = torch.empty(t.shape, dtype=t.dtype, device=self.device)
            slot["d_" + name] = d
        d.copy_(t, non_blocking=True)
        return d

    def _stage(self, batch):
        """Issue the H2D copies (and the target assignment) of one batch on the copy stream."""
        slot = self._slots[self._next]
        self._next = (self._next + 1) % len(self._slots)
        if slot.get("ready") is not None:
            slot["ready"].synchronize()         # the copies that last read this slot's pinned buffers have finished
        with torch.cuda.stream(self.copy_stream):
            if slot.get("consumed") is not None:
                self.copy_stream.wait_event(slot["consumed"])   # the step that read this slot's device tensors is done
            if len(batch) == 3:                 # raw mode: uint8 images, padded labels, counts
                imgs, lab, cnt = batch
                d_imgs = self._h2d(slot, "imgs", imgs)
                d_lab, d_cnt = self._h2d(slot, "lab", lab), self._h2d(slot, "cnt", cnt)
                B, maxn = int(lab.shape[0]), int(lab.shape[1])
                grids = [self.img_size // 8, self.img_size // 16, self.img_size // 32]
                targets = slot.get("d_targets") if self.static else None
                if targets is None or targets[0].shape[0] != B:
                    targets = [torch.empty(B, g, g, 3, 5 + self.nc, device=self.device, dtype=torch.float32) for g in grids]
                    if self.static:
                        slot["d_targets"] = targets
                L.check(L.lib().yh_assign_targets(d_lab.data_ptr(), d_cnt.data_ptr(), B, maxn, L.floats(self.a18), L.int3(grids),
                                                  self.nc, self.img_size, L.ptr3(targets), self.copy_stream.cuda_stream),
                        "assign_targets")
                keep = (d_lab, d_cnt)
            else:                               # the reference's batch: float images, list[B][3] of dense targets
                imgs, tl = batch
                d_imgs = self._h2d(slot, "imgs", imgs)
                if len(tl) == 3 and all(torch.is_tensor(t) and t.dim() == 5 for t in tl):
                    stacked = list(tl)
                else:
                    stacked = [torch.stack([t[s] for t in tl]) for s in range(3)]
                targets = [self._h2d(slot, f"t{s}", stacked[s]) for s in range(3)]
                keep = ()
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
            slot["ready"] = ready
        return slot, d_imgs, targets, keep

    def __iter__(self):
        it = iter(self.loader)
        queue = []
        for _ in range(self.depth):
            b = next(it, None)
            if b is None:
                break
            queue.append(self._stage(b))
        while queue:
            slot, d_imgs, targets, keep = queue.pop(0)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(slot["ready"])
            if not self.static:
                for t in (d_imgs, *targets, *keep):      # allocator: these blocks are in use on the consumer's stream too
                    t.record_stream(cur)
            yield d_imgs, targets
            # the consumer is back for the next batch: everything it enqueued on its stream so far used this slot
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            slot["consumed"] = done
            nxt = next(it, None)
            if nxt is not None:
                queue.append(self._stage(nxt))
