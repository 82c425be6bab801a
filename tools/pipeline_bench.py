#!/usr/bin/env python3
"""PCIe-inclusive training throughput: host batches -> device -> fused step, three ways:
  resident   images and targets already in HBM (what bench.py times)
  reference  float32 NCHW images + dense targets in pageable host memory, `.to(device)` inside the loop (train.py:897-903)
  prefetch   uint8 HWC images + raw labels through DevicePrefetcher (pinned, async, /255 and target assignment on the GPU)

  dataset    (--png N) a directory of N synthetic 640x480 PNG files + label files read by YOLODataset through a torch DataLoader
             (decode + letterbox in worker processes): reference style (float images + dense targets built on the host,
             train.py:60-222, 897-903) and the raw uint8 path through DevicePrefetcher

    python tools/pipeline_bench.py [--batch 64] [--steps 20] [--png 512] [--workers 14]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_png_dataset(root, n, seed=3):
    """n photographs' worth of synthetic 640x480 RGB PNGs (smooth gradients + rectangles + mild noise: ~250-400 KB each, i.e. a
    realistic decode cost, unlike white noise) and YOLO label files with 1-6 boxes."""
    import numpy as np
    from PIL import Image
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    os.makedirs(os.path.join(root, "labels"), exist_ok=True)
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:480, 0:640]
    for i in range(n):
        ph = rng.uniform(0, 6.28, 3)
        img = np.stack([127 + 90 * np.sin(xx / rng.uniform(40, 200) + ph[0]) * np.cos(yy / rng.uniform(40, 200)),
                        127 + 90 * np.sin((xx + yy) / rng.uniform(60, 240) + ph[1]),
                        127 + 90 * np.cos(yy / rng.uniform(30, 150) + ph[2])], -1)
        rows = []
        for _ in range(int(rng.integers(1, 7))):
            w, h = rng.uniform(0.05, 0.3, 2)
            xc, yc = rng.uniform(w / 2, 1 - w / 2), rng.uniform(h / 2, 1 - h / 2)
            x0, x1, y0, y1 = int((xc - w / 2) * 640), int((xc + w / 2) * 640), int((yc - h / 2) * 480), int((yc + h / 2) * 480)
            img[y0:y1, x0:x1] = rng.uniform(0, 255, 3)
            rows.append(f"0 {xc:.6f} {yc:.6f} {w:.6f} {h:.6f}")
        img = np.clip(img + rng.normal(0, 4, img.shape), 0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(root, "images", f"im{i:05d}.png"))
        with open(os.path.join(root, "labels", f"im{i:05d}.txt"), "w") as fh:
            fh.write("\n".join(rows) + "\n")
    return os.path.join(root, "images")


def dataset_legs(a, y, tr, dev, S, nc, res, timed):
    import tempfile
    from torch.utils.data import DataLoader
    from yolo_from_scratch_amd.hostside import YOLODataset, raw_collate_fn, yolo_collate_fn
    with tempfile.TemporaryDirectory(dir="/tmp") as root:
        t0 = time.perf_counter()
        img_dir = make_png_dataset(root, a.png)
        sz = sum(os.path.getsize(os.path.join(img_dir, f)) for f in os.listdir(img_dir)) / a.png / 1e3
        print(f"dataset: {a.png} PNG files of 640x480, {sz:.0f} KB each on average, written in {time.perf_counter() - t0:.1f} s")
        nb = a.png // a.batch
        kw = dict(batch_size=a.batch, shuffle=False, num_workers=a.workers, drop_last=True, persistent_workers=True, prefetch_factor=2)

        def epochs(loader, conv):            # a.steps + 3 batches out of as many passes over the files as that takes
            n = 0
            while n < a.steps + 3:
                for b in loader:
                    yield conv(b)
                    n += 1
                    if n == a.steps + 3:
                        return

        # reference style: float32 NCHW images + per-sample dense targets built by the workers, stacked and moved in the loop
        ref = DataLoader(YOLODataset(img_dir, nc, img_size=S), collate_fn=yolo_collate_fn, **kw)

        def ref_conv(b):
            imgs, tg = b
            return imgs.to(dev), [torch.stack([t[s] for t in tg]).to(dev) for s in range(3)]
        timed("dataset_reference_style", epochs(ref, ref_conv))
        del ref
        raw = DataLoader(YOLODataset(img_dir, nc, img_size=S, raw=True), collate_fn=raw_collate_fn, pin_memory=True, **kw)

        class Repeat:                        # DevicePrefetcher iterates its loader once: hand it the multi-pass stream
            def __iter__(self): return epochs(raw, lambda b: b)
            def __len__(self): return a.steps + 3
        timed("dataset_prefetch_pinned", y.DevicePrefetcher(Repeat(), dev, img_size=S, num_classes=nc, depth=2))
        res["dataset_files"], res["dataloader_workers"], res["batches_per_pass"] = a.png, a.workers, nb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--png", type=int, default=0, help="also time a DataLoader over this many generated PNG files")
    ap.add_argument("--workers", type=int, default=14)
    a = ap.parse_args()
    import yolo_from_scratch_amd as y
    dev = torch.device("cuda:0")
    B, S, nc = a.batch, 640, 1
    torch.manual_seed(0)
    model = y.YOLO(num_classes=nc, img_size=S).to(dev).train()
    tr = y.HipTrainer(model)
    g = torch.Generator().manual_seed(1)
    u8 = [torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g) for _ in range(4)]
    lab = torch.zeros(B, 8, 5, dtype=torch.float64)
    lab[:, :, 1:3] = torch.rand(B, 8, 2, generator=g, dtype=torch.float64) * 0.8 + 0.1
    lab[:, :, 3:5] = torch.rand(B, 8, 2, generator=g, dtype=torch.float64) * 0.15 + 0.02
    cnt = torch.full((B,), 8, dtype=torch.int32)
    labels = [[tuple([0] + r[1:]) for r in rows] for rows in lab.tolist()]
    dense_dev = y.assign_targets_gpu(labels, S, nc, dev)
    dense_host = [t.cpu() for t in dense_dev]
    f32 = [(b.permute(0, 3, 1, 2).float() / 255.0).contiguous() for b in u8]
    res = {}

    def timed(name, it):
        n, t0 = 0, None
        for imgs, tg in it:
            if n == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            tr.step(imgs, tg)
            n += 1
        torch.cuda.synchronize()
        res[name] = B * (n - 3) / (time.perf_counter() - t0)

    x_dev = f32[0].to(dev)
    timed("resident", ((x_dev, dense_dev) for _ in range(a.steps + 3)))
    timed("reference", ((f32[i % 4].to(dev), [t.to(dev) for t in dense_host]) for i in range(a.steps + 3)))
    loader = [(u8[i % 4], lab, cnt) for i in range(a.steps + 3)]
    timed("prefetch_pageable", y.DevicePrefetcher(loader, dev, img_size=S, num_classes=nc, depth=2))
    u8p = [b.pin_memory() for b in u8]              # what DataLoader(pin_memory=True) hands over
    loader = [(u8p[i % 4], lab.pin_memory(), cnt.pin_memory()) for i in range(a.steps + 3)]
    timed("prefetch_pinned", y.DevicePrefetcher(loader, dev, img_size=S, num_classes=nc, depth=2))
    t0 = time.perf_counter()
    tmp = torch.empty_like(u8[0])
    for _ in range(5):
        tmp.copy_(u8[1])
    print(f"host memcpy of one uint8 batch: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms")
    if a.png:
        dataset_legs(a, y, tr, dev, S, nc, res, timed)
    mb_ref = (f32[0].numel() * 4 + sum(t.numel() for t in dense_host) * 4) / 1e6
    mb_raw = (u8[0].numel() + lab.numel() * 8 + cnt.numel() * 4) / 1e6
    print({k: round(v, 1) for k, v in res.items()}, f"img/s; host->device MB per batch: reference {mb_ref:.1f}, prefetch {mb_raw:.1f}")


if __name__ == "__main__":
    main()
