#!/usr/bin/env python3
"""PCIe-inclusive training throughput: host batches -> device -> fused step, three ways:
  resident   images and targets already in HBM (what bench.py times)
  reference  float32 NCHW images + dense targets in pageable host memory, `.to(device)` inside the loop (train.py:897-903)
  prefetch   uint8 HWC images + raw labels through DevicePrefetcher (pinned, async, /255 and target assignment on the GPU)

    python tools/pipeline_bench.py [--batch 64] [--steps 20]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
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
    mb_ref = (f32[0].numel() * 4 + sum(t.numel() for t in dense_host) * 4) / 1e6
    mb_raw = (u8[0].numel() + lab.numel() * 8 + cnt.numel() * 4) / 1e6
    print({k: round(v, 1) for k, v in res.items()}, f"img/s; host->device MB per batch: reference {mb_ref:.1f}, prefetch {mb_raw:.1f}")


if __name__ == "__main__":
    main()
