#!/usr/bin/env python3
"""BASELINE config 5: bs=1 640x640 end-to-end inference latency (H2D image copy, NCHW->NHWC, BN-folded fused
forward, candidate extraction, global NMS, D2H of the detections), eager launches vs one captured hipGraph.

    python bench_infer.py [--iters 200]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--nc", type=int, default=1)
    a = ap.parse_args()
    import yolo_from_scratch_amd as y
    torch.manual_seed(0)
    m = y.YOLO(num_classes=a.nc, img_size=640)
    m.initialize_detection_biases(prior=0.2)          # untrained net: spread objectness so that NMS has work
    with torch.no_grad():
        for hd in (m.head_p3, m.head_p4, m.head_p5):
            hd[-1].weight.mul_(60.0)
    m = m.cuda().train()
    with torch.no_grad():                              # warm the BN running statistics (untrained net)
        for _ in range(3):
            m(torch.rand(4, 3, 640, 640, device="cuda"))
    m.eval()
    img = torch.rand(1, 3, 640, 640).pin_memory()
    with torch.no_grad():
        obj = torch.cat([torch.sigmoid(p[..., 4]).flatten() for p in m(img.cuda())])
    thr = float(torch.sort(obj, descending=True).values[3000])   # ~3000 candidates into NMS
    out = {}
    for name, use_graph in (("eager", False), ("hipgraph", True)):
        ses = y.InferenceSession(m, conf_threshold=thr, iou_threshold=0.4, use_graph=use_graph)
        for _ in range(10):
            dets = ses.run(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            dets = ses.run(img)                        # includes the D2H fetch (host sync) like predict()
        dt = (time.perf_counter() - t0) / a.iters
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ses.run(img, fetch=False)
        e1.record()
        torch.cuda.synchronize()
        out[name] = {"end_to_end_ms": round(dt * 1e3, 3), "device_ms": round(e0.elapsed_time(e1) / a.iters, 3),
                     "candidates": int(ses.det.count.item()), "kept": len(dets)}
    print(json.dumps({"metric": "bs=1 640x640 inference latency incl. global NMS", "unit": "ms", "dtype": "f32",
                      "config": {"workload": f"nc={a.nc} 640x640 bs=1, BN folded, 1 MI355X"}, **out}))


if __name__ == "__main__":
    main()
