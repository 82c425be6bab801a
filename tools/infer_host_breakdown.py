#!/usr/bin/env python3
"""Where the host time of one bs=1 InferenceSession.run goes (BASELINE config 5): per-phase wall clock over N images."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import yolo_from_scratch_amd as y
    from bench import detecting_model
    m, thr = detecting_model(y, 1)
    ses = y.InferenceSession(m, conf_threshold=thr, iou_threshold=0.4, use_graph=True)
    img = torch.rand(1, 3, 640, 640).pin_memory()
    for _ in range(20):
        ses.run(img)
    N = 300
    acc = dict(check=0.0, h2d=0.0, replay=0.0, sync=0.0, read=0.0)
    g = ses.graphs["f32"]
    st = torch.cuda.current_stream()
    for _ in range(N):
        t0 = time.perf_counter(); ses._check_state()
        t1 = time.perf_counter(); ses.x.copy_(img, non_blocking=True)
        t2 = time.perf_counter(); g.replay()
        t3 = time.perf_counter(); st.synchronize()
        t4 = time.perf_counter(); d = ses.det.read()
        t5 = time.perf_counter()
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[k] += v
    print(json.dumps({k: round(1e3 * v / N, 4) for k, v in acc.items()} | {"kept": len(d), "unit": "ms per image"}))


if __name__ == "__main__":
    main()
