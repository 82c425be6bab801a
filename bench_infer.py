#!/usr/bin/env python3
"""BASELINE config 5: bs=1 640x640 end-to-end inference latency (H2D image copy, NCHW->NHWC, BN-folded fused
forward, candidate extraction, global NMS, result table, D2H, python list), eager launches vs one captured hipGraph.
The same measurement rides in bench.py's line as the `infer` key; this is the standalone form.

    python bench_infer.py [--iters 200] [--nc 1]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--nc", type=int, default=1)
    a = ap.parse_args()
    import yolo_from_scratch_amd as y
    from bench import infer_latency
    out = infer_latency(y, a.nc, a.iters)
    print(json.dumps({"metric": "bs=1 640x640 inference latency incl. global NMS", "dtype": "f32", **out}))


if __name__ == "__main__":
    main()
