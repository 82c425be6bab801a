#!/usr/bin/env python3
"""Copy the files written by tools/refresh_evidence.sh (gpurun_out/ev_*.json) into profiles/r04_*: the bench lines as they are,
the model-size runs condensed into profiles/r04_sizes.json.  Every file gets the provenance stamp of the tree it ran from
(the bench lines carry none of their own)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from provenance import stamp

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")


def main():
    st = stamp()
    # the files in gpurun_out/ must have been measured on THIS tree (refresh_evidence.sh records the hash it ran on): stamping the
    # results of an earlier collection with the current hash would be a false provenance
    ran_on = open(os.path.join(G, "collected_hash.txt")).read().strip() if os.path.exists(os.path.join(G, "collected_hash.txt")) else None
    if ran_on != st["csrc_sha16"]:
        sys.exit(f"gpurun_out/ev_* were collected on csrc hash {ran_on}, the tree is {st['csrc_sha16']}: run tools/refresh_evidence.sh on the GPU box first")
    for name in ("bench_f32", "bench_f32_80_640_64", "bench_f32_80_1280_16", "bench_bf16_1_640_64", "bench_bf16_80_640_64",
                 "bench_bf16_80_1280_16", "infer_latency"):
        d = json.load(open(os.path.join(G, f"ev_{name}.json")))
        d["stamp"] = st
        json.dump(d, open(os.path.join(P, f"r04_{name}.json"), "w"), indent=1)
    runs = {}
    for sz in "nmlx":
        for dt in ("f32", "bf16"):
            d = json.load(open(os.path.join(G, f"ev_size_{sz}_{dt}.json")))
            r = d.get("roofline", {})
            runs[f"{sz}_{dt}"] = {"images_per_s": d["value"], "ms_per_step": d["ms_per_step"], "config": d["config"]["workload"],
                                  "global_batch": d["config"].get("global_batch"),
                                  "forward_conv": {k: r.get(k) for k in ("bound", "achieved", "unit", "frac", "kernel_ms_per_step",
                                                                         "algorithmic_gflop_per_step", "conv_tflops")}}
    json.dump({"note": "informational: the reference's other model sizes (train.py:1346-1352), nc=1 640x640, one training step; "
                       "l / x at batch 32; not the reported metric", "stamp": st, "runs": runs},
              open(os.path.join(P, "r04_sizes.json"), "w"), indent=1)
    print("stored", st)


if __name__ == "__main__":
    main()
