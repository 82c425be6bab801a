#!/usr/bin/env python3
"""Per-step forward-convolution time by kernel from a rocprofv3 kernel trace of bench.py (the cross-check for
`roofline.*_ms_per_step` in the bench JSON, which are HIP-event times taken in the same process).

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    YH_OVERLAP=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace0 -o t -- python3 bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 2
    python tools/trace_summary.py gpurun_out/trace0 profiles/r01_forward_conv_trace.json
"""
import csv
import glob
import json
import sys

CONV = ("gather_gemm_kernel", "wino_kernel", "pw_gemm_kernel", "pw_stream_kernel", "stem_conv_kernel")


def main():
    d, out = sys.argv[1:3]
    f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "nchw_to_nhwc_kernel" in r["Kernel_Name"]]
    steps = []
    for a, b in zip(starts[:-1], starts[1:]):
        step = rows[a:b]
        loss = next((i for i, r in enumerate(step) if "loss_" in r["Kernel_Name"]), None)
        if loss is None:
            continue
        fwd = step[:loss]
        per = {}
        for r in fwd:
            for k in CONV:
                if k in r["Kernel_Name"]:
                    e = per.setdefault(k, [0, 0.0])
                    e[0] += 1
                    e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        span = (int(fwd[-1]["End_Timestamp"]) - int(fwd[0]["Start_Timestamp"])) / 1e6
        steps.append((per, span, (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6))
    steps = steps[1:]                              # drop the first (cold) step
    n = len(steps)
    doc = {"source": "rocprofv3 --kernel-trace of `YH_OVERLAP=0 python3 bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 2`",
           "steps_averaged": n, "forward_conv_kernels": {}}
    for k in CONV:
        cnt = sum(s[0].get(k, [0, 0])[0] for s in steps) / n
        ms = sum(s[0].get(k, [0, 0])[1] for s in steps) / n
        doc["forward_conv_kernels"][k] = {"launches_per_step": cnt, "ms_per_step": round(ms, 3), "avg_us": round(1e3 * ms / cnt, 1) if cnt else None}
    doc["forward_conv_ms_per_step"] = round(sum(v["ms_per_step"] for v in doc["forward_conv_kernels"].values()), 3)
    doc["forward_pass_span_ms"] = round(sum(s[1] for s in steps) / n, 3)
    doc["step_span_ms_serial"] = round(sum(s[2] for s in steps) / n, 3)
    doc["algorithmic_tflops_forward_convs"] = round(467.01 / doc["forward_conv_ms_per_step"], 2)
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
