#!/usr/bin/env python3
"""Kernel timeline of one bs=1 inference from a rocprofv3 --kernel-trace run of bench_infer.py (rocpd sqlite output).

    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_inf -o inf -- python3 $GRAFT_REPO_ROOT/bench_infer.py --iters 50
    python tools/infer_timeline.py gpurun_out/prof_inf/inf_results.db [out.json]
"""
import json
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name,start,end,grid_x,grid_y,workgroup_x from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "nchw_to_nhwc" in r[0]]
    a, b = idx[30], idx[31]                  # an eager iteration after the warm-ups
    t0 = rows[a][1]
    agg, lines = {}, []
    for r in rows[a:b]:
        n = re.sub(r"\(anonymous namespace\)::", "", r[0]).split("(")[0].replace("void ", "")
        lines.append(f"{(r[1] - t0) / 1e3:9.1f} us  dur {(r[2] - r[1]) / 1e3:7.1f}  grid {r[3] // max(r[5], 1):5d}x{r[4]}  {n[:70]}")
        key = n.split("<")[0]
        e = agg.setdefault(key, [0, 0.0])
        e[0] += 1
        e[1] += (r[2] - r[1]) / 1e3
    nms_end = max((r[2] for r in rows[a:b] if "nms_scan" in r[0]), default=rows[b - 1][2])
    doc = {"launches": b - a, "device_span_us_to_nms_end": round((nms_end - t0) / 1e3, 1),
           "sum_kernel_us": round(sum(v[1] for v in agg.values()), 1),
           "by_kernel": {k: {"launches": v[0], "us": round(v[1], 1)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}}
    print("\n".join(lines))
    print(json.dumps(doc, indent=1))
    if len(sys.argv) > 2:
        json.dump(doc, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
