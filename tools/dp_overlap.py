#!/usr/bin/env python3
"""Evidence that a gradient bucket's reduction runs UNDER the remaining backward kernels, from a rocprofv3
kernel + memory-copy trace of one rank of the 2-rank rehearsal (both ranks on the one GPU of the box, gloo: the exchange
is device->host copy, host reduce, host->device copy, issued per bucket by HipTrainer while yh_run keeps launching the next
backward segment; on a multi-GPU node the same call sites go to RCCL).

    cd /tmp && export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2
    (RANK=1 LOCAL_RANK=1 python3 $R/tests/dp_gpu_worker.py /tmp/dp 6 f32 big &) ; \\
    RANK=0 LOCAL_RANK=0 rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/dp_trace -o r0 -- python3 $R/tests/dp_gpu_worker.py /tmp/dp 6 f32 big
    python tools/dp_overlap.py gpurun_out/dp_trace/r0_results.db profiles/r02_dp2_overlap.json
"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from provenance import stamp


def main():
    db, outp = sys.argv[1:3]
    c = sqlite3.connect(db)
    allk = c.execute("select name,start,end,stream_id,grid_x from kernels order by start").fetchall()
    kern = [(k[0], k[1], k[2]) for k in allk]
    main_stream = max(set(k[3] for k in allk), key=lambda sid: sum(1 for k in allk if k[3] == sid))
    cols = [r[1] for r in c.execute("pragma table_info('memory_copies')")]
    size_col = "size" if "size" in cols else next(x for x in cols if "size" in x or "bytes" in x)
    name_col = "name" if "name" in cols else cols[0]
    copies = c.execute(f"select {name_col},start,end,{size_col} from memory_copies order by start").fetchall()
    big = [m for m in copies if m[3] and m[3] >= (1 << 20) and False]   # (input uploads: not part of the exchange)
    # gloo moves a CUDA bucket with blit kernels on its own streams: large __amd_rocclr_copyBuffer launches off the main stream
    big += [("blit " + k[0].split("(")[0], k[1], k[2], 4 * k[4]) for k in allk if "copyBuffer" in k[0] and k[3] != main_stream and k[4] >= 65536]
    big.sort(key=lambda m: m[1])
    bwd = [(k[0], k[1], k[2]) for k in allk if k[3] == main_stream and
           any(t in k[0] for t in ("wgrad", "bwd", "gather_gemm", "wino_kernel", "pw_gemm", "colsum", "bf16_gemm"))]
    rows, tot, cov = [], 0, 0
    for name, s, e, size in big:
        over = 0
        names = set()
        for kn, ks, ke in bwd:
            if ke <= s:
                continue
            if ks >= e:
                break
            over += min(e, ke) - max(s, ks)
            names.add(kn.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "").split("<")[0])
        over = min(over, e - s)
        tot += e - s
        cov += over
        rows.append({"copy": str(name), "bytes": int(size), "start_us": round((s - big[0][1]) / 1e3, 1), "dur_us": round((e - s) / 1e3, 1),
                     "overlapped_by_backward_kernels_us": round(over / 1e3, 1), "kernels": sorted(names)[:6]})
    # where inside each step's backward the bucket copies sit: position as a fraction of [loss kernel, grad-norm kernel]
    losses = [k for k in allk if "loss_main" in k[0]]
    norms = [k for k in allk if "sqnorm_stage1" in k[0]]
    steps = []
    for lk in losses:
        nk = next((n for n in norms if n[1] > lk[1]), None)
        if nk is None:
            continue
        span = nk[1] - lk[2]
        cs = [m for m in big if lk[2] <= m[1] and m[2] <= nk[1]]
        if not cs or span <= 0:
            continue
        steps.append({"backward_span_us": round(span / 1e3, 1), "bucket_copies": len(cs),
                      "copy_windows_as_fraction_of_backward": [[round((m[1] - lk[2]) / span, 3), round((m[2] - lk[2]) / span, 3)] for m in cs],
                      "exposed_after_last_copy_us": round((nk[1] - cs[-1][2]) / 1e3, 1)})
    doc = {"stamp": stamp(), "rank": 0, "world": 2, "steps": steps[1:], "backend": "gloo (2 ranks share the one GPU of the box)",
           "bucket_copies": len(rows), "copy_time_us": round(tot / 1e3, 1), "copy_time_under_backward_kernels_us": round(cov / 1e3, 1),
           "fraction_overlapped": round(cov / tot, 3) if tot else None, "copies": rows[:48],
           "method": "rocprofv3 --kernel-trace --memory-copy-trace on rank 0 of tests/dp_gpu_worker.py (world 2); a copy >= 1 MiB is a gradient "
                     "bucket leaving / re-entering the device; overlap = time of that copy during which a backward kernel of the same process ran"}
    json.dump(doc, open(outp, "w"), indent=1)
    print(json.dumps({k: v for k, v in doc.items() if k != "copies"}, indent=1))
    for r in rows[:12]:
        print(r)


if __name__ == "__main__":
    main()
