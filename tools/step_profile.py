#!/usr/bin/env python3
"""Per-kernel time and HBM traffic of ONE training step, forward and backward, from three rocprofv3 runs of bench.py with the
lanes serialised (YH_OVERLAP=0): a kernel trace and two PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md, rocprofv3 PMC slots).  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests of wide streaming reads
at 64 bytes; checked here against the nchw_to_nhwc launch whose traffic is known), counter unit KiB.

    cd /tmp && export TMPDIR=/tmp
    YH_OVERLAP=0 rocprofv3 --kernel-trace   -d $R/gpurun_out/sp_trace -o t -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extras
    YH_OVERLAP=0 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/sp_fetch -o c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras
    YH_OVERLAP=0 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/sp_write -o c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras
    python tools/step_profile.py gpurun_out/sp_trace/t_results.db gpurun_out/sp_fetch/c_results.db gpurun_out/sp_write/c_results.db profiles/r04_step_profile_f32.json
(the same with YH_BENCH_DTYPE=bf16 YH_BENCH_SHAPE=80,640,64 for the bf16 path)
"""
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from provenance import stamp

FWD_CONV = ("gather_gemm_kernel", "wino_kernel", "wino_lds_kernel", "pw_gemm_kernel", "pw_tile_kernel", "pw_stream_kernel", "bf16_gemm_kernel",
            "bf16_fstream_kernel", "narrow_conv_kernel", "narrow_first_bf16_kernel", "narrow_s2_16x32_bf16_kernel", "s2_lds_kernel", "lat_conv_kernel")
DGRAD_ONLY = ("narrow_dgrad_s2_kernel", "narrow_dgrad_s2_bf16_kernel")
WGRAD = ("wgrad_kernel", "wino_wgrad_kernel", "wino_wgrad_lds_kernel", "pw_wgrad_kernel", "bf16_wgrad_kernel", "bf16_wgrad_stream_kernel", "wgrad_reduce")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    name = name.split("(")[0].split("<")[0]
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", name)          # mangled template instantiations of elementwise kernels
    if m:
        rest = name[m.end():]
        return rest[: int(m.group(1))]
    return name


def family(name, phase):
    n = short(name)
    if any(k in n for k in WGRAD):
        return "wgrad:" + n
    if n in DGRAD_ONLY:
        return "dgrad:" + n
    if n in FWD_CONV or any(n.startswith(k) for k in FWD_CONV):
        return ("fwd_conv:" if phase == "fwd" else "dgrad:") + n
    return "other:" + n


def steps_of(rows, key_name, key_start):
    """split dispatches (sorted by start) into steps at the input-layout kernel; returns list of (fwd rows, bwd rows)"""
    starts = [i for i, r in enumerate(rows) if "nchw_to_nhwc_kernel" in r[key_name] and not any("loss_" in q[key_name] for q in rows[max(0, i - 3):i])]
    out = []
    for a, b in zip(starts, starts[1:] + [len(rows)]):
        step = rows[a:b]
        loss = next((i for i, r in enumerate(step) if "loss_" in r[key_name]), None)
        if loss is None:
            continue
        end = next((i for i, r in enumerate(step) if "adam" in r[key_name]), len(step) - 1)
        out.append((step[:loss], step[loss:end + 1]))
    return out


def main():
    tdb, fdb, wdb, outp = sys.argv[1:5]
    c = sqlite3.connect(tdb)
    rows = [dict(name=r[0], start=r[1], dur=r[2]) for r in c.execute("select name,start,duration from kernels order by start")]
    st = steps_of(rows, "name", "start")[1:]              # drop the first (cold) step
    n = len(st)
    fam = {}
    for fwd, bwd in st:
        for phase, part in (("fwd", fwd), ("bwd", bwd)):
            for r in part:
                e = fam.setdefault(family(r["name"], phase), {"launches": 0, "ms": 0.0})
                e["launches"] += 1
                e["ms"] += r["dur"] / 1e6
    for e in fam.values():
        e["launches_per_step"] = round(e.pop("launches") / n, 2)
        e["ms_per_step"] = round(e.pop("ms") / n, 4)
    traffic = {}
    calib = {}
    for label, db, counter in (("fetch", fdb, "FETCH_SIZE"), ("write", wdb, "WRITE_SIZE")):
        cc = sqlite3.connect(db)
        pr = [dict(name=r[0], start=r[1], val=r[2]) for r in
              cc.execute("select name,start,counter_value from pmc_events where counter_name=? order by start", (counter,))]
        fwd, bwd = steps_of(pr, "name", "start")[-1]
        calib[label] = fwd[0]["val"] * 1024.0
        for phase, part in (("fwd", fwd), ("bwd", bwd)):
            for r in part:
                traffic.setdefault(family(r["name"], phase), {"fetch": 0.0, "write": 0.0})[label] += r["val"] * 1024.0
    for k, e in fam.items():
        t = traffic.get(k)
        if t:
            e["hbm_bytes_per_step"] = round(2.0 * t["fetch"] + t["write"])
            e["hbm_GBps"] = round(e["hbm_bytes_per_step"] / (e["ms_per_step"] * 1e-3) / 1e9, 1) if e["ms_per_step"] > 0 else None
    grp = {}
    for k, e in fam.items():
        g = grp.setdefault(k.split(":")[0], {"ms_per_step": 0.0, "hbm_bytes_per_step": 0})
        g["ms_per_step"] = round(g["ms_per_step"] + e["ms_per_step"], 4)
        g["hbm_bytes_per_step"] += e.get("hbm_bytes_per_step", 0)
    doc = {"stamp": stamp(), "steps_averaged": n, "groups": grp,
           "kernels": dict(sorted(fam.items(), key=lambda kv: -kv[1]["ms_per_step"])),
           "calibration": {"kernel": "nchw_to_nhwc (first launch of the step: reads the NCHW fp32 batch with 4-byte loads)",
                           "FETCH_SIZE_bytes_raw": calib.get("fetch"), "WRITE_SIZE_bytes": calib.get("write")},
           "method": "rocprofv3 --kernel-trace (time, mean over steps) and --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (last step) "
                     "over `YH_OVERLAP=0 python3 bench.py --no-cpu-baseline --no-roofline --no-extras`; HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE "
                     "(gfx950 reports half the bytes of 16-B/lane streaming reads, MI355X_MICROARCH.md section HBM); tools/step_profile.py"}
    json.dump(doc, open(outp, "w"), indent=1)
    print(json.dumps({"groups": grp, "top": dict(list(doc["kernels"].items())[:12])}, indent=1))


if __name__ == "__main__":
    main()
