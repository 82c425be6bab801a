#!/usr/bin/env python3
"""HBM traffic of the convolution kernels of ONE training step from two rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM
section: FETCH_SIZE and WRITE_SIZE collected in separate passes, unit KiB, FETCH_SIZE x2 on gfx950 for streaming reads --
the x2 is checked against the nchw_to_nhwc launch of the same run, whose traffic is known exactly).

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    YH_OVERLAP=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_FETCH_SIZE -o c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
    YH_OVERLAP=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_WRITE_SIZE -o c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
    python tools/hbm_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/r01_hbm_traffic.json
"""
import csv
import glob
import json
import sys

CONV = ("gather_gemm_kernel", "wino_kernel", "pw_gemm_kernel", "pw_stream_kernel", "stem_conv_kernel")
WGRAD = ("wgrad_kernel", "wino_wgrad_kernel", "pw_wgrad_kernel")


def load(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def last_step(rows):
    """dispatches of the last step: from the last input-layout kernel to the end; forward part ends at the first loss kernel"""
    starts = [i for i, r in enumerate(rows) if "nchw_to_nhwc_kernel" in r["Kernel_Name"]]
    s = starts[-1]
    step = rows[s:]
    loss = next(i for i, r in enumerate(step) if "loss_" in r["Kernel_Name"])
    return step, loss


def main():
    fd, wd, out = sys.argv[1:4]
    res = {}
    for name, d, counter in (("fetch", fd, "FETCH_SIZE"), ("write", wd, "WRITE_SIZE")):
        step, loss = last_step(load(d, counter))
        kib = lambda sel: sum(float(r["Counter_Value"]) for r in sel) * 1024.0
        fwd = [r for r in step[:loss] if any(k in r["Kernel_Name"] for k in CONV)]
        bwd = [r for r in step[loss:] if any(k in r["Kernel_Name"] for k in CONV) and not any(k in r["Kernel_Name"] for k in WGRAD)]
        wg = [r for r in step[loss:] if any(k in r["Kernel_Name"] for k in WGRAD)]
        res[name] = {"fwd": kib(fwd), "dgrad": kib(bwd), "wgrad": kib(wg), "n_fwd": len(fwd), "n_dgrad": len(bwd), "n_wgrad": len(wg),
                     "calib": kib([step[0]]),
                     "by_kernel_fwd": {k: kib([r for r in fwd if k in r["Kernel_Name"]]) for k in CONV}}
    corr = 2.0
    total = lambda part: res["fetch"][part] * corr + res["write"][part]
    doc = {
        "kernels": f"forward convolutions of one bs=64 step: {res['fetch']['n_fwd']} launches "
                   "(gather_gemm_kernel + wino_kernel + pw_gemm_kernel + pw_stream_kernel + stem_conv_kernel)",
        "fetch_size_bytes_raw": res["fetch"]["fwd"], "fetch_correction": corr, "write_size_bytes": res["write"]["fwd"],
        "hbm_bytes_per_step": total("fwd"),
        "hbm_bytes_per_step_by_kernel": {k: res["fetch"]["by_kernel_fwd"][k] * corr + res["write"]["by_kernel_fwd"][k] for k in CONV},
        "algorithmic_bytes_per_step": 8414720000.0,
        "dgrad_hbm_bytes_per_step": total("dgrad"), "wgrad_hbm_bytes_per_step": total("wgrad"),
        "launches": {"fwd": res["fetch"]["n_fwd"], "dgrad": res["fetch"]["n_dgrad"], "wgrad": res["fetch"]["n_wgrad"]},
        "calibration": {"kernel": "nchw_to_nhwc (reads 314.6 MB NCHW with 4-byte loads, writes 419.4 MB NHWC4)",
                        "FETCH_SIZE_bytes_raw": res["fetch"]["calib"], "WRITE_SIZE_bytes": res["write"]["calib"]},
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes over `YH_OVERLAP=0 python3 bench.py --steps 2 "
                  "--warmup 1 --no-cpu-baseline --no-roofline`; counters summed over the forward convolution launches of the last step "
                  "(tools/hbm_traffic.py); FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B/lane streaming reads, "
                  "MI355X_MICROARCH.md section HBM); counter unit KiB",
    }
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
