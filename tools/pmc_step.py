#!/usr/bin/env python3
"""Per-kernel SQ counter table of one training step from tools/pmc_step.sh.
usage: python tools/pmc_step.py gpurun_out/pmcstep_<tag> profiles/r04_pmc_step_<tag>.json
Columns: launches, MFMA-pipe busy share of the SIMD cycles the kernel was resident (SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs x
8 CUs per SE-record ... reported as busy cycles per launch and as a share of GRBM-equivalent time via SQ_BUSY_CYCLES), vector instructions
per wave-cycle, and the share of wave time spent issuing VALU / waiting on an instruction."""
import collections, csv, glob, json, re, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from provenance import stamp

src, out = sys.argv[1], sys.argv[2]
f = glob.glob(src + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for row in csv.DictReader(open(f)):
    name = re.sub(r"^void ", "", row["Kernel_Name"])
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
    disp[name].add(row["Dispatch_Id"])
rows = []
for name, c in acc.items():
    n = len(disp[name])
    wave = c.get("SQ_WAVE_CYCLES", 0.0) * 4.0            # wave-cycles
    busy = c.get("SQ_BUSY_CYCLES", 0.0)                  # summed over the 32 shader engines
    simd_cycles = busy / 32.0 * 1024.0                   # SIMD-cycles available while the kernel was resident
    rows.append({"kernel": name, "launches": n,
                 "mfma_busy_frac": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles, 4) if simd_cycles else None,
                 "waves_per_simd": round(wave / simd_cycles, 2) if simd_cycles else None,
                 "valu_issue_frac_of_wave_time": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / wave, 3) if wave else None,
                 "wait_inst_frac_of_wave_time": round(c.get("SQ_WAIT_INST_ANY", 0.0) * 4.0 / wave, 3) if wave else None,
                 "valu_insts": int(c.get("SQ_INSTS_VALU", 0.0)), "mfma_busy_cycles": int(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)),
                 "busy_cycles_per_launch_per_se": round(busy / 32.0 / n)})
rows.sort(key=lambda r: -r["busy_cycles_per_launch_per_se"] * r["launches"])
json.dump({"stamp": stamp(), "method": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES "
           "SQ_WAIT_INST_ANY over `YH_OVERLAP=0 python3 bench.py --steps 2 --warmup 1` (3 steps in the file); sums over all launches of a kernel; "
           "SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are in units of 4 cycles, SQ_BUSY_CYCLES is summed over 32 shader engines, "
           "SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs", "kernels": rows}, open(out, "w"), indent=1)
for r in rows[:25]:
    print(f"{r['kernel'][:60]:60s} n={r['launches']:4d} mfma_busy {r['mfma_busy_frac']}  waves/simd {r['waves_per_simd']}  valu {r['valu_issue_frac_of_wave_time']}  wait {r['wait_inst_frac_of_wave_time']}")
