#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output of ROCm 7.2): calls, total and average
duration per kernel name over the whole process, written as CSV.  With --per-step N the totals are divided by N steps.

    python tools/kernel_stats.py gpurun_out/prof/x_results.db profiles/r02_x_kernel_stats.csv [--per-step 13]
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def main():
    db, out = sys.argv[1:3]
    per = float(sys.argv[sys.argv.index("--per-step") + 1]) if "--per-step" in sys.argv else 1.0
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
                     "order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,percent" + (",ms_per_step\n" if per != 1.0 else "\n"))
        for n, k, s, a, lo, hi in rows:
            f.write(f"\"{short(n)}\",{k},{s / 1e6:.3f},{a / 1e3:.2f},{lo / 1e3:.2f},{hi / 1e3:.2f},{100 * s / tot:.2f}"
                    + (f",{s / 1e6 / per:.3f}\n" if per != 1.0 else "\n"))
    for n, k, s, a, lo, hi in rows[:25]:
        print(f"{short(n)[:70]:70s} calls {k:6d} total {s / 1e6:9.3f} ms  avg {a / 1e3:9.2f} us  {100 * s / tot:5.1f}%" +
              (f"  {s / 1e6 / per:7.3f} ms/step" if per != 1.0 else ""))


if __name__ == "__main__":
    main()
