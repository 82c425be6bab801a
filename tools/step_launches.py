#!/usr/bin/env python3
"""Every launch of ONE training step (name, grid, microseconds) from a rocprofv3 --kernel-trace run of bench.py with the
lanes serialised (YH_OVERLAP=0), in launch order; optional substring filter.

    python tools/step_launches.py <results.db> [filter]
"""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = c.execute("select name,start,end,grid_x,grid_y,workgroup_x from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "nchw_to_nhwc" in r[0]]
    a, b = idx[-2], idx[-1]
    t0 = rows[a][1]
    tot = 0.0
    for r in rows[a:b]:
        n = re.sub(r"\(anonymous namespace\)::", "", r[0]).split("(")[0].replace("void ", "")
        if flt and flt not in n:
            continue
        tot += (r[2] - r[1]) / 1e3
        print(f"{(r[1] - t0) / 1e3:9.1f} us  dur {(r[2] - r[1]) / 1e3:7.1f}  grid {r[3] // max(r[5], 1):6d}x{r[4]:<3d} {n[:90]}")
    print(f"total {tot:.1f} us")


if __name__ == "__main__":
    main()
