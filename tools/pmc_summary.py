#!/usr/bin/env python3
"""Summarise the csv counter files of tools/pmc_bf16_gemm.sh: mean per dispatch of every counter, per kernel name substring.
usage: python tools/pmc_summary.py <kernel substring> gpurun_out/pmc_bf16_<tag>_p*"""
import collections, csv, glob, sys
sub = sys.argv[1]
for d in sys.argv[2:]:
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    acc, n = collections.defaultdict(float), collections.defaultdict(int)
    for row in csv.DictReader(open(fs[0])):
        if sub in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    print(d.split("/")[-1], {k: round(v / n[k]) for k, v in acc.items()}, "dispatches", max(n.values()) if n else 0)
