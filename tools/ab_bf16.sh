#!/bin/bash
# A/B of the flat-stream kernels on the bf16 nc=80 step: YH_BF16_STREAM x YH_OVERLAP, two runs each (ms per step)
for st in 0 1; do for ov in 0 1; do for r in 1 2; do
  echo -n "stream=$st overlap=$ov: "
  YH_BF16_STREAM=$st YH_OVERLAP=$ov YH_BENCH_DTYPE=bf16 YH_BENCH_SHAPE=80,640,64 python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done; done; done
