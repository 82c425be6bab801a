#!/bin/bash
# split-K target sweep of the bs=1 inference forward (diagnostic build: make EXTRA=-DYH_WGS_TUNE)
for tgt in ${TARGETS:-1 64 96 128 160 192}; do
  echo -n "target=$tgt  "
  YH_SK_TARGET=$tgt python bench_infer.py --iters 100 2>&1 | grep -o '"eager".*' | cut -c1-470
done
