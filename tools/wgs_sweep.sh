#!/bin/bash
# split-count / tile sweep of the flat-stream weight gradient (diagnostic build: make EXTRA=-DYH_WGS_TUNE)
for shape in 64,40,40,64,64,3,1 64,80,80,32,32,3,1 64,80,80,64,64,3,1 64,40,40,128,128,3,1 64,20,20,256,256,3,1 64,20,20,128,128,3,1; do
  for nj in 1 2; do
    for sp in 16 32 64 128 256; do
      echo -n "nj=$nj splits=$sp  "
      YH_WGS_NJ=$nj YH_WGS_SPLITS=$sp ITERS=10 python tools/wgrad_bench.py $shape 2>&1 | grep -v amdgpu
    done
  done
done
