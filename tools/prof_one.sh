#!/bin/bash
# usage (GPU box, repo root): tools/prof_one.sh <tag> <python script + args ...>   -> gpurun_out/<tag>_stats.csv (per-kernel summary)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$tag -o p -- python3 "$@" > $R/gpurun_out/prof_$tag.log 2>&1
db=$(find $R/gpurun_out/prof_$tag -name '*_results.db' | head -1)
python3 $R/tools/kernel_stats.py "$db" $R/gpurun_out/${tag}_stats.csv > /dev/null 2>&1
rm -rf $R/gpurun_out/prof_$tag
cut -c1-160 $R/gpurun_out/${tag}_stats.csv | head -8
