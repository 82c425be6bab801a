#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_step_profile.sh <tag> [env assignments for bench.py ...]
# three rocprofv3 runs of bench.py with the lanes serialised: kernel trace, FETCH_SIZE pass, WRITE_SIZE pass (separate passes:
# the TCC block has 4 counter slots, FETCH_SIZE takes 3 and WRITE_SIZE 2)
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
export YH_OVERLAP=0
rocprofv3 --kernel-trace   -d $R/gpurun_out/sp_${tag}_trace -o t -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/sp_${tag}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/sp_${tag}_fetch -o c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/sp_${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/sp_${tag}_write -o c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/sp_${tag}_write.log 2>&1
echo "collected $tag"
