#!/bin/bash
# usage (GPU box, repo root): tools/pmc_step.sh <tag> [env assignments for bench.py ...]  -- one rocprofv3 --pmc pass (SQ counters only,
# no trace domains) over bench.py with the lanes serialised; per-kernel sums by tools/pmc_step.py
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
export YH_OVERLAP=0
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmcstep_${tag} -o c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/pmcstep_${tag}.log 2>&1
echo "pmc-step $tag"
