#!/bin/bash
# usage (GPU box, repo root): tools/pmc_bf16_gemm.sh [fwd|wgrad]  -- PMC passes over one 3x3 and one 1x1 bf16 forward layer (tools/one_conv_bf16.py);
# counters only (no trace domains), one group per pass; results as csv under gpurun_out/pmc_bf16_<layer>_<pass>/
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P3="SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P4="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
P5="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
WHAT=${1:-fwd}
for layer in 64,80,80,64,64,3,1 64,80,80,64,128,1,1; do
  tag=${WHAT}_$(echo $layer | tr ',' '_')
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/pmc_bf16_${tag}_p$i -o c -- python3 $R/tools/one_conv_bf16.py --shape $layer --what $WHAT --iters 3 > $R/gpurun_out/pmc_bf16_${tag}_p$i.log 2>&1 || echo "pass $i of $tag failed"
  done
done
echo pmc-done
