#!/bin/bash
# usage (GPU box, repo root): tools/final_collect.sh -- the stamped step profiles (fp32 + bf16), the serial kernel table and the
# headline bench line, all for the tree as it stands (run after the LAST change under csrc/: bench.py quotes the PMC traffic only
# when the profile's stamp equals the tree's hash).  Outputs land in gpurun_out/; copy them with tools/store_evidence.py.
set -e
bash tools/collect_step_profile.sh f32
python3 tools/step_profile.py gpurun_out/sp_f32_trace/t_results.db gpurun_out/sp_f32_fetch/c_results.db gpurun_out/sp_f32_write/c_results.db profiles/r04_step_profile_f32.json > gpurun_out/sp_f32.log 2>&1
python3 tools/kernel_stats.py gpurun_out/sp_f32_trace/t_results.db profiles/r04_kernel_stats_f32_serial.csv --per-step 5 > /dev/null 2>&1
bash tools/collect_step_profile.sh bf16 YH_BENCH_DTYPE=bf16 YH_BENCH_SHAPE=80,640,64
python3 tools/step_profile.py gpurun_out/sp_bf16_trace/t_results.db gpurun_out/sp_bf16_fetch/c_results.db gpurun_out/sp_bf16_write/c_results.db profiles/r04_step_profile_bf16.json > gpurun_out/sp_bf16.log 2>&1
cp profiles/r04_step_profile_f32.json profiles/r04_step_profile_bf16.json profiles/r04_kernel_stats_f32_serial.csv gpurun_out/
bash tools/pmc_step.sh f32
python3 tools/pmc_step.py gpurun_out/pmcstep_f32 profiles/r04_pmc_step_f32.json > gpurun_out/pmc_step_f32.log 2>&1
bash tools/pmc_step.sh bf16 YH_BENCH_DTYPE=bf16 YH_BENCH_SHAPE=80,640,64
python3 tools/pmc_step.py gpurun_out/pmcstep_bf16 profiles/r04_pmc_step_bf16.json > gpurun_out/pmc_step_bf16.log 2>&1
cp profiles/r04_pmc_step_f32.json profiles/r04_pmc_step_bf16.json gpurun_out/
rm -rf gpurun_out/sp_f32_trace gpurun_out/sp_f32_fetch gpurun_out/sp_f32_write gpurun_out/sp_bf16_trace gpurun_out/sp_bf16_fetch gpurun_out/sp_bf16_write gpurun_out/pmcstep_f32 gpurun_out/pmcstep_bf16
python3 bench.py > gpurun_out/ev_bench_f32.json 2> gpurun_out/ev_bench_f32.err
echo final2
