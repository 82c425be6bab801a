#!/bin/bash
# usage (GPU box, repo root): tools/refresh_evidence.sh  -- the bench lines of every BASELINE shape (fp32 and bf16), the model
# sizes n/m/l/x and the bs=1 inference latency into gpurun_out/ev_*.json (copied into profiles/r04_* by tools/store_evidence.py)
set -e
O=gpurun_out
python3 bench.py > $O/ev_bench_f32.json 2> $O/ev_bench_f32.err
YH_BENCH_SHAPE=80,640,64 python3 bench.py --no-cpu-baseline > $O/ev_bench_f32_80_640_64.json 2> $O/ev_b2.err
YH_BENCH_SHAPE=80,1280,16 python3 bench.py --no-cpu-baseline > $O/ev_bench_f32_80_1280_16.json 2> $O/ev_b3.err
YH_BENCH_DTYPE=bf16 python3 bench.py --no-cpu-baseline > $O/ev_bench_bf16_1_640_64.json 2> $O/ev_b4.err
YH_BENCH_DTYPE=bf16 YH_BENCH_SHAPE=80,640,64 python3 bench.py --no-cpu-baseline > $O/ev_bench_bf16_80_640_64.json 2> $O/ev_b5.err
YH_BENCH_DTYPE=bf16 YH_BENCH_SHAPE=80,1280,16 python3 bench.py --no-cpu-baseline > $O/ev_bench_bf16_80_1280_16.json 2> $O/ev_b6.err
python3 bench_infer.py > $O/ev_infer_latency.json 2> $O/ev_infer.err
for sz in n m l x; do
  for dt in f32 bf16; do
    extra=""; if [ $sz = l ] || [ $sz = x ]; then extra="YH_BENCH_SHAPE=1,640,32"; fi
    env YH_BENCH_SIZE=$sz YH_BENCH_DTYPE=$dt $extra python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/ev_size_${sz}_${dt}.json 2> $O/ev_size_${sz}_${dt}.err
  done
done
python3 -c "import sys; sys.path.insert(0, 'tools'); from provenance import csrc_hash; print(csrc_hash())" > $O/collected_hash.txt
echo refreshed
