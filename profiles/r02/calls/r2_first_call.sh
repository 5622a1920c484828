#!/bin/bash
# First GPU call of round 2 on this branch: parity of every env-gated variant, then timings.  ~60 s on the GPU box.
# usage: gpurun --timeout 200 -- scripts/r2_first_call.sh
mkdir -p gpurun_out
L=gpurun_out/r2_first_call.log
: > $L
for env in "" "AGZ_WINO_STREAM=3" "AGZ_WINO_FUSE=1" "AGZ_WINO_STREAM=3 AGZ_WINO_FUSE=1"; do
  echo "== parity [$env]" >> $L
  env $env timeout 60 python -m pytest tests/test_wino_gpu.py -q -m gpu --tb=line 2>&1 | tail -4 >> $L
done
for env in "" "AGZ_WINO_STREAM=3" "AGZ_WINO_STREAM=2" "AGZ_WINO_FUSE=1" "AGZ_WINO_STREAM=3 AGZ_WINO_FUSE=1"; do
  echo "== nn_bench --wino [$env]" >> $L
  env $env timeout 40 python scripts/nn_bench.py --wino 2>/dev/null >> $L
done
echo "== latency, 16 lanes: f32 split-K vs Winograd blocks" >> $L
timeout 60 python scripts/latency_bench.py --lanes 16 --moves 4 2>/dev/null >> $L
AGZ_WINO_LATENCY_TILES=200 timeout 60 python scripts/latency_bench.py --lanes 16 --moves 4 --compute wino 2>/dev/null >> $L
cut -c1-400 $L
