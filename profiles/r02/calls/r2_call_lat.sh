#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r2_latency_modes.log
: > $L
for B in 1 8 16 32; do
  echo "== B=$B L=40 f32 (latency regime)" >> $L
  timeout 60 python scripts/nn_bench.py --B $B --L 40 --iters 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_pass'],4), 'conv', round(d['conv_ms_avg'],4), 'heads', round(d['heads_ms_avg'],4))" >> $L
  for w in 0 1; do
  echo "== B=$B L=40 wino_h2 forced WIDE=$w" >> $L
  AGZ_WINO_H2_WIDE=$w timeout 60 python scripts/nn_bench.py --B $B --L 40 --iters 20 --wino-h2 --force --no-latency 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_pass'],4), 'conv', round(d['conv_ms_avg'],4), 'heads', round(d['heads_ms_avg'],4), d['wino'])" >> $L
  done
done
cat $L
