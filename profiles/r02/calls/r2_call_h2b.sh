#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r2_wino_h2_pfa.log
: > $L
for env in "AGZ_WINO_H2_PFA=3" "AGZ_WINO_H2_WIDE=1 AGZ_WINO_H2_PFA=3"; do
  echo "== parity [$env]" >> $L
  env $env timeout 300 python -m pytest tests/test_wino_gpu.py -q -m gpu --tb=short -x -k "h2 or WINO_H2 or 5" 2>&1 | tail -4 >> $L
done
for w in 0 1; do for p in 1 2 3 4; do
  echo "== nn_bench --wino-h2 [WIDE=$w PFA=$p]" >> $L
  AGZ_WINO_H2_WIDE=$w AGZ_WINO_H2_PFA=$p timeout 60 python scripts/nn_bench.py --wino-h2 --iters 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_pass'],3), round(d['conv_ms_avg'],4), d['wino'])" >> $L
done; done
cat $L
