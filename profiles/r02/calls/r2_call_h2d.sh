#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r2_wino_h2_d.log
: > $L
echo "== parity" >> $L
AGZ_WINO_H2_PFA=2 timeout 300 python -m pytest tests/test_wino_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -4 >> $L
for env in "AGZ_WINO_H2_PFA=2" "AGZ_WINO_H2_WIDE=1 AGZ_WINO_H2_PFA=2"; do
  echo "== nn_bench --wino-h2 [$env]" >> $L
  env $env timeout 60 python scripts/nn_bench.py --wino-h2 --iters 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_pass'],3), round(d['conv_ms_avg'],4), d['wino'])" >> $L
done
cat $L
