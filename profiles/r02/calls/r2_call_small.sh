#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r2_init_heads.log
: > $L
for env in "AGZ_INIT_X3=0" "AGZ_INIT_X3=1" "AGZ_INIT_X3=1 AGZ_HEADS_SPREAD_MAX=512"; do
  echo "== nn_bench --wino-h2 [$env]" >> $L
  env $env timeout 60 python scripts/nn_bench.py --wino-h2 --iters 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_pass'],3), 'init', round(d['init_ms_avg'],4), 'heads', round(d['heads_ms_avg'],4))" >> $L
done
echo "== parity" >> $L
timeout 600 python -m pytest tests/test_headline_parity_gpu.py tests/test_net_gpu.py tests/test_wino_gpu.py -q -m gpu --tb=short -x -k "not one_tree and not engine" 2>&1 | tail -3 >> $L
cat $L
