#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r2_chunks.log
: > $L
for env in "AGZ_WINO_H2_STREAMS=1" "AGZ_WINO_H2_STREAMS=2" "AGZ_WINO_H2_STREAMS=2 AGZ_WINO_H2_CHUNK=128" "AGZ_WINO_H2_STREAMS=2 AGZ_WINO_H2_CHUNK=86" "AGZ_WINO_H2_STREAMS=2 AGZ_WINO_H2_CHUNK=64" "AGZ_WINO_H2_STREAMS=1 AGZ_WINO_H2_CHUNK=86" "AGZ_WINO_H2_STREAMS=2 AGZ_WINO_H2_CHUNK=43"; do
  echo "== nn_bench --wino-h2 [$env]" >> $L
  env $env timeout 60 python scripts/nn_bench.py --wino-h2 --iters 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_pass'],3))" >> $L
done
echo "== parity chunks+streams" >> $L
AGZ_WINO_H2_STREAMS=2 AGZ_WINO_H2_CHUNK=86 timeout 300 python -m pytest tests/test_headline_parity_gpu.py -q -m gpu --tb=short -k "headline_network and wino_h2" 2>&1 | tail -3 >> $L
cat $L
