#!/bin/bash
# Winograd fp16x2: parity first, then timings of both GEMM tiles against the bf16x3 Winograd block
mkdir -p gpurun_out
L=gpurun_out/r2_wino_h2.log
: > $L
echo "== parity" >> $L
timeout 300 python -m pytest tests/test_wino_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -15 >> $L
timeout 300 python -m pytest "tests/test_headline_parity_gpu.py" -q -m gpu --tb=short -k "wino_h2" -s 2>&1 | tail -15 >> $L
for env in "" "AGZ_WINO_H2_WIDE=1"; do
  echo "== nn_bench --wino-h2 [$env]" >> $L
  env $env timeout 60 python scripts/nn_bench.py --wino-h2 2>/dev/null >> $L
done
echo "== nn_bench --wino" >> $L
timeout 60 python scripts/nn_bench.py --wino 2>/dev/null >> $L
cut -c1-700 $L
