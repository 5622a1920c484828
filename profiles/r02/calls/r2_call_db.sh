#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r2_gemm_db.log
: > $L
echo "== parity DB" >> $L
AGZ_WINO_H2_DB=1 AGZ_WINO_H2_WIDE=0 timeout 300 python -m pytest tests/test_wino_gpu.py -q -m gpu --tb=short -x -k "256 and WINO_H2 or 5-" 2>&1 | tail -3 >> $L
for env in "AGZ_WINO_H2_WIDE=1" "AGZ_WINO_H2_WIDE=0" "AGZ_WINO_H2_WIDE=0 AGZ_WINO_H2_DB=1 AGZ_WINO_H2_PFA=2" "AGZ_WINO_H2_WIDE=0 AGZ_WINO_H2_DB=1 AGZ_WINO_H2_PFA=3" "AGZ_WINO_H2_WIDE=0 AGZ_WINO_H2_DB=1 AGZ_WINO_H2_PFA=2 AGZ_WINO_H2_DBG=7" "AGZ_WINO_H2_WIDE=0 AGZ_WINO_H2_DB=1 AGZ_WINO_H2_PFA=2 AGZ_WINO_H2_STREAMS=2"; do
  echo "== nn_bench --wino-h2 [$env]" >> $L
  env $env timeout 60 python scripts/nn_bench.py --wino-h2 --iters 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_pass'],3), round(d['conv_ms_avg'],4), round(d['wino']['gemm_ms_avg'],4))" >> $L
done
cat $L
