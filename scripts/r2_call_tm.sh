#!/bin/bash
# EXPERIMENT: would contiguous A-operand runs speed the GEMM up?  (fake addresses, timing only) + the new trainer test
mkdir -p gpurun_out
L=gpurun_out/r2_wino_tm.log
: > $L
run() {
  echo "== fake-a=$1" >> $L
  AGZ_WINO_H2_FAKEA=$1 timeout 60 python scripts/nn_bench.py --wino-h2 --iters 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); w=d['wino']; print(round(d['ms_per_pass'],3), round(d['conv_ms_avg'],4), 'in', round(w['in_ms_avg'],4), 'gemm', round(w['gemm_ms_avg'],4), 'out', round(w['out_ms_avg'],4))" >> $L
}
run 0; run 1; run 0; run 1
timeout 300 python -m pytest tests/test_train_gpu.py -q -m gpu --tb=short -k "headline_width" -s 2>&1 | grep -E "trainer|passed|failed" >> $L
cat $L
