#!/bin/bash
# A/B: output transform form 3 with the parameter loads issued late (<= 128 registers, 4 waves per SIMD) against the default
mkdir -p gpurun_out
L=gpurun_out/r2_wino_tm.log
: > $L
run() {  # late
  echo "== TM=5 out-form=3 late-E=$1" >> $L
  AGZ_WINO_H2_OUT_LATE_E=$1 timeout 60 python scripts/nn_bench.py --wino-h2 --iters 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); w=d['wino']; print(round(d['ms_per_pass'],3), round(d['conv_ms_avg'],4), 'in', round(w['in_ms_avg'],4), 'gemm', round(w['gemm_ms_avg'],4), 'out', round(w['out_ms_avg'],4))" >> $L
}
run 0; run 1; run 0; run 1
echo "== parity late-E=1" >> $L
AGZ_WINO_H2_OUT_LATE_E=1 timeout 300 python -m pytest tests/test_headline_parity_gpu.py -q -m gpu --tb=short -k "headline_network and wino_h2" -s 2>&1 | grep -E "parity|passed|failed" >> $L
cat $L
