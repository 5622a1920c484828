#!/bin/bash
# A/B: board-range reduction fused into the next input transform, then parity
mkdir -p gpurun_out
L=gpurun_out/r2_wino_tm.log
: > $L
run() {
  echo "== fuse-max=$1" >> $L
  AGZ_WINO_H2_FUSE_MAX=$1 timeout 60 python scripts/nn_bench.py --wino-h2 --iters 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); w=d['wino']; print(round(d['ms_per_pass'],3), round(d['conv_ms_avg'],4), 'in', round(w['in_ms_avg'],4), 'gemm', round(w['gemm_ms_avg'],4), 'out', round(w['out_ms_avg'],4))" >> $L
}
run 0; run 1; run 0; run 1
echo "== parity" >> $L
timeout 400 python -m pytest tests/test_wino_gpu.py -q -m gpu --tb=short -k "WINO_H2 or 5- or h2" 2>&1 | tail -3 >> $L
timeout 300 python -m pytest tests/test_headline_parity_gpu.py -q -m gpu --tb=short -k "wino_h2" -s 2>&1 | grep -E "parity|passed|failed" >> $L
timeout 300 python -m pytest tests/test_fullsize_gpu.py -q -m gpu --tb=short -k "split_mode or batch_independence" 2>&1 | tail -2 >> $L
cat $L
