#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r2_outdbg.log
: > $L
for d in 0 1 2 3 0; do
  echo "== out dbg=$d" >> $L
  AGZ_WINO_H2_OUT_DBG=$d timeout 60 python scripts/nn_bench.py --wino-h2 --iters 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); w=d['wino']; print(round(d['ms_per_pass'],3), round(d['conv_ms_avg'],4), 'in', round(w['in_ms_avg'],4), 'gemm', round(w['gemm_ms_avg'],4), 'out', round(w['out_ms_avg'],4))" >> $L
done
cat $L
