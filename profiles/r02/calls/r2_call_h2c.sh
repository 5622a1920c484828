#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r2_wino_h2_dbg.log
: > $L
for w in 0 1; do for d in 0 1 2 4 3 6 7; do
  echo "== [WIDE=$w PFA=2 DBG=$d]" >> $L
  AGZ_WINO_H2_WIDE=$w AGZ_WINO_H2_PFA=2 AGZ_WINO_H2_DBG=$d timeout 60 python scripts/nn_bench.py --wino-h2 --iters 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gemm_ms', round(d['wino']['gemm_ms_avg'],4))" >> $L
done; done
cat $L
