#!/bin/bash
# A/B of the WINO_H2 tuning knobs on the headline block (NN-only bench, HIP events per kernel class), then parity at the defaults.
# usage (on the GPU box): bash scripts/r2_call_tm.sh        -> gpurun_out/r2_wino_tm.log
mkdir -p gpurun_out
L=gpurun_out/r2_wino_tm.log
: > $L
run() {  # tm out-form extra-env...
  local tm=$1 form=$2; shift 2
  echo "== TM=$tm out-form=$form $*" >> $L
  env "$@" AGZ_WINO_H2_TM=$tm AGZ_WINO_H2_OUT_PAIR=$form timeout 60 python scripts/nn_bench.py --wino-h2 --iters 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); w=d['wino']; print(round(d['ms_per_pass'],3), round(d['conv_ms_avg'],4), 'in', round(w['in_ms_avg'],4), 'gemm', round(w['gemm_ms_avg'],4), 'out', round(w['out_ms_avg'],4))" >> $L
}
run 5 3 AGZ_X=0; run 5 0 AGZ_X=0; run 4 0 AGZ_X=0; run 4 3 AGZ_X=0
run 5 3 AGZ_WINO_H2_IN_SWAP=0; run 5 3 AGZ_WINO_H2_FUSE_MAX=0; run 5 3 AGZ_WINO_H2_LAYOUT=plain
echo "== parity at the defaults" >> $L
timeout 400 python -m pytest tests/test_wino_gpu.py -q -m gpu --tb=short 2>&1 | tail -3 >> $L
timeout 300 python -m pytest tests/test_headline_parity_gpu.py -q -m gpu --tb=short -k "headline_network and wino_h2" -s 2>&1 | grep -E "parity|passed|failed" >> $L
cat $L
