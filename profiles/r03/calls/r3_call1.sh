#!/bin/bash
# round 3, first GPU call: parity of the fused Winograd kernel (three variants, forced at every batch size) + NN-only timing A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r3_call1.log
: > $O
for var in 0 1 2; do
  echo "== parity fused var=$var" >> $O
  AGZ_WINO_H2_FUSED=2 AGZ_WINO_H2_FUSED_VAR=$var timeout 300 python scripts/fused_check.py >> $O 2>&1
done
echo "== parity unfused" >> $O
AGZ_WINO_H2_FUSED=0 timeout 300 python scripts/fused_check.py >> $O 2>&1
for rep in 1 2; do
  echo "== timing unfused (rep $rep)" >> $O
  AGZ_WINO_H2_FUSED=0 timeout 300 python scripts/nn_bench.py --wino-h2 --iters 5 >> $O 2>&1
  for var in 0 1 2; do
    echo "== timing fused var=$var (rep $rep)" >> $O
    AGZ_WINO_H2_FUSED=1 AGZ_WINO_H2_FUSED_VAR=$var timeout 300 python scripts/nn_bench.py --wino-h2 --iters 5 >> $O 2>&1
  done
done
tail -c 6000 $O
