#!/bin/bash
# round 3, call 2: timing-only probes of the fused kernel (what bounds the 0.93 ms?)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r3_call2.log
: > $O
for var in 1 4 11 12 14 16 18 19; do
  echo "== timing fused var=$var" >> $O
  AGZ_WINO_H2_FUSED=1 AGZ_WINO_H2_FUSED_VAR=$var timeout 300 python scripts/nn_bench.py --wino-h2 --iters 3 2>&1 | grep -v amdgpu.ids >> $O
done
python3 - <<'PY' >> $O
import json
for line in open('gpurun_out/r3_call2.log'):
    if line.startswith('=='): print(line.strip(), end='  ')
    elif line.startswith('{'):
        d=json.loads(line); print('gemm %.4f in %.4f pass %.3f'%(d['wino']['gemm_ms_avg'], d['wino']['in_ms_avg'], d['ms_per_pass']))
PY
tail -12 $O
