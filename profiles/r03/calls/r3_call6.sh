#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3_call6.log
: > $O
for probe in 0 1 2 3; do
  echo "== timing Z probe=$probe" >> $O
  AGZ_WINO_H2_Z=1 AGZ_WINO_H2_Z_PROBE=$probe timeout 300 python scripts/nn_bench.py --wino-h2 --iters 3 2>&1 | grep -v amdgpu.ids >> $O
done
python3 - <<'PY'
import json
for line in open('gpurun_out/r3_call6.log'):
    if line.startswith('=='): print(line.strip(), end='  ')
    elif line.startswith('{"B"'):
        d=json.loads(line); w=d['wino']; print('in %.4f gemm %.4f out %.4f pass %.3f'%(w['in_ms_avg'], w['gemm_ms_avg'], w['out_ms_avg'], d['ms_per_pass']))
PY
