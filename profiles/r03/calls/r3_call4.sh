#!/bin/bash
# round 3, call 3: fused kernel with contiguous layouts + LDS epilogue: parity, timing probes, PMC
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r3_call4.log
: > $O
echo "== parity fused" >> $O
AGZ_WINO_H2_FUSED=2 timeout 300 python scripts/fused_check.py 2>&1 | grep -v amdgpu.ids >> $O
for probe in 0 99 32 6 38 40 16 2 4; do
  echo "== timing fused probe=$probe" >> $O
  AGZ_WINO_H2_FUSED=1 AGZ_WINO_H2_FUSED_PROBE=$probe timeout 300 python scripts/nn_bench.py --wino-h2 --iters 3 2>&1 | grep -v amdgpu.ids >> $O
done
PMC_GROUPS="tcc tcp fetch sq" PMC_PASS_TIMEOUT=150 AGZ_WINO_H2_FUSED=1 bash scripts/pmc_run.sh gpurun_out/pmc_fused wino_fused4 -- python $PWD/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/pmc_fused.json 2>&1
python3 - <<'PY' >> $O
import json
for line in open('gpurun_out/r3_call4.log'):
    if line.startswith('=='): print(line.strip(), end='  ')
    elif line.startswith('{"B"'):
        d=json.loads(line); print('gemm %.4f in %.4f pass %.3f'%(d['wino']['gemm_ms_avg'], d['wino']['in_ms_avg'], d['ms_per_pass']))
    elif line.startswith('{"env"'):
        d=json.loads(line)
        for r in d['results']: print('\n  ',r['shape'], 'dpol_f32 %.2e dpol_or %.2e f32_or %.2e dval %.2e fin %s'%(r['dpol_f32'],r['dpol_oracle'],r['f32_dpol_oracle'],r['dval_f32'],r['finite']), end='')
        print()
PY
tail -30 $O; tail -3 gpurun_out/pmc_fused.json
