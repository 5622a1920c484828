#!/bin/bash
# round 3, call 5: Z-form GEMM (row transform inside the GEMM kernel): parity, timing A/B, PMC
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r3_call5.log
: > $O
echo "== parity Z forced" >> $O
AGZ_WINO_H2_Z=2 timeout 300 python scripts/fused_check.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== parity Z forced, TM=4" >> $O
AGZ_WINO_H2_Z=2 AGZ_WINO_H2_TM=4 timeout 300 python scripts/fused_check.py 2>&1 | grep -v amdgpu.ids >> $O
for rep in 1 2; do
for z in 0 1; do
  echo "== timing Z=$z" >> $O
  AGZ_WINO_H2_Z=$z timeout 300 python scripts/nn_bench.py --wino-h2 --iters 5 2>&1 | grep -v amdgpu.ids >> $O
done
done
PMC_GROUPS="fetch write sq tcc tcp grbm" PMC_PASS_TIMEOUT=150 bash scripts/pmc_run.sh gpurun_out/pmc_z wino_gemm_z -- python $PWD/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/pmc_z.json 2>&1
python3 - <<'PY' >> $O
import json
for line in open('gpurun_out/r3_call5.log'):
    if line.startswith('=='): print(line.strip(), end='  ')
    elif line.startswith('{"B"'):
        d=json.loads(line); w=d['wino']; print('in %.4f gemm %.4f out %.4f pass %.3f'%(w['in_ms_avg'], w['gemm_ms_avg'], w['out_ms_avg'], d['ms_per_pass']))
    elif line.startswith('{"env"'):
        d=json.loads(line)
        for r in d['results']: print('\n  ',r['shape'], 'dpol_f32 %.2e dpol_or %.2e f32_or %.2e dval %.2e fin %s'%(r['dpol_f32'],r['dpol_oracle'],r['f32_dpol_oracle'],r['dval_f32'],r['finite']), end='')
        print()
    elif 'Error' in line or 'error' in line: print(line.strip()[:300])
PY
tail -28 $O; tail -3 gpurun_out/pmc_z.json
