#!/bin/bash
# full GPU suite, then the default bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x --tb=short > gpurun_out/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -5 gpurun_out/gpu_suite.log
timeout 600 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_n1.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline'])
e=d['extra']
for k in ('full_move_19x19','games_leg','go9_leg','latency_leg'):
    print(k, json.dumps(e.get(k))[:600])
print(json.dumps(e.get('wino'))[:1500])
PY
