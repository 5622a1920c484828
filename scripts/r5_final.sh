#!/bin/bash
# Round 5's last GPU call on the final tree: GPU suite, smoke, fuzz soak on fresh seeds, default bench line.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -q > gpurun_out/r5f_gpu_suite.log 2>&1; tail -2 gpurun_out/r5f_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -1
bash scripts/fuzz_soak.sh 550000 1500 150 > /dev/null 2>&1; tail -4 gpurun_out/fuzz_soak_550000.log
python bench.py > gpurun_out/r5f_bench_n1.json 2> gpurun_out/r5f_bench_n1.err; python -c "
import json; d=json.load(open('gpurun_out/r5f_bench_n1.json')); print(d['value'], d['ms_per_step'], d['full_move_sims_per_s'], d['roofline']['frac'], d['extra']['train_leg']['step_ms'])"
