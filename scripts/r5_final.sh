#!/bin/bash
# Round 5's last GPU call on the final tree: GPU suite, smoke, default bench line.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r5f_gpu_suite.log 2>&1; grep -E "passed|failed|error" gpurun_out/r5f_gpu_suite.log | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r5f_bench_n1.json 2> gpurun_out/r5f_bench_n1.err; python -c "
import json; d=json.load(open('gpurun_out/r5f_bench_n1.json')); print(d['value'], d['ms_per_step'], d['full_move_sims_per_s'], d['roofline']['frac'], d['extra']['train_leg']['step_ms'])"
