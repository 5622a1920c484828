#!/bin/bash
# Round 6's evidence, one GPU call on the final tree: the GPU suite, smoke, the default bench line; the ONE-QUEUE bench under
# rocprofv3 --kernel-trace --stats (full-batch launches: the csv the roofline's launch duration has to agree with); HBM counters of the chained
# block's two kernels (separate --pmc passes, scripts/r5_pmc_tower.sh); the trainer step per mode and its kernel table.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r6_gpu_suite.log 2>&1; grep -E "passed|failed|error" gpurun_out/r6_gpu_suite.log | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r6_bench_n1.json 2> gpurun_out/r6_bench_n1.err; tail -c 300 gpurun_out/r6_bench_n1.err
python -c "
import json; d=json.load(open('gpurun_out/r6_bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['whole_move'], d['roofline']['frac'], d['extra']['train_leg']['step_ms'])"
LEGS="--no-cpu-baseline --no-games-leg --no-go9-leg --no-latency-leg --no-f32-leg --no-train-leg --no-deep-leg --no-complete-games-leg"
python bench.py --tower-queues 1 $LEGS > gpurun_out/r6_bench_n1_one_queue.json 2> /dev/null
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench1 -- python $R/bench.py --tower-queues 1 $LEGS > $R/gpurun_out/r6_bench_one_queue_under_rocprof.json 2> /dev/null )
find gpurun_out/prof_bench1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r6_bench_kernel_stats_one_queue.csv
rm -rf gpurun_out/prof_bench1
head -6 gpurun_out/r6_bench_kernel_stats_one_queue.csv | cut -c1-160
PMC_PASS_TIMEOUT=150 bash scripts/r5_pmc_tower.sh > gpurun_out/r6_pmc_tower.log 2>&1; tail -3 gpurun_out/r6_pmc_tower.log
{ for m in "" "--x3" "--wino-h2"; do echo "train_bench.py $m"; python scripts/train_bench.py $m --steps 8 | tail -1; done; } > gpurun_out/r6_train_step_modes.log 2>&1
cat gpurun_out/r6_train_step_modes.log
bash scripts/train_prof.sh --wino-h2 --steps 3 > gpurun_out/r6_train_kernels.txt 2>&1; head -12 gpurun_out/r6_train_kernels.txt
