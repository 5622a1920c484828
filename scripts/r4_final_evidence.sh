#!/bin/bash
# Round 4's last GPU call on the final tree: GPU suite, smoke, default bench line, the ONE-QUEUE bench under rocprofv3 --kernel-trace --stats
# (full-batch launches: the csv the roofline's launch duration agrees with), trainer step per mode + per-kernel averages.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -q > gpurun_out/r4f_gpu_suite.log 2>&1; tail -2 gpurun_out/r4f_gpu_suite.log
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py > gpurun_out/r4f_bench_n1.json 2> gpurun_out/r4f_bench_n1.err; tail -c 200 gpurun_out/r4f_bench_n1.json
LEGS="--no-cpu-baseline --no-games-leg --no-go9-leg --no-latency-leg --no-f32-leg --no-train-leg"
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench1 -- python $R/bench.py --tower-queues 1 $LEGS > $R/gpurun_out/r4f_bench_one_queue_under_rocprof.json 2> /dev/null )
find gpurun_out/prof_bench1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4f_bench_kernel_stats_one_queue.csv
rm -rf gpurun_out/prof_bench1
head -12 gpurun_out/r4f_bench_kernel_stats_one_queue.csv
{ for m in "" "--x3" "--wino-h2"; do echo "train_bench.py $m"; python scripts/train_bench.py $m | tail -1; done; } > gpurun_out/r4f_train_step_modes.log 2>&1
cat gpurun_out/r4f_train_step_modes.log
bash scripts/train_prof.sh --wino-h2 --steps 2 > /dev/null 2>&1
