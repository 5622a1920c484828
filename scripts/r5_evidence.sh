#!/bin/bash
# Round 5's evidence, one GPU call on the final tree: the default bench line; the ONE-QUEUE bench under rocprofv3 --kernel-trace --stats
# (full-batch launches: the csv the roofline's launch duration has to agree with), the same with the persistent GEMM (AGZ_WINO_H2_GEMM=2);
# the memory system's own rates on 2 GB streams (appended to profiles/r05/rw_probe.json by the caller); the trainer step per mode.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python bench.py > gpurun_out/r5_bench_n1.json 2> gpurun_out/r5_bench_n1.err; tail -c 300 gpurun_out/r5_bench_n1.err
LEGS="--no-cpu-baseline --no-games-leg --no-go9-leg --no-latency-leg --no-f32-leg --no-train-leg"
python bench.py --tower-queues 1 $LEGS > gpurun_out/r5_bench_n1_one_queue.json 2> /dev/null
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench1 -- python $R/bench.py --tower-queues 1 $LEGS > $R/gpurun_out/r5_bench_one_queue_under_rocprof.json 2> /dev/null )
find gpurun_out/prof_bench1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r5_bench_kernel_stats_one_queue.csv
rm -rf gpurun_out/prof_bench1
( cd /tmp && export TMPDIR=/tmp && AGZ_WINO_H2_GEMM=2 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench2 -- python $R/bench.py --tower-queues 1 $LEGS > $R/gpurun_out/r5_bench_one_queue_persistent_gemm_under_rocprof.json 2> /dev/null )
find gpurun_out/prof_bench2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r5_bench_kernel_stats_one_queue_persistent_gemm.csv
rm -rf gpurun_out/prof_bench2
head -6 gpurun_out/r5_bench_kernel_stats_one_queue.csv | cut -c1-160
head -4 gpurun_out/r5_bench_kernel_stats_one_queue_persistent_gemm.csv | cut -c1-160
[ -x scripts/probes/rw_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/probes/rw_probe scripts/probes/rw_probe.hip
./scripts/probes/rw_probe > gpurun_out/r5_rw_probe.log 2>&1; grep "GEMM mix\|out->in" gpurun_out/r5_rw_probe.log
{ for m in "" "--x3" "--wino-h2"; do echo "train_bench.py $m"; python scripts/train_bench.py $m | tail -1; done; } > gpurun_out/r5_train_step_modes.log 2>&1
cat gpurun_out/r5_train_step_modes.log
