#!/bin/bash
# Round 4's closing evidence, one GPU call: the GPU suite, the default bench line, the ONE-QUEUE bench (full-batch launches) plain and under
# rocprofv3 --kernel-trace --stats (its csv is what the roofline's launch duration has to agree with: VERDICT r3 item 2), the two-queue
# bench under rocprofv3 (labelled: half-batch launches overlapping), FETCH / WRITE PMC passes of the chained block's two kernels, the
# memory system's own rates on 2 GB streams, the trainer step.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -q > gpurun_out/r4_gpu_suite.log 2>&1; tail -3 gpurun_out/r4_gpu_suite.log
python bench.py > gpurun_out/r4_bench_n1.json 2> gpurun_out/r4_bench_n1.err; tail -c 300 gpurun_out/r4_bench_n1.err
LEGS="--no-cpu-baseline --no-games-leg --no-go9-leg --no-latency-leg --no-f32-leg --no-train-leg"
python bench.py --tower-queues 1 $LEGS > gpurun_out/r4_bench_n1_one_queue.json 2> /dev/null
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench1 -- python $R/bench.py --tower-queues 1 $LEGS > $R/gpurun_out/r4_bench_one_queue_under_rocprof.json 2> $R/gpurun_out/r4_bench_one_queue_under_rocprof.err )
find gpurun_out/prof_bench1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4_bench_kernel_stats_one_queue.csv
rm -rf gpurun_out/prof_bench1
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench2 -- python $R/bench.py $LEGS > $R/gpurun_out/r4_bench_two_queues_under_rocprof.json 2> /dev/null )
find gpurun_out/prof_bench2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4_bench_kernel_stats_two_queues_half_batches.csv
rm -rf gpurun_out/prof_bench2
head -8 gpurun_out/r4_bench_kernel_stats_one_queue.csv
export AGZ_WINO_H2_QUEUES=1 PMC_GROUPS="fetch write" PMC_PASS_TIMEOUT=90
bash scripts/pmc_run.sh gpurun_out/pmc_r4_gemm wino_gemm_h2g -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/r4_pmc_gemm.json 2>&1
bash scripts/pmc_run.sh gpurun_out/pmc_r4_oi wino_oip_h2c -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/r4_pmc_oip.json 2>&1
tail -1 gpurun_out/r4_pmc_gemm.json; tail -1 gpurun_out/r4_pmc_oip.json
rm -rf gpurun_out/pmc_r4_gemm gpurun_out/pmc_r4_oi
unset AGZ_WINO_H2_QUEUES
./scripts/probes/rw_probe > gpurun_out/r4_rw_probe.log 2>&1
{ for m in "" "--x3" "--wino-h2"; do echo "train_bench.py $m"; python scripts/train_bench.py $m | tail -1; done; } > gpurun_out/r4_train_step_modes.log 2>&1
cat gpurun_out/r4_train_step_modes.log
