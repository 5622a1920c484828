#!/bin/bash
# Round 3's closing evidence, one GPU call: the default bench line, rocprofv3 kernel stats of the same command (legs off), FETCH / WRITE
# PMC passes of the three Winograd kernels, trainer step times per mode + per-kernel stats, the batch-1 kernel trace and probe.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python bench.py > gpurun_out/r3_bench_n1.json 2> gpurun_out/r3_bench_n1.err; tail -c 300 gpurun_out/r3_bench_n1.err
python bench.py --tower-queues 1 --no-cpu-baseline --no-games-leg --no-go9-leg --no-latency-leg --no-f32-leg > gpurun_out/r3_bench_n1_one_queue.json 2> /dev/null
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python $R/bench.py --no-cpu-baseline --no-games-leg --no-go9-leg --no-latency-leg --no-f32-leg > $R/gpurun_out/r3_bench_under_rocprof.json 2> $R/gpurun_out/r3_bench_under_rocprof.err )
find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3_bench_kernel_stats.csv
find gpurun_out/prof_bench -name "*domain_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3_bench_domain_stats.csv
rm -rf gpurun_out/prof_bench
head -6 gpurun_out/r3_bench_kernel_stats.csv
export PMC_GROUPS="fetch write"; export PMC_PASS_TIMEOUT=60
bash scripts/pmc_run.sh gpurun_out/pmc_r3_gemm wino_gemm_h2d -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/r3_pmc_wino_h2_gemm.json 2>&1
bash scripts/pmc_run.sh gpurun_out/pmc_r3_in wino_in_h2 -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/r3_pmc_wino_h2_in.json 2>&1
bash scripts/pmc_run.sh gpurun_out/pmc_r3_out wino_out_ -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/r3_pmc_wino_h2_out.json 2>&1
tail -1 gpurun_out/r3_pmc_wino_h2_gemm.json; tail -1 gpurun_out/r3_pmc_wino_h2_in.json; tail -1 gpurun_out/r3_pmc_wino_h2_out.json
rm -rf gpurun_out/pmc_r3_gemm gpurun_out/pmc_r3_in gpurun_out/pmc_r3_out
{ for m in "" "--x3" "--wino-h2"; do echo "train_bench.py $m"; python scripts/train_bench.py $m | tail -1; done; } > gpurun_out/r3_train_step_modes.log 2>&1
bash scripts/train_prof.sh --wino-h2 > /dev/null 2>&1; cat gpurun_out/trainprof_summary.txt >> gpurun_out/r3_train_step_modes.log
cat gpurun_out/r3_train_step_modes.log | head -8
bash scripts/r3_lat_prof.sh > /dev/null 2>&1; cp gpurun_out/latprof_summary.txt gpurun_out/r3_latency_kernel_trace_summary.txt
python scripts/latency_bench.py --moves 3 --open 60 | tail -1 > gpurun_out/r3_latency_lanes1.json
python scripts/latency_bench.py --moves 3 --open 60 --lanes 8 | tail -1 > gpurun_out/r3_latency_lanes8.json
python scripts/latency_bench.py --moves 3 --open 60 --lanes 16 --compute wino_h2 | tail -1 > gpurun_out/r3_latency_lanes16_wino_h2.json
head -c 400 gpurun_out/r3_latency_lanes1.json; echo
