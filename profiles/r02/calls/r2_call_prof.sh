#!/bin/bash
# profiles for round 2: rocprofv3 kernel stats of the bench command (legs off: the per-kernel averages do not depend on them),
# FETCH/WRITE PMC passes of the default Winograd fp16x2 GEMM and of the two transform kernels, and the two new
# (the GEMM, the input transform and the output transform of the default F(5x5,3x3) tile)
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-games-leg --no-go9-leg --no-latency-leg --no-f32-leg > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/bench_under_rocprof.err )
find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/bench_kernel_stats_r02.csv
find gpurun_out/prof_bench -name "*domain_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/bench_domain_stats_r02.csv
rm -rf gpurun_out/prof_bench
head -8 gpurun_out/bench_kernel_stats_r02.csv
export PMC_GROUPS="fetch write sq grbm lds"; export PMC_PASS_TIMEOUT=60
bash scripts/pmc_run.sh gpurun_out/pmc_h2_gemm_final wino_gemm_h2d -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/pmc_h2_gemm_final.json 2>&1
export PMC_GROUPS="fetch write"
bash scripts/pmc_run.sh gpurun_out/pmc_h2_in wino_in_h2 -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/pmc_h2_in.json 2>&1
bash scripts/pmc_run.sh gpurun_out/pmc_h2_out wino_out_ -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/pmc_h2_out.json 2>&1
tail -1 gpurun_out/pmc_h2_gemm_final.json; tail -1 gpurun_out/pmc_h2_in.json; tail -1 gpurun_out/pmc_h2_out.json
