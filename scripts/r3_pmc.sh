#!/bin/bash
# FETCH / WRITE PMC passes of the three Winograd fp16x2 kernels on the full 512-board launch (one queue: under two queues every launch
# is a half batch)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export AGZ_WINO_H2_QUEUES=1 PMC_GROUPS="fetch write" PMC_PASS_TIMEOUT=60
bash scripts/pmc_run.sh gpurun_out/pmc_r3_gemm wino_gemm_h2d -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/r3_pmc_wino_h2_gemm.json 2>&1
bash scripts/pmc_run.sh gpurun_out/pmc_r3_in wino_in_h2 -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/r3_pmc_wino_h2_in.json 2>&1
bash scripts/pmc_run.sh gpurun_out/pmc_r3_out wino_out_ -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/r3_pmc_wino_h2_out.json 2>&1
tail -1 gpurun_out/r3_pmc_wino_h2_gemm.json; tail -1 gpurun_out/r3_pmc_wino_h2_in.json; tail -1 gpurun_out/r3_pmc_wino_h2_out.json
rm -rf gpurun_out/pmc_r3_gemm gpurun_out/pmc_r3_in gpurun_out/pmc_r3_out
