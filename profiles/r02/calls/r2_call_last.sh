#!/bin/bash
# L2-side counters of the default GEMM (TCC / TCP groups) and a sweep of the weight gradient's rows per workgroup
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
export PMC_GROUPS="tcc tcp ta"; export PMC_PASS_TIMEOUT=60
bash scripts/pmc_run.sh gpurun_out/pmc_gemm_l2 wino_gemm_h2d -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/pmc_gemm_l2.json 2>&1
tail -1 gpurun_out/pmc_gemm_l2.json
for r in 1024 2048 4096 8192; do echo "wgrad rows $r"; AGZ_WGRAD_ROWS=$r timeout 200 python scripts/train_bench.py --wino-h2 2>/dev/null | tail -1 | cut -c1-120; done
