#!/bin/bash
# where does the output transform wait?  SQ / TA / TCP / TCC counter groups for the output- and the input-transform kernel
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
export PMC_GROUPS="sq act vmem ta tcp tcc grbm"; export PMC_PASS_TIMEOUT=60
bash scripts/pmc_run.sh gpurun_out/pmc_out_tile wino_out_ -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/pmc_out_tile.json 2>&1
bash scripts/pmc_run.sh gpurun_out/pmc_in_k wino_in_h2 -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/pmc_in_k.json 2>&1
tail -1 gpurun_out/pmc_out_tile.json; tail -1 gpurun_out/pmc_in_k.json
