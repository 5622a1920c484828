#!/bin/bash
# counters of the two transform kernels of the default build: FETCH/WRITE (traffic) and SQ / TCP / TCC groups (where they wait);
# preceded by the Winograd parity tests
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_wino_gpu.py tests/test_headline_parity_gpu.py -q -m gpu --tb=short 2>&1 | tail -3
export PMC_GROUPS="fetch write sq tcp tcc grbm"; export PMC_PASS_TIMEOUT=60
bash scripts/pmc_run.sh gpurun_out/pmc_out_seq wino_out_ -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/pmc_out_seq.json 2>&1
bash scripts/pmc_run.sh gpurun_out/pmc_in_k wino_in_h2 -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/pmc_in_k.json 2>&1
tail -1 gpurun_out/pmc_out_seq.json; tail -1 gpurun_out/pmc_in_k.json
