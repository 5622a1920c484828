#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | tail -8 > gpurun_out/gpu_suite.log
cat gpurun_out/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
AGZ_FUZZ_BASE=5000 AGZ_FUZZ_N=40 timeout 600 python -m pytest tests/test_engine_fuzz_gpu.py tests/test_net_fuzz_gpu.py -m gpu -q --tb=short 2>&1 | tail -4 > gpurun_out/fuzz_soak_r02.log
cat gpurun_out/fuzz_soak_r02.log
