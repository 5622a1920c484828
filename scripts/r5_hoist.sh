#!/bin/bash
# weight images at the start of the step on the side stream (hook bit 3 clear) against per layer in line (set): trainer tests, then A/B in one process
mkdir -p gpurun_out
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_train_fuzz_gpu.py tests/test_comm_fake_gpu.py tests/test_comm_gpu.py tests/test_learn_parity_gpu.py -m gpu -q -x > gpurun_out/r5_hoist_tests.log 2>&1; tail -3 gpurun_out/r5_hoist_tests.log
timeout 300 python scripts/train_bench.py --wino-h2 --steps 8 --hooks 1,9,1,9,1,9 > gpurun_out/r5_hoist_ab.log 2>&1; tail -2 gpurun_out/r5_hoist_ab.log
