#!/bin/bash
# k_conv_h2dma3 (nine taps of a chunk from one x image; default) against k_conv_h2dma (hook bit 5): parity tests, A/B in one process, decomposition
mkdir -p gpurun_out
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 400 python -m pytest tests/test_train_gpu.py -m gpu -q -x > gpurun_out/r5_conv3_tests.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r5_conv3_tests.log | tail -5
timeout 300 python scripts/train_bench.py --wino-h2 --steps 6 --fb --hooks 1,33,1,33,65,129,193 > gpurun_out/r5_conv3_ab.log 2>&1; tail -2 gpurun_out/r5_conv3_ab.log | head -1 | cut -c1-900
