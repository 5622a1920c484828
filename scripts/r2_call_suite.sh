#!/bin/bash
# the whole GPU suite (what the driver runs at round end) + the default bench
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | tail -15 > gpurun_out/gpu_suite.log
cat gpurun_out/gpu_suite.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02_b.json 2> gpurun_out/bench_r02_b.err
tail -2 gpurun_out/bench_r02_b.err; wc -c gpurun_out/bench_r02_b.json
