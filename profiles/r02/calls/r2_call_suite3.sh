#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -8 > gpurun_out/gpu_suite.log
cat gpurun_out/gpu_suite.log
timeout 60 scripts/probes/stream_probe | tee gpurun_out/stream_probe.log
