#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r2_lanes.log
: > $L
echo "== parity (split lane kernels)" >> $L
timeout 600 python -m pytest tests/test_parallel_lanes_gpu.py tests/test_net_gpu.py tests/test_headline_parity_gpu.py -m gpu -q -x --tb=short -k "not headline_network and not headline_engine" 2>&1 | tail -5 >> $L
echo "== parity (fused lane kernels, AGZ_LANES_FUSED=1)" >> $L
AGZ_LANES_FUSED=1 timeout 600 python -m pytest tests/test_parallel_lanes_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -3 >> $L
for cfg in "1 f32" "16 f32" "16 wino_h2" "8 wino_h2"; do
  set -- $cfg
  echo "== latency lanes=$1 compute=$2 (split lane kernels)" >> $L
  timeout 120 python scripts/latency_bench.py --lanes $1 --compute $2 --moves 4 --open 60 2>/dev/null | cut -c1-700 >> $L
done
echo "== latency lanes=16 compute=f32 AGZ_LANES_FUSED=1" >> $L
AGZ_LANES_FUSED=1 timeout 120 python scripts/latency_bench.py --lanes 16 --compute f32 --moves 4 --open 60 2>/dev/null | cut -c1-700 >> $L
cat $L
