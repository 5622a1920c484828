#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r2_engine.log
: > $L
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_engine_edges_gpu.py tests/test_engine_fuzz_gpu.py tests/test_tournament_gpu.py tests/test_random_moves_gpu.py tests/test_parallel_lanes_gpu.py tests/test_mcts_handle_gpu.py tests/test_gtp_gpu.py tests/test_host_cpp_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -5 >> $L
timeout 300 python -m pytest tests/test_headline_parity_gpu.py -m gpu -q -x --tb=short -k "one_tree" 2>&1 | tail -3 >> $L
for cfg in "1 f32" "16 wino_h2"; do
  set -- $cfg
  echo "== latency lanes=$1 compute=$2" >> $L
  timeout 120 python scripts/latency_bench.py --lanes $1 --compute $2 --moves 4 --open 60 2>/dev/null | cut -c1-700 >> $L
done
cat $L
