#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_engine_edges_gpu.py tests/test_engine_fuzz_gpu.py tests/test_tournament_gpu.py tests/test_random_moves_gpu.py tests/test_parallel_lanes_gpu.py tests/test_mcts_handle_gpu.py tests/test_lifecycle_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-games-leg --no-go9-leg --no-latency-leg --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['extra']['full_move_19x19']['sims_per_s'], d['extra']['timed_region'])"
