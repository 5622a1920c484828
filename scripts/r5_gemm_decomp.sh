#!/bin/bash
# Round-5: what bounds wino_gemm_h2p_kernel — the same launch without its M stores (1), without its DMA (2), without either (3)
mkdir -p gpurun_out
for m in 0 1 2 3; do
  echo "== mode $m (bit 0: no M stores, bit 1: no DMA)" >> gpurun_out/r5_gemm_decomp.log
  PROBE_TIMING_ONLY=1 PROBE_VARIANTS=$((2 + 16 * m)) PROBE_QUEUES=1 PROBE_REPS=1 PROBE_OUT=gpurun_out/r5_decomp_$m.json timeout 200 python scripts/r5_gemm_probe.py 2>&1 | grep "^rep" >> gpurun_out/r5_gemm_decomp.log
done
cat gpurun_out/r5_gemm_decomp.log
