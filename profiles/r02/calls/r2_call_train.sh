#!/bin/bash
# trainer: parity of the bf16x3 weight gradient (small shapes, forced) and the G19 / B=256 step time with and without it
mkdir -p gpurun_out
L=gpurun_out/r2_train.log
: > $L
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_train_fuzz_gpu.py -q -m gpu --tb=short 2>&1 | tail -15 >> $L
echo "== train_bench fp32" >> $L
timeout 200 python scripts/train_bench.py 2>/dev/null | tail -1 >> $L
echo "== train_bench --x3, fp32 wgrad" >> $L
AGZ_WGRAD_X3=0 timeout 200 python scripts/train_bench.py --x3 2>/dev/null | tail -1 >> $L
echo "== train_bench --x3 (bf16x3 wgrad)" >> $L
timeout 200 python scripts/train_bench.py --x3 2>/dev/null | tail -1 >> $L
cat $L | head -40
echo "== train_bench --wino-h2" >> $L
timeout 200 python scripts/train_bench.py --wino-h2 2>/dev/null | tail -1 >> $L
tail -3 $L
