#!/bin/bash
# Soak: every fuzz family of the GPU suite on fresh seeds (device vs oracle).  usage: scripts/fuzz_soak.sh BASE N_ENGINE [N_OTHER [N_NARROW]]
# Failures are listed (seed in the test id) in gpurun_out/fuzz_soak_<BASE>.log; nothing stops at the first one.
BASE=${1:-100000}; NE=${2:-2000}; NO=${3:-200}; ND=${4:-$NE}   # AGZ_FUZZ_WIDE=1 in the environment: the broader engine distribution
mkdir -p gpurun_out
LOG=gpurun_out/fuzz_soak_$BASE.log
: > $LOG
AGZ_FUZZ_BASE=$BASE AGZ_FUZZ_N=$NE timeout 300 python -m pytest tests/test_engine_fuzz_gpu.py -q -m gpu -p no:cacheprovider --tb=line 2>&1 | grep -v "^\.*  *\[" | tail -60 >> $LOG
AGZ_FUZZ_BASE=$BASE AGZ_FUZZ_N=$NO timeout 200 python -m pytest tests/test_net_fuzz_gpu.py tests/test_train_fuzz_gpu.py tests/test_tournament_gpu.py -q -m gpu -p no:cacheprovider --tb=line -k "fuzz or random" 2>&1 | grep -v "^\.*  *\[" | tail -60 >> $LOG
# the tie-heavy family (round 6): narrow trees through host inferencers on both sides
AGZ_FUZZ_BASE=$BASE AGZ_FUZZ_N=$ND timeout 900 python -m pytest tests/test_deep_tree_fuzz_gpu.py -q -m gpu -p no:cacheprovider --tb=line 2>&1 | grep -v "^\.*  *\[" | tail -20 >> $LOG
cat $LOG | tail -60
