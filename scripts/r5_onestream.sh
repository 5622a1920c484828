#!/bin/bash
# the trainer's kernels without overlap (hook bit 4: no side stream): every kernel's own duration
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/train_prof.sh --wino-h2 --steps 3 --hook 17 > /dev/null 2>&1
cp gpurun_out/trainprof_summary.txt gpurun_out/r5_trainprof_one_stream.txt; cat gpurun_out/r5_trainprof_one_stream.txt; tail -1 gpurun_out/trainprof_run.log | cut -c1-100
