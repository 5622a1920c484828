#!/bin/bash
# per-kernel averages of one dual.Train configuration (G19, B=256): rocprofv3 --kernel-trace --stats around scripts/train_bench.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/trainprof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trainprof -o tr -- python $R/scripts/train_bench.py "$@" > $R/gpurun_out/trainprof_run.log 2>&1
tail -1 $R/gpurun_out/trainprof_run.log
python - <<'PY'
import csv, os
R = os.environ["GRAFT_REPO_ROOT"]
rows = list(csv.DictReader(open(R + "/gpurun_out/trainprof/tr_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = ["%-72s n=%6s avg=%9.1f us  %5.1f%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot) for r in rows[:18]]
open(R + "/gpurun_out/trainprof_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -f $R/gpurun_out/trainprof/*trace.csv
