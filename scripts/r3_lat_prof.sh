#!/bin/bash
# kernel trace of the batch-1 search: kernel durations vs in-stream gaps for the latency tower
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/latprof -o lat -- python $R/scripts/latency_bench.py --moves 1 --open 60 > $R/gpurun_out/latprof.log 2>&1
tail -1 $R/gpurun_out/latprof.log
python - <<'PY'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(R + "/gpurun_out/latprof/**/*kernel_trace.csv", recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import collections
d = collections.defaultdict(list); gaps = collections.defaultdict(list)
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"][:60]
    d[n].append(e - s)
    if prev is not None: gaps[n].append(s - prev)
    prev = e
out = []
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]:
    g = sorted(gaps[n]); 
    out.append("%-60s n=%7d avg=%8.2f us  gap_before p50=%6.2f us" % (n, len(v), sum(v) / len(v) / 1e3, (g[len(g)//2] if g else 0) / 1e3))
open(R + "/gpurun_out/latprof_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf $R/gpurun_out/latprof/*/*.db
find $R/gpurun_out/latprof -name '*kernel_trace.csv' -delete
