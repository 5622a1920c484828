#!/bin/bash
# Round-5: HBM-side counters of the search kernels on the bench workload after round 3 rewrote them.
# One rocprofv3 --pmc pass per counter (FETCH_SIZE, WRITE_SIZE; --kernel-trace only), the bench's steady-state steps: deep trees
# (pre-grown), a move boundary inside the run (k_end_move + k_begin_move).  Summary per kernel -> gpurun_out/r5_pmc_mcts/summary.json
set -u
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$repo/gpurun_out/r5_pmc_mcts
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cmd="python $repo/bench.py --steps 24 --warmup 4 --budget 60 --no-cpu-baseline --no-games-leg --no-go9-leg --no-latency-leg --no-train-leg --no-f32-leg"
i=0
for g in FETCH_SIZE WRITE_SIZE; do
  timeout ${PMC_PASS_TIMEOUT:-200} rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out/p$i" -- $cmd > "$out/p$i.log" 2>&1
  i=$((i+1))
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- $cmd > "$out/stats.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
pats = ["k_select(", "k_expand(", "k_begin_move(", "k_end_move("]
agg = {p: collections.defaultdict(lambda: [0.0, 0]) for p in pats}
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for p in pats:
            if p in r["Kernel_Name"]:
                a = agg[p][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
dur = {}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for p in pats:
            if p in r["Name"]:
                dur[p] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
res = {}
for p in pats:
    d = {k: v[0] / max(v[1], 1) for k, v in agg[p].items()}
    d["launches_counted"] = max((v[1] for v in agg[p].values()), default=0)
    d.update(dur.get(p, {}))
    res[p.rstrip("(")] = d
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$out"/p*/ "$out"/stats/
