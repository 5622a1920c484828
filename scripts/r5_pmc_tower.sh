#!/bin/bash
# Round 5: HBM counters of the chained dual block's two kernels on the round's tree (M and V2c stored non-temporally since round 4's pass).
# One rocprofv3 --pmc pass per counter (--kernel-trace only), tower on ONE queue so that a launch is the whole 512-board batch.
# -> gpurun_out/r5_pmc_tower/pmc_wino_h2c.json (same fields as profiles/pmc_wino_h2c.json of round 4)
set -u
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$repo/gpurun_out/r5_pmc_tower
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export AGZ_WINO_H2_QUEUES=1
cmd="python $repo/scripts/nn_bench.py --wino-h2 --L 4 --iters 2"
i=0
for g in FETCH_SIZE WRITE_SIZE; do
  timeout ${PMC_PASS_TIMEOUT:-120} rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out/p$i" -- $cmd > "$out/p$i.log" 2>&1
  i=$((i+1))
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
pats = {"wino_gemm_h2g_kernel<8>": "wino_gemm_h2g_kernel<8", "wino_oip_h2c_kernel<5>": "wino_oip_h2c_kernel<5"}
agg = {k: collections.defaultdict(lambda: [0.0, 0]) for k in pats}
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k, p in pats.items():
            if p in r["Kernel_Name"]:
                a = agg[k][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
T, K, N, NPOS = 8192, 256, 512, 49          # G19, B = 512: 16 tiles of 5x5 per board
v2c = NPOS * T * K * 4                      # fp16 hi + lo
mc = NPOS * T * N * 4
u2c = NPOS * K * N * 4
alg = {"wino_gemm_h2g_kernel<8>": {"read": v2c + u2c, "write": mc}, "wino_oip_h2c_kernel<5>": {"read": mc, "write": v2c}}
res = {"round": "r05",
       "collected_with": "scripts/r5_pmc_tower.sh: AGZ_WINO_H2_QUEUES=1, one rocprofv3 --pmc pass per counter (--kernel-trace only) around scripts/nn_bench.py --wino-h2 --L 4 --iters 2",
       "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE as reported (KB)",
       "workload": "G19 chained dual block, B=512: 49 x (M=8192 tiles, N=512, K=256); M and V2c stored non-temporally"}
for k in pats:
    f = agg[k].get("FETCH_SIZE", [0.0, 0]); w = agg[k].get("WRITE_SIZE", [0.0, 0])
    if not f[1] or not w[1]:
        res[k] = {"error": "no launches counted"}; continue
    fk, wk = f[0] / f[1], w[0] / w[1]
    fb, wb = int(2 * fk * 1024), int(wk * 1024)
    a = alg[k]
    res[k] = {"FETCH_SIZE_KB_raw_avg": fk, "WRITE_SIZE_KB_avg": wk, "launches": min(f[1], w[1]), "fetch_bytes_per_launch_corrected": fb,
              "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
              "algorithmic_bytes_per_launch": {"read": a["read"], "write": a["write"], "total": a["read"] + a["write"]},
              "traffic_over_algorithmic": round((fb + wb) / (a["read"] + a["write"]), 3)}
g = res.get("wino_gemm_h2g_kernel<8>", {})
if "hbm_bytes_per_launch" in g:
    res["hbm_bytes_per_launch"] = g["hbm_bytes_per_launch"]
    res["kernel"] = "agz::wino_gemm_h2g_kernel<8, true>(agz::WinoH2Args)"
json.dump(res, open(out + "/pmc_wino_h2c.json", "w"), indent=1)
print(json.dumps({k: res[k] for k in pats}, indent=1))
PY
rm -rf "$out"/p*/
