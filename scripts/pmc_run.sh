#!/bin/bash
# PMC passes for one command (each --pmc group in its own rocprofv3 run, with --kernel-trace only — gpurun refuses
# pmc + sys/hip/hsa traces).  usage: scripts/pmc_run.sh <outdir> <kernel-substring> -- <command...>
# Each pass is cut off after PMC_PASS_TIMEOUT (default 120) seconds: a counter group the hardware rejects can hang the tool
# (one such run cost 20 GPU-minutes).  Keep <kernel-substring> free of shell/regex specials.
# Prints a JSON summary {counter: average per launch over launches of kernels whose name contains the substring}.
set -u
out=$1; shift; pat=$1; shift; shift
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$repo/$out"
cd /tmp && export TMPDIR=/tmp
# counter groups (one rocprofv3 pass each); choose with PMC_GROUPS="sq fetch write grbm" (default: all)
declare -A G
G[sq]="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY"
G[inst]="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"
G[lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS"
G[act]="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"
G[vmem]="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"
G[fetch]="FETCH_SIZE"          # FETCH_SIZE + WRITE_SIZE together exceed the hardware's counter budget
G[write]="WRITE_SIZE"
G[grbm]="GRBM_GUI_ACTIVE"
G[ta]="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum"
G[tcp]="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
G[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum TCC_REQ_sum"
sel=${PMC_GROUPS:-"sq inst lds act vmem fetch write grbm"}
groups=()
for n in $sel; do groups+=("${G[$n]}"); done
i=0
for g in "${groups[@]}"; do
  timeout ${PMC_PASS_TIMEOUT:-120} rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$repo/$out/p$i" -- "$@" > "$repo/$out/p$i.log" 2>&1
  i=$((i+1))
done
python3 - "$repo/$out" "$pat" <<'PY'
import csv, glob, json, sys, collections
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
res = {k: v[0] / max(v[1], 1) for k, v in sorted(agg.items())}
res["_launches"] = max((v[1] for v in agg.values()), default=0)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res))
PY
# raw traces are large: keep only the summary + logs
rm -rf "$repo/$out"/p*/
