#!/bin/bash
# rocprofv3 kernel stats of the bench command (legs off) + the one-queue bench line, on the final tree
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python bench.py --tower-queues 1 --no-cpu-baseline --no-games-leg --no-go9-leg --no-latency-leg --no-f32-leg > gpurun_out/r3_bench_n1_one_queue.json 2> /dev/null
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python $R/bench.py --no-cpu-baseline --no-games-leg --no-go9-leg --no-latency-leg --no-f32-leg > $R/gpurun_out/r3_bench_under_rocprof.json 2> $R/gpurun_out/r3_bench_under_rocprof.err )
find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3_bench_kernel_stats.csv
find gpurun_out/prof_bench -name "*domain_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3_bench_domain_stats.csv
rm -rf gpurun_out/prof_bench
head -5 gpurun_out/r3_bench_kernel_stats.csv
python - <<'PY'
import json
for f in ("r3_bench_n1_one_queue", "r3_bench_under_rocprof"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    kc = d["extra"]["kernel_classes"]
    print(f, round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], kc["wino_in"]["avg_ms"], kc["wino_gemm"]["avg_ms"], kc["wino_out"]["avg_ms"])
PY
export AGZ_WINO_H2_QUEUES=1 PMC_GROUPS="fetch write" PMC_PASS_TIMEOUT=60
bash scripts/pmc_run.sh gpurun_out/pmc_r3_out wino_out_ -- python $R/scripts/nn_bench.py --wino-h2 --L 4 --iters 2 > gpurun_out/r3_pmc_wino_h2_out.json 2>&1
tail -1 gpurun_out/r3_pmc_wino_h2_out.json; rm -rf gpurun_out/pmc_r3_out
