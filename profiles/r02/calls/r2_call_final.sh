#!/bin/bash
# round-2 closing call: full GPU suite, default bench line, rocprofv3 kernel stats of the bench command, two-rank bench on one GPU
# (gloo bookkeeping, --shared-gpu: exercises the N > 1 code path incl. the gather-leg watchdog), trainer step times
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests -q -m gpu -x --tb=short > gpurun_out/gpu_suite.log 2>&1
echo "suite rc=$?"; grep -E "passed|failed" gpurun_out/gpu_suite.log | tail -1
timeout 600 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
echo "bench rc=$?"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-games-leg --no-go9-leg --no-latency-leg --no-f32-leg > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/bench_under_rocprof.err )
find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/bench_kernel_stats_r02.csv
find gpurun_out/prof_bench -name "*domain_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/bench_domain_stats_r02.csv
rm -rf gpurun_out/prof_bench
head -6 gpurun_out/bench_kernel_stats_r02.csv
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --shared-gpu --steps 4 --warmup 1 --games 64 > gpurun_out/bench_n2_shared.json 2> gpurun_out/bench_n2_shared.err
echo "n2 rc=$?"; python - <<'PY'
import json
for f in ('gpurun_out/bench_n1.json','gpurun_out/bench_n2_shared.json'):
    try:
        lines=[l for l in open(f).read().strip().splitlines() if l.strip()]
        d=json.loads(lines[-1]); print(f, len(lines), 'line(s):', d['value'], d['ms_per_step'], d['n_gpus'], d['roofline']['frac'] if d.get('roofline') else None, d['extra'].get('examples_allgather'))
    except Exception as e: print(f, 'ERR', e)
PY
for m in "" "--x3" "--wino-h2"; do timeout 200 python scripts/train_bench.py $m 2>/dev/null | tail -1; done
AGZ_TRAIN_WINO_FWD=0 timeout 200 python scripts/train_bench.py --wino-h2 2>/dev/null | tail -1
