#!/bin/bash
# A/B on ONE box: the 512-board tower (scripts/nn_bench.py --wino-h2, one queue and two) of round 5's tree (.r5tree, a worktree of the round's first
# commit, built beforehand) against this tree's — the inference kernels are unchanged, so any difference between rounds is the box's
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  for t in .r5tree .; do
    for q in 1 2; do
      ( cd $R/$t && AGZ_WINO_H2_QUEUES=$q python scripts/nn_bench.py --wino-h2 --iters 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$t', 'queues', $q, 'ms_per_pass', round(d['ms_per_pass'],3), 'gemm', d['wino'] if not isinstance(d.get('wino'),dict) else {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['wino'].items()})" )
    done
  done
done
