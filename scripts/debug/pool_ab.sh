#!/bin/bash
# A/B: does the node-pool size (the default went from two to four searches' worth of expansions in round 6) move the bench?  One box, one queue.
LEGS="--no-cpu-baseline --no-games-leg --no-go9-leg --no-latency-leg --no-f32-leg --no-train-leg --no-deep-leg --no-complete-games-leg"
for mn in 580664 0 580664 0; do
  python bench.py --tower-queues 1 --max-nodes $mn $LEGS 2>/dev/null > /tmp/ab.json
  python -c "import json; d=json.load(open('/tmp/ab.json')); print('max_nodes', $mn, 'value', round(d['value']), 'gemm_ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'hbm_GB', round(d['config']['hbm_used_bytes_per_rank'][0]/1e9,1))"
done
