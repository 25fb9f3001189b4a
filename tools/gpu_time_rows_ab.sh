#!/bin/bash
# A/B of the batched timestep rows (DiffusionInferer.BATCHED_TIME_ROWS) on one box: C2 bench (eager launches) and the C3 latent chain (graph replay)
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for v in 0 1; do
  echo -n "GM_BATCHED_TIME_ROWS=$v C2: "; GM_BATCHED_TIME_ROWS=$v timeout 200 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_ddim_iteration'])"
done; done
for v in 0 1; do echo -n "GM_BATCHED_TIME_ROWS=$v C3: "; GM_BATCHED_TIME_ROWS=$v timeout 200 python tools/bench_c3.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['sample_s'], d['sample_s_hip_graph'], d['graph_vs_eager_maxdiff'])"; done
