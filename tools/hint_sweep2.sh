#!/bin/bash
# GPU box: the headline's cold run under different hint weights / thresholds (environment switches of libspkm.so)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for cfg in "2 1.5" "1 1.5" "0.5 1.5" "0 1.5" "0 1.25" "0 1.1" "1 1.25"; do
  set -- $cfg
  SPKM_HINT_W=$1 SPKM_HINT_C=$2 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample 0 ${SWEEP_ARGS:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['regimes']['block']
print('w=$1 c=$2', round(d['value'],2),'it/s; run', round(g['run_to_convergence_iters_per_s'],1), '; per iter',[round(x,1) for x in g['per_iter_ms'][:10]])
print('        screen',[round(x,1) for x in g['kernels_ms']['k_screen_quad'][:10]], 'uncertified last', d['config'].get('uncertified_points_last_iter'))
"
done
