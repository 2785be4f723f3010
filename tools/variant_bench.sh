#!/bin/bash
# GPU box: the headline's traced run (per-iteration times of both hot kernels) for each experiment build given
#   tools/variant_bench.sh a2 a4 ...        ("base" = the shipped library)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
cp sparsifiedkmeans_amd/libspkm.so /tmp/libspkm_base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/libspkm_base.so sparsifiedkmeans_amd/libspkm.so; else cp build_tmp/libspkm_$v.so sparsifiedkmeans_amd/libspkm.so; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample 0 ${VARIANT_ARGS:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value'],2),'it/s')
for o,g in d.get('regimes',{}).items():
    print('   ',o,round(g['run_to_convergence_iters_per_s'],1),'it/s; per iter',[round(x,1) for x in g['per_iter_ms'][:11]])
    print('        screen',[round(x,1) for x in g['kernels_ms']['k_screen_quad'][:11]])
"
done
cp /tmp/libspkm_base.so sparsifiedkmeans_amd/libspkm.so
