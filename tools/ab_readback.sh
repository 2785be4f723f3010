#!/bin/bash
# GPU box: the bench loop with spkm_lloyd_iter_host (results through mapped host memory) against iterate + a device-to-host read.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/abrb
run() { # tag, env, args...
  local tag=$1 rb=$2; shift 2
  SPKM_BENCH_READBACK=$rb timeout 500 python bench.py "$@" --no-pmc --cpu-sample 0 --no-regimes 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag readback=$rb', round(b['value'],2), 'it/s', round(b['ms_per_step'],4), 'ms')"
}
for rb in "" 1 "" 1; do run shard100 "$rb" --n-total 1.25e7 --steps 100 --warmup 5; done
for rb in "" 1; do run shard20 "$rb" --n-total 1.25e7 --steps 20 --warmup 5; done
for rb in "" 1; do run head100 "$rb" --steps 100 --warmup 5; done
for rb in "" 1; do run head20 "$rb" --steps 20 --warmup 5; done
