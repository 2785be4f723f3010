#!/bin/bash
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
run() { echo -n "[$*] "; env "$@" timeout 300 python bench.py --order $ORDER --cpu-sample 0 --start planted --steps 6 --warmup 2 --no-regimes 2>/tmp/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k:(round(v['kernel_ms'],2) if v['kernel_ms'] else None) for k,v in d['roofline']['by_kernel'].items()})" || tail -3 /tmp/err.log; }
for v in X=1 "SPKM_ACC_THREADS=512 SPKM_ACC_BLOCKS=2" "SPKM_SEG=2048" "SPKM_SEG=4096" "SPKM_SEG=16384" "SPKM_SEG=32768" "SPKM_ACC_THREADS=512 SPKM_ACC_BLOCKS=2 SPKM_SEG=4096"; do
  ORDER=block run $v 2>&1 | sed "s/^/block /"
done
ORDER=shuffled run X=1 | sed "s/^/shuffled /"
ORDER=shuffled run SPKM_ACC_THREADS=512 SPKM_ACC_BLOCKS=2 | sed "s/^/shuffled /"
