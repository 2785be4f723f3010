#!/bin/bash
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2e; mkdir -p $out
cd $root
run() { echo -n "[$*] "; env "$@" timeout 300 python bench.py --order $ORDER --cpu-sample 0 --start planted --steps 6 --warmup 2 --no-regimes 2>$out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k:(round(v['kernel_ms'],2) if v['kernel_ms'] else None) for k,v in d['roofline']['by_kernel'].items()})" || tail -3 $out/err.log; }
for e in ${EXPS:-0 15}; do
  ORDER=block run SPKM_REC=1 SPKM_REC_PIPE=1 SPKM_EXP=$e 2>&1 | sed "s/^/block /"
done 2>&1 | tee $out/acc_exp.txt
ORDER=shuffled run SPKM_REC=1 SPKM_REC_PIPE=1 2>&1 | sed "s/^/shuffled /" | tee -a $out/acc_exp.txt
