#!/bin/bash
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2d; mkdir -p $out
cd $root
( SPKM_REC=1 SPKM_REC_PIPE=1 timeout 900 python -m pytest tests/test_gpu_screen.py tests/test_gpu_fullsize.py tests/test_gpu_lloyd.py tests/test_gpu_sweeps.py -m gpu -x -q ) > $out/tests_recpipe.log 2>&1
tail -3 $out/tests_recpipe.log
run() { echo -n "[$*] "; env "$@" timeout 300 python bench.py --order $ORDER --cpu-sample 0 --start planted --steps 6 --warmup 2 --no-regimes 2>$out/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k:(round(v['kernel_ms'],2) if v['kernel_ms'] else None) for k,v in d['roofline']['by_kernel'].items()})" || tail -3 $out/err.log; }
for order in block shuffled; do
  for v in X=1 "SPKM_REC=1 SPKM_ACC_NT=1" "SPKM_REC=1 SPKM_REC_PIPE=1"; do
    ORDER=$order run $v 2>&1 | sed "s/^/$order /"
  done
done 2>&1 | tee $out/acc_variants.txt
