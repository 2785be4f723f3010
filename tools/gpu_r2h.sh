#!/bin/bash
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2h; mkdir -p $out
cd $root
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $out/all_tests.log 2>&1
grep -E "passed|failed" $out/all_tests.log; grep -B5 -A60 "^____" $out/all_tests.log | grep -E "^E|^tests|Error" | head -30
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $out/bench_headline.json 2> $out/bench_headline.err; tail -2 $out/bench_headline.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out/r2h/bench_*.json'))):
    r=json.load(open(f))
    print(os.path.basename(f), round(r['value'],2),'it/s', round(r['ms_per_step'],3),'ms', r['roofline']['kernel'], r['roofline']['frac'], {k:(round(v['kernel_ms'],3), round(v['frac'],3)) for k,v in r['roofline']['by_kernel'].items() if v['kernel_ms']})
    for o,g in r.get('regimes',{}).items():
        print('   ',o, g['iterations'], round(g['run_to_convergence_iters_per_s'],1),'it/s cold',round(g['cold_no_carry_ms'],2),'conv',round(g['converged_ms'],2), g['per_iter_ms'][:14], g['kernels_ms']['k_screen_quad'][6:14])
PY
