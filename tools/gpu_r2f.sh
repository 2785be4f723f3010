#!/bin/bash
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2f; mkdir -p $out
cd $root
( time timeout 1200 python -m pytest tests/test_gpu_config5.py -m gpu -x -q -s ) > $out/config5_test.log 2>&1
grep -E "config 5|passed|failed|Error|error" $out/config5_test.log | head -20
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $out/all_tests.log 2>&1
tail -4 $out/all_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_headline.json 2> $out/bench_headline.err; tail -2 $out/bench_headline.err
timeout 900 python bench.py --workload config5 --steps 20 --warmup 5 --cpu-sample 0 > $out/bench_config5.json 2> $out/bench_config5.err; tail -2 $out/bench_config5.err
timeout 600 python bench.py --workload config3 --steps 100 --warmup 5 --cpu-sample 0 > $out/bench_config3.json 2> $out/bench_config3.err; tail -2 $out/bench_config3.err
timeout 600 python bench.py --n-total 2e7 --clusters 10 --steps 20 --warmup 5 --cpu-sample 0 > $out/bench_k10_n2e7.json 2> $out/bench_k10.err; tail -2 $out/bench_k10.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out/r2f/bench_*.json'))):
    try:
        r=json.load(open(f))
    except Exception as e:
        print(f, 'unreadable', e); continue
    print(os.path.basename(f), round(r['value'],2),'it/s', round(r['ms_per_step'],3),'ms', r['roofline']['kernel'], r['roofline']['frac'], {k:(round(v['kernel_ms'],3), round(v['frac'],3)) for k,v in r['roofline']['by_kernel'].items() if v['kernel_ms']})
    for o,g in r.get('regimes',{}).items():
        print('   ',o, g['iterations'], round(g['run_to_convergence_iters_per_s'],1),'it/s cold',round(g['cold_no_carry_ms'],2),'conv',round(g['converged_ms'],2))
    if 'ingest' in r['config']: print('   ingest', r['config']['ingest']['GBs'])
PY
