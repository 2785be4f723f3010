#!/bin/bash
# round 2, first GPU call: new driver tests, whole -m gpu suite, regime-explicit bench, K=10 kernel trace
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2a; mkdir -p $out
cd $root
( time timeout 900 python -m pytest tests/test_gpu_driver_fastpath.py tests/test_known_answers.py -m gpu -x -q ) > $out/new_tests.log 2>&1
tail -5 $out/new_tests.log
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $out/all_tests.log 2>&1
tail -5 $out/all_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
tail -c 3000 $out/bench.json; tail -5 $out/bench.err
PROF_TRACE_ONLY=1 timeout 600 bash tools/prof.sh r2a_k10 --n-total 2e7 --clusters 10 --steps 10 --warmup 2 --no-regimes --cpu-sample 0
cp gpurun_out/prof_r2a_k10/kernel_stats.csv $out/k10_kernel_stats.csv
tail -3 gpurun_out/prof_r2a_k10/bench_trace.log | cut -c1-1500
