#!/bin/bash
# Measurement set of a round (usage: tools/gpu_final.sh r04): bench lines for the headline and the other configs, rocprofv3 kernel traces + PMC passes, the
# shard timeline.  Everything lands under gpurun_out/<round>final (tools/collect_profiles.sh <round> copies the summaries to profiles/).
R=${1:-r05}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${R}final; mkdir -p $out
cd $root
timeout 900 python bench.py --steps 20 --warmup 5 --detail-out $out/bench_headline_detail.json > $out/bench_headline.json 2> $out/bench_headline.err
timeout 900 python bench.py --steps 100 --warmup 5 --cpu-sample 0 --no-regimes --detail-out $out/bench_headline_100steps_detail.json > $out/bench_headline_100steps.json 2>> $out/bench_headline.err
timeout 900 python bench.py --order shuffled --steps 20 --warmup 5 --cpu-sample 0 --no-regimes --detail-out $out/bench_headline_shuffled_detail.json > $out/bench_headline_shuffled.json 2>> $out/bench_headline.err
SPKM_BENCH_EAGER_STATS=1 timeout 900 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-regimes --detail-out $out/bench_headline_eager_stats_detail.json > $out/bench_headline_eager_stats.json 2>> $out/bench_headline.err
timeout 900 python bench.py --workload config5 --steps 20 --warmup 5 --cpu-sample 0 --detail-out $out/bench_config5_detail.json > $out/bench_config5.json 2> $out/bench_config5.err
timeout 600 python bench.py --workload config3 --steps 100 --warmup 5 --cpu-sample 0 --detail-out $out/bench_config3_detail.json > $out/bench_config3.json 2> $out/bench_config3.err
timeout 600 python bench.py --n-total 2e7 --clusters 10 --steps 20 --warmup 5 --cpu-sample 0 --detail-out $out/bench_k10_n2e7_detail.json > $out/bench_k10_n2e7.json 2> $out/bench_k10.err
timeout 600 python bench.py --n-total 1e7 --steps 20 --warmup 5 --cpu-sample 0 --detail-out $out/bench_config2_n1e7_detail.json > $out/bench_config2_n1e7.json 2> $out/bench_config2.err
for f in 1.25e7 2.5e7 5e7; do timeout 600 python bench.py --n-total $f --steps 20 --warmup 5 --cpu-sample 0 --no-regimes --detail-out $out/bench_shard_${f}_detail.json > $out/bench_shard_$f.json 2>> $out/bench_shards.err; done
for f in 1.25e7; do timeout 600 python bench.py --n-total $f --steps 100 --warmup 5 --cpu-sample 0 --no-regimes --detail-out $out/bench_shard_${f}_100steps_detail.json > $out/bench_shard_${f}_100steps.json 2>> $out/bench_shards.err; done
timeout 900 python tools/driver_bench.py 1e7 100 100 2>&1 | grep -v Warn | tail -3 > $out/driver_bench_n1e7.txt
bash tools/prof.sh ${R}_headline --steps 20 --warmup 5 --cpu-sample 0 --no-regimes > $out/prof_headline.log 2>&1
PROF_TRACE_ONLY=1 bash tools/prof.sh ${R}_k10 --n-total 2e7 --clusters 10 --steps 20 --warmup 5 --cpu-sample 0 --no-regimes > $out/prof_k10.log 2>&1
PROF_TRACE_ONLY=1 bash tools/prof.sh ${R}_shuffled --order shuffled --steps 20 --warmup 5 --cpu-sample 0 --no-regimes > $out/prof_shuffled.log 2>&1
PROF_TRACE_ONLY=1 bash tools/prof.sh ${R}_config5 --workload config5 --steps 20 --warmup 5 --cpu-sample 0 --no-regimes > $out/prof_config5.log 2>&1
for seed in 21 22 23 24; do timeout 900 python tools/stress_parity.py 240 $seed 2>&1 | tail -1 | sed "s/^/seed $seed: /"; done > $out/stress_parity.txt
./tools/ubench_quad > $out/ubench_quad.txt 2>&1
bash tools/timeline.sh ${R}_shard --n-total 1.25e7 --steps 30 --warmup 2 --cpu-sample 0 --no-regimes > /dev/null 2>&1
TIMELINE_SHOW=60,90 bash tools/timeline.sh ${R}_shard100 --n-total 1.25e7 --steps 100 --warmup 2 --cpu-sample 0 --no-regimes > /dev/null 2>&1
SPKM_ROUND=$R python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out/'+os.environ.get('SPKM_ROUND','r04')+'final/bench_*_detail.json'))):
    try: r=json.load(open(f))
    except Exception as e: print(f,'unreadable'); continue
    print(os.path.basename(f), round(r['value'],2),'it/s', round(r['ms_per_step'],3),'ms', r['roofline']['kernel'], round(r['roofline']['frac'] or 0,3), r['config'].get('hbm_resident_GB'), {k:(round(v['kernel_ms'],3), round(v['frac'],3)) for k,v in r['roofline']['by_kernel'].items() if v['kernel_ms']})
    for o,g in r.get('regimes',{}).items():
        print('   ',o, g['iterations'], g['ended_by'], round(g['whole_run_iters_per_s'],1),'it/s cold',round(g['cold_no_carry_ms'],2),'conv',round(g['converged_ms'],2), 'whole_iter_cold frac', round(g['whole_iter_cold']['frac'],3))
    if 'ingest' in r['config']: print('   ingest', r['config']['ingest']['GBs'])
PY
