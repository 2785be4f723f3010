#!/bin/bash
# copies the measurement set of tools/gpu_r2_final.sh (+ prof_median / timeline outputs) from gpurun_out/ into profiles/
cd "$(dirname "$0")/.."
for f in headline headline_100steps headline_shuffled config5 config3 k10_n2e7 config2_n1e7 shard_1.25e7 shard_2.5e7 shard_5e7; do cp gpurun_out/r2final/bench_$f.json profiles/r02_bench_$f.json; done
cp gpurun_out/prof_r02_headline/kernel_stats.csv profiles/r02_headline_kernel_stats.csv
cp gpurun_out/prof_r02_headline/pmc_summary.txt profiles/r02_headline_pmc_summary.txt
for t in k10 shuffled config5; do cp gpurun_out/prof_r02_$t/kernel_stats.csv profiles/r02_${t}_kernel_stats.csv; done
cp gpurun_out/r2final/kernel_median.txt profiles/r02_converged_kernel_median.txt
cp gpurun_out/r2final/driver_bench_n1e7.txt profiles/r02_driver_bench_n1e7.txt
[ -f gpurun_out/timeline_${1:-shard8f}/timeline.txt ] && cp gpurun_out/timeline_${1:-shard8f}/timeline.txt profiles/r02_timeline_shard_1.25e7.txt
python - <<'PY'
import json, csv
root='./'
pm=json.load(open(root+'gpurun_out/prof_r02_headline/pmc_summary.json'))
ks={r['Name'].split('(')[0]:r for r in csv.DictReader(open(root+'gpurun_out/prof_r02_headline/kernel_stats.csv'))}
def ms(prefix):
    for k,r in ks.items():
        if k.startswith(prefix): return float(r['AverageNs'])/1e6, int(r['Calls'])
    return None, 0
L=json.load(open(root+'profiles/pmc_latest.json'))
for rec in L:
    nm='void '+rec['kernel']
    cands=[k for k in pm if k.startswith(nm.split('>')[0])] if nm not in pm else [nm]
    if nm not in pm:
        print('no PMC record for', nm, '->', cands); continue
    c=pm[nm]; g=lambda x: c[x]['mean_per_dispatch']
    t,calls=ms(nm)
    rec['FETCH_SIZE_raw_KB']=round(g('FETCH_SIZE'),1); rec['WRITE_SIZE_raw_KB']=round(g('WRITE_SIZE'),1)
    rec['hbm_bytes_per_launch']=(g('FETCH_SIZE')*2+g('WRITE_SIZE'))*1024
    if t: rec['kernel_ms_trace']=round(t,4)
    rec['GRBM_GUI_ACTIVE']=g('GRBM_GUI_ACTIVE')
    if 'SQ_INSTS_VALU' in rec: rec['SQ_INSTS_VALU']=g('SQ_INSTS_VALU')
    if 'SQ_LDS_IDX_ACTIVE' in rec: rec['SQ_LDS_IDX_ACTIVE']=g('SQ_LDS_IDX_ACTIVE'); rec['SQ_LDS_BANK_CONFLICT']=g('SQ_LDS_BANK_CONFLICT')
    if 'valu_issue_utilization' in rec and t:
        cyc=g('GRBM_GUI_ACTIVE')/8
        rec['valu_issue_utilization']=round(g('SQ_INSTS_VALU')/1024*4/cyc,3)
        rec['effective_clock_ghz']=round(cyc/(t*1e-3)/1e9,3)
    print(rec['kernel'][:50], calls, t, 'ms', round(rec['hbm_bytes_per_launch']/1e9,2),'GB')
json.dump(L, open(root+'profiles/pmc_latest.json','w'), indent=1)
PY
