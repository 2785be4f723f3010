#!/bin/bash
# usage: tools/collect_profiles.sh r04 -- copies the measurement set of tools/gpu_final.sh from gpurun_out/ (scratch) into profiles/ (tracked) and rebuilds
# profiles/pmc_latest.json from the PMC passes of tools/prof.sh (FETCH_SIZE x 2 + WRITE_SIZE: MI355X_MICROARCH.md's HBM recipe)
cd "$(dirname "$0")/.."
R=${1:-r05}
export R
for f in headline headline_100steps headline_shuffled headline_eager_stats config5 config3 k10_n2e7 config2_n1e7 shard_1.25e7 shard_1.25e7_100steps shard_2.5e7 shard_5e7; do
  [ -f gpurun_out/${R}final/bench_$f.json ] && cp gpurun_out/${R}final/bench_$f.json profiles/${R}_bench_$f.json
  # (round 6: stdout carries the compact line only; the full result object is the detail file beside it)
  [ -f gpurun_out/${R}final/bench_${f}_detail.json ] && cp gpurun_out/${R}final/bench_${f}_detail.json profiles/${R}_bench_${f}_detail.json
done
cp gpurun_out/prof_${R}_headline/kernel_stats.csv profiles/${R}_headline_kernel_stats.csv
cp gpurun_out/prof_${R}_headline/pmc_summary.txt profiles/${R}_headline_pmc_summary.txt
for t in k10 shuffled config5; do cp gpurun_out/prof_${R}_$t/kernel_stats.csv profiles/${R}_${t}_kernel_stats.csv; done
cp gpurun_out/${R}final/driver_bench_n1e7.txt profiles/${R}_driver_bench_n1e7.txt
cp gpurun_out/timeline_${R}_shard/timeline.txt profiles/${R}_timeline_shard_1.25e7.txt
[ -f gpurun_out/timeline_${R}_shard100/timeline.txt ] && cp gpurun_out/timeline_${R}_shard100/timeline.txt profiles/${R}_timeline_shard_1.25e7_100steps.txt
cp gpurun_out/${R}final/stress_parity.txt profiles/${R}_stress_parity.txt
cp gpurun_out/${R}final/ubench_quad.txt profiles/${R}_ubench_quad.txt
python - <<'PY'
import json, csv
import os
R = os.environ['R']
pm = json.load(open(f'gpurun_out/prof_{R}_headline/pmc_summary.json'))
ks = {r['Name'].split('(')[0]: r for r in csv.DictReader(open(f'gpurun_out/prof_{R}_headline/kernel_stats.csv'))}
n, K, p2, s = 100000000, 100, 1024, 51
b_iter = n * s * 12 + (n + 1) * 8 + n * 12 + 24 * p2 * K
b_acc = n * s * 10 + n * 8 + 16 * p2 * K
recs = []
def rec(kernel, headline_for, alg_bytes, note):
    c = pm.get(kernel)
    if c is None:
        print('no PMC record for', kernel); return
    g = lambda x: c[x]['mean_per_dispatch'] if x in c else None
    t = float(ks[kernel]['AverageNs']) / 1e6 if kernel in ks else None
    r = {"round": int(R[1:]), "n_local": n, "K": K, "p2": p2, "start": "sample", "kernel": kernel.replace('void ', ''),
         "headline_for": headline_for,
         "command": "python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-regimes (tools/prof.sh " + R + "_headline; mean over this kernel's launches)",
         "dispatches": c['FETCH_SIZE']['dispatches'] if 'FETCH_SIZE' in c else None,
         "FETCH_SIZE_raw_KB": g('FETCH_SIZE'), "WRITE_SIZE_raw_KB": g('WRITE_SIZE'),
         "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests at 64 B; profiles/r01_fetch_calibration.txt)",
         "hbm_bytes_per_launch": (g('FETCH_SIZE') * 2 + g('WRITE_SIZE')) * 1024 if g('FETCH_SIZE') is not None and g('WRITE_SIZE') is not None else None,
         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_trace": t, "note": note}
    for k in ('SQ_INSTS_VALU', 'GRBM_GUI_ACTIVE', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES'):
        if g(k) is not None: r[k] = g(k)
    if g('SQ_INSTS_VALU') and g('GRBM_GUI_ACTIVE') and t:
        cyc = g('GRBM_GUI_ACTIVE') / 8
        r["valu_issue_utilization"] = round(g('SQ_INSTS_VALU') / 1024 * 4 / cyc, 3)
        r["effective_clock_ghz"] = round(cyc / (t * 1e-3) / 1e9, 3)
        r["valu_note"] = "SQ_INSTS_VALU / 1024 SIMDs x 4 issue cycles over GRBM_GUI_ACTIVE / 8 XCDs (an overestimate: plain 32-bit ops issue in 2); clock = those cycles / the traced mean duration"
    recs.append(r)
    print(kernel[:60], r['dispatches'], t, 'ms', None if r['hbm_bytes_per_launch'] is None else round(r['hbm_bytes_per_launch'] / 1e9, 2), 'GB', r.get('SQ_INSTS_VALU'), r.get('effective_clock_ghz'))
rec('void k_screen_quad<13, unsigned short, 0, false>', 'k_screen_quad', b_iter,
    "plain form, every 16-point step (a run's first call): SURVEY 8(d) bytes of an iteration; the screen streams its own f32 / u16 copy once per centroid tile, the three tiles of a team meet in their XCD's L2")
rec('void k_screen_quad<13, unsigned short, 2, false>', None, b_iter, "hinted form, late split (7 of 13 rounds for all centroids); all steps on the screen")
rec('void k_screen_quad<13, unsigned short, 1, false>', None, None, "hinted / two-phase form with the early split (3 of 13 rounds); most launches run over a short list of steps")
rec('void k_exact_accumulate_rec<unsigned short, 4, true, false>', 'k_exact_accumulate', b_acc, "full accumulation pass over the record layout, sums only (a lazy run's first call): values + 16-bit row ids once, permutation in; bytes as SURVEY 8(d)'s separate accumulation pass (the upper-bound store it no longer does included)")
rec('void k_exact_accumulate_rec<unsigned short, 4, false, true>', None, b_acc, "distances + statistics on demand (once per run): the same records, no sums")
rec('void k_accumulate_events<unsigned short, true>', None, None, "incremental calls, pair events: the points that changed cluster, each read ONCE (one slab per run of an (old, new) pair, added to the new cluster's rows and subtracted from the old one's); bytes = movers x 512 B")
rec('void k_accumulate_events<unsigned short, false>', None, None, "incremental calls with few movers per pair: two events per mover, each read on its own; bytes = 2 x movers x 512 B")
# launch-weighted mean over every form of the screen kernel in the profiled run (its 25 launches: 5 warm-up + 20 timed)
tot_b = tot_d = 0.0
forms = []
for kname, cc in pm.items():
    if 'k_screen_quad' in kname and 'FETCH_SIZE' in cc and 'WRITE_SIZE' in cc:
        d = cc['FETCH_SIZE']['dispatches']
        b = (cc['FETCH_SIZE']['mean_per_dispatch'] * 2 + cc['WRITE_SIZE']['mean_per_dispatch']) * 1024
        tot_b += d * b; tot_d += d
        forms.append(f"{kname.replace('void ', '')} x{d}")
if tot_d:
    recs.append({"round": int(R[1:]), "n_local": n, "K": K, "p2": p2, "start": "sample", "kernel": "k_screen_quad (all forms)",
                 "window_for": "k_screen_quad", "dispatches": int(tot_d), "hbm_bytes_per_launch": tot_b / tot_d,
                 "note": "launch-weighted mean of FETCH_SIZE x 2 + WRITE_SIZE over every k_screen_quad launch of the profiled bench run (" + "; ".join(forms) + "): the plain, hinted and list forms of a run's first iterations as the timed window mixes them"})
    print('window mean', round(tot_b / tot_d / 1e9, 2), 'GB over', int(tot_d), 'launches')
json.dump(recs, open('profiles/pmc_latest.json', 'w'), indent=1)
PY
ls profiles | grep $R
