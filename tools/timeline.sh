#!/bin/bash
# Timeline of single Lloyd iterations (GPU box): rocprofv3 kernel trace of a bench run, cut at k_finalize_centers; prints
# for chosen iterations every kernel with its start offset, duration and the idle gap in front of it, and per iteration
# the busy / idle split -- what the launches and the host's per-iteration read cost when the shard is small.
#   tools/timeline.sh TAG [bench.py arguments]        (TIMELINE_SCRIPT=tools/driver_bench.py: another entry point;
#                                                      TIMELINE_SHOW=3,4,7: which iterations of the trace to list)
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
raw=/tmp/timeline_$tag; rm -rf $raw; mkdir -p $raw $root/gpurun_out/timeline_$tag
cd /tmp
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $raw -o t -- python $root/${TIMELINE_SCRIPT:-bench.py} "$@" > $root/gpurun_out/timeline_$tag/bench.log 2>&1
f=$(find $raw -name "*kernel_trace.csv" | head -1)
m=$(find $raw -name "*memory_copy_trace.csv" | head -1)
python - "$f" "$m" > $root/gpurun_out/timeline_$tag/timeline.txt <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:64]))
try:
    for r in csv.DictReader(open(sys.argv[2])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'memcpy ' + r.get('Direction', '')))
except Exception as e:
    print('no memcpy trace', e)
rows.sort()
# iterations: from the first kernel after a k_finalize_centers to the next k_finalize_centers
its, cur = [], []
for s, e, n in rows:
    cur.append((s, e, n))
    if n.startswith('k_finalize_centers'):
        its.append(cur)
        cur = []
print(len(its), 'iterations in the trace')
summ = []
for j, it in enumerate(its):
    t0, t1 = it[0][0], it[-1][1]
    busy = sum(e - s for s, e, _ in it)
    prev_end = its[j - 1][-1][1] if j else t0
    summ.append((j, (t1 - prev_end) / 1e3, busy / 1e3, len(it)))
print('iter  wall_us(from previous finalize)  busy_us  launches')
for j, w, b, c in summ[:80]:
    print(f'{j:4d} {w:10.1f} {b:10.1f} {c:4d}')
def show(j):
    it = its[j]
    prev_end = its[j - 1][-1][1] if j else it[0][0]
    print(f'--- iteration {j}: {len(it)} launches')
    last = prev_end
    for s, e, n in it:
        print(f'  +{(s - prev_end) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - last) / 1e3:7.1f}  {n}')
        last = max(last, e)
if its:
    # the bench's timed region follows the warm-up: show a cold-ish and a converged iteration from the last run in the trace
    import os
    pick = [int(v) for v in os.environ.get('TIMELINE_SHOW', '').split(',') if v] or [max(0, len(its) - 20), max(0, len(its) - 18), len(its) - 3]
    for j in sorted(set(j for j in pick if 0 <= j < len(its))):
        show(j)
PY
head -150 $root/gpurun_out/timeline_$tag/timeline.txt
