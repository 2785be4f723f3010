#!/bin/bash
# per-kernel MEDIAN duration over a run (the converged regime when most iterations are converged): rocprofv3 kernel trace
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
raw=/tmp/prof_med_$tag; rm -rf $raw; mkdir -p $raw $root/gpurun_out/prof_$tag
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $raw -o t -- python $root/bench.py "$@" > $root/gpurun_out/prof_$tag/bench.log 2>&1
f=$(find $raw -name "*kernel_trace.csv" | head -1)
python - "$f" > $root/gpurun_out/prof_$tag/kernel_median.txt <<'PY'
import csv, sys, statistics as st
from collections import defaultdict
d=defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name'].split('(')[0][:70]
    if 'at::native' in n: continue
    d[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
rows=sorted(d.items(), key=lambda kv:-st.median(kv[1])*len(kv[1]))
print(f"{'kernel':72s} calls  median_us   mean_us")
for n,v in rows:
    if len(v)<5: continue
    print(f"{n:72s} {len(v):5d} {st.median(v):10.1f} {sum(v)/len(v):10.1f}")
for n,v in rows:
    if any(t in n for t in ("k_bounds_steps","k_scatter_by","k_combine_screen","k_copy_i32")): print(n[:30], [round(x) for x in v[:24]])
PY
cat $root/gpurun_out/prof_$tag/kernel_median.txt | head -40
