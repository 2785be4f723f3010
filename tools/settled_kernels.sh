#!/bin/bash
# GPU box: per-kernel mean duration over the last 15 iterations of a settled lazy run (tools/settled_probe.py under
# rocprofv3 --kernel-trace).   tools/settled_kernels.sh TAG [N] [iters] [order]      (environment passes through)
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
raw=/tmp/sk_$tag; rm -rf $raw; mkdir -p $raw
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $raw -o t -- python $root/tools/settled_probe.py "$@" > $raw/probe.log 2>&1
f=$(find $raw -name "*kernel_trace.csv" | head -1)
echo "== $tag"
tail -4 $raw/probe.log
python - "$f" <<'PY'
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60]))
rows.sort()
its, cur = [], []
for s, e, n in rows:
    cur.append((s, e, n))
    if n.startswith('k_finalize_centers'):
        its.append(cur); cur = []
last = its[-15:]
acc = collections.OrderedDict()
for it in last:
    for s, e, n in it:
        acc.setdefault(n, []).append((e - s) / 1e3)
wall = [(b[-1][1] - a[-1][1]) / 1e3 for a, b in zip(last, last[1:])]
print(f'wall per iteration (finalize to finalize): {sum(wall) / len(wall):.1f} us')
busy = 0
for n, v in acc.items():
    m = sum(v) / len(last)
    busy += m
    print(f'  {m:8.1f} us  x{len(v) / len(last):.1f}  {n}')
print(f'  busy {busy:.1f} us')
PY
