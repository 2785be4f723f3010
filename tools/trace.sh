#!/bin/bash
# kernel-trace + stats only: tools/trace.sh <tag> [bench args]  -> gpurun_out/trace_<tag>/kernel_stats.csv
tag=$1; shift
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/trace_$tag; raw=/tmp/trace_raw_$tag
rm -rf $raw; mkdir -p $out $raw; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $raw -o t -- python $root/bench.py "$@" > $out/bench.log 2>&1
find $raw -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
python - <<PY
import csv
for r in csv.DictReader(open("$out/kernel_stats.csv")):
    if r["Name"].startswith(("k_","void k_")): print(f"{r['Name'].split('(')[0][:48]:48s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:10.1f}")
PY
tail -1 $out/bench.log | cut -c1-330
