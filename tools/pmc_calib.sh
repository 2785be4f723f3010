#!/bin/bash
# FETCH_SIZE calibration in the assignment kernel's own access pattern: with K <= 16 there is one
# centroid tile, so every entry of X is read exactly once per launch (known byte count).
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
raw=/tmp/prof_calib; rm -rf $raw; mkdir -p $raw $root/gpurun_out/calib
cd /tmp
for K in 16 100; do
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_assign_tile" --output-format csv -d $raw/k$K -o pmc -- python $root/bench.py --n-total 2e7 --clusters $K --steps 2 --warmup 1 --cpu-sample 0 > $root/gpurun_out/calib/bench_k$K.log 2>&1
python - <<PY
import csv, glob
rows=[r for f in glob.glob("$raw/k$K/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if r["Counter_Name"]=="FETCH_SIZE"]
v=[float(r["Counter_Value"]) for r in rows]
n=2e7; s=51
xbytes=n*s*10
print(f"K=$K launches={len(v)} FETCH_SIZE mean={sum(v)/len(v):.6g} (x1024 = {sum(v)/len(v)*1024/1e9:.3f} GB); X bytes (f64+u16) = {xbytes/1e9:.3f} GB; ratio raw*1024/X = {sum(v)/len(v)*1024/xbytes:.3f}")
PY
done
