#!/bin/bash
# Usage (GPU box): tools/pmc_fetch.sh <tag> [bench args]  -- one PMC pass: FETCH_SIZE + GRBM_GUI_ACTIVE of our kernels
tag=$1; shift
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
raw=/tmp/pmcf_$tag; rm -rf $raw; mkdir -p $raw
cd /tmp
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-include-regex "k_screen|k_exact" --output-format csv -d $raw -o pmc -- python $root/bench.py "$@" > $raw/bench.log 2>&1
python - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in per.items():
    print("$tag", k, {c: sum(v) / len(v) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
tail -1 $raw/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['roofline']['kernel_ms'])"
