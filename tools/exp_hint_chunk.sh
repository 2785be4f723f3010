#!/bin/bash
# Round-6 experiment (the cheap variant): smaller chunks in the two-phase screen launches, so that the three
# workgroups of a team stay within the XCD's 4 MB of L2.  Per chunk size: the 20-step headline window (timed twice) and the
# FETCH_SIZE of every k_screen_quad dispatch of a 10-step run.   usage (GPU box): tools/exp_hint_chunk.sh
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/exp_hint_chunk; mkdir -p $out
cd $root
for c in 0 1024 512 256; do
  export SPKM_X_HINT_CHUNK=$c
  for rep in 1 2; do
    SPKM_BENCH_DUMP=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-regimes --cpu-sample 0 --detail-out /dev/null > $out/line_${c}_$rep.json 2> $out/err_${c}_$rep.txt
    python - <<PY
import json,re
d=json.loads(open("$out/line_${c}_$rep.json").read().strip().splitlines()[-1])
err=open("$out/err_${c}_$rep.txt").read()
m=re.search(r"per-call ms: \[([^\]]*)\]", err)
ks=[float(x) for x in m.group(1).split(",")] if m else []
print("chunk $c rep $rep:", round(d["value"],2), "it/s; screen ms per call:", [round(x,1) for x in ks[0::2][:10]])
PY
  done
  raw=/tmp/pmc_hc_$c; rm -rf $raw
  (cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_screen_quad" --output-format csv -d $raw -o pmc -- python $root/bench.py --no-pmc --no-regimes --cpu-sample 0 --steps 10 --warmup 1 --detail-out /dev/null > $out/pmc_$c.log 2>&1)
  python - <<PY
import csv,glob
rows=[]
for f in glob.glob("$raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="FETCH_SIZE": rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-28:], float(r["Counter_Value"])*2*1024/1e9))
rows.sort()
print("chunk $c FETCH_SIZE x2 per k_screen_quad dispatch (GB):", [(n.split("short, ")[-1], round(g,1)) for _,n,g in rows])
PY
done 2>&1 | tee $out/summary.txt
