#!/bin/bash
# A/B on ONE box: bench.py on the library in the tree (B) against another build (A = $1, e.g. tools/ab/libspkm_r05.so),
# alternating, PAIRS pairs.   usage: tools/ab_bench.sh <A.so> <tag> [bench args...]
A=$1; tag=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ab_$tag; mkdir -p $out
for i in $(seq 1 ${PAIRS:-2}); do
  for v in A B; do
    if [ $v = A ]; then export SPKM_AB_LIB=$root/$A; else unset SPKM_AB_LIB; fi
    timeout 600 python $root/bench.py --no-pmc --cpu-sample 0 "$@" --detail-out $out/detail_${v}_$i.json > $out/line_${v}_$i.json 2> $out/err_${v}_$i.txt
    python - <<PY
import json
try:
    d=json.load(open("$out/detail_${v}_$i.json"))
    r=d.get("regimes",{})
    print("$v$i", round(d["value"],2), "it/s", "full-work ms", round(d["roofline"]["kernel_ms"],3), "window screen ms", round(d["roofline"]["window"]["kernel_ms_mean"],3),
          {k:round(v["whole_run_iters_per_s"],1) for k,v in r.items()}, {k:[round(x,1) for x in v["per_iter_ms"][:9]] for k,v in r.items()})
except Exception as e:
    print("$v$i failed", e)
PY
  done
done
