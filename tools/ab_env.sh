#!/bin/bash
# A/B of one environment switch on ONE box, alternating: tools/ab_env.sh VAR A_VALUE B_VALUE PAIRS [bench args]
var=$1; a=$2; b=$3; pairs=$4; shift 4
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for i in $(seq 1 $pairs); do
  for v in "$a" "$b"; do
    export $var=$v
    timeout 400 python bench.py --no-pmc --cpu-sample 0 --detail-out /tmp/ab_env_detail.json "$@" > /tmp/ab_env_line.json 2> /tmp/ab_env_err.txt
    python - <<PY
import json
d=json.load(open("/tmp/ab_env_detail.json"))
r=d.get("regimes",{})
print("$var=$v pair $i:", round(d["value"],2), "it/s", {k:round(x["whole_run_iters_per_s"],1) for k,x in r.items()})
PY
  done
done
