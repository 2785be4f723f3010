#!/bin/bash
# Round 6: the headline benchmark away from the point its policy thresholds were tuned on: other noise
# levels, cluster counts and dimensions at N = 1e8 (K = 200: 5e7).  One line per run: first-20 window and whole runs.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out/sweep_r6; mkdir -p $out
run() {
  tag=$1; shift
  timeout 900 python bench.py --steps 20 --warmup 5 --no-pmc --cpu-sample 0 --detail-out $out/$tag.json "$@" > $out/$tag.line 2> $out/$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$out/$tag.json")); r=d.get("regimes",{})
    print("$tag:", round(d["value"],1), "it/s first 20;", " ".join(f"{k} {v['iterations']} it {v['ended_by']} {v['whole_run_iters_per_s']:.1f} it/s" for k,v in r.items()),
          "| cold", round(d["roofline"]["kernel_ms"],2), "ms frac", round(d["roofline"]["frac"] or 0,3), "| listed last", d["config"]["uncertified_points_last_iter"])
except Exception as e:
    print("$tag: failed", e, open("$out/$tag.err").read()[-300:])
PY
}
run noise0.3 --noise 0.3
run noise0.6 --noise 0.6
run noise1.0 --noise 1.0
run K50 --clusters 50
run K200_n5e7 --clusters 200 --n-total 5e7
run K37 --clusters 37
run d512 --dim 512
run d2048_n5e7 --dim 2048 --n-total 5e7
run planted --start planted
run sparsity0.1_n5e7 --sparsity 0.1 --n-total 5e7
