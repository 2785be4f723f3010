#!/bin/bash
# GPU box: block-ordered headline data with and without the regrouping (SPKM_NO_REGROUP=1), same box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/abrg
for v in 0 1; do
  export SPKM_NO_REGROUP=$v
  timeout 300 python tools/settled_probe.py 1e8 60 block > gpurun_out/abrg/probe_$v.txt 2>&1
  timeout 400 python bench.py --steps 20 --warmup 5 --no-pmc --cpu-sample 0 --no-regimes > gpurun_out/abrg/bench20_$v.json 2> gpurun_out/abrg/bench20_$v.err
  timeout 400 python bench.py --steps 100 --warmup 5 --no-pmc --cpu-sample 0 --no-regimes > gpurun_out/abrg/bench100_$v.json 2> gpurun_out/abrg/bench100_$v.err
done
for v in 0 1; do echo "== NO_REGROUP=$v"; python - <<P
import json
for f in ("bench20_$v","bench100_$v"):
    try:
        b=json.loads(open("gpurun_out/abrg/%s.json"%f).read().strip().splitlines()[-1]); print(f, b["value"], b["ms_per_step"])
    except Exception as e: print(f, "failed", e)
P
awk 'NR>1{print $1, $2}' gpurun_out/abrg/probe_$v.txt | tr '\n' ';' | cut -c1-1500; echo; done
