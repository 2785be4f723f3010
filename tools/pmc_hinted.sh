#!/bin/bash
# Usage (GPU box): tools/pmc_hinted.sh <tag> [bench args]: PMC passes over k_screen_quad, PER DISPATCH (the hinted launches of a
# run's iterations 2-8 one by one): issue / wait / memory-pipeline counters, to see what bounds the early-finished steps.
tag=$1; shift
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmch_$tag; raw=/tmp/pmch_raw_$tag
rm -rf $raw; mkdir -p $out $raw; cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --kernel-include-regex "k_screen_quad" --output-format csv -d $raw/pmc$i -o pmc -- python $root/bench.py --no-pmc --no-regimes --cpu-sample 0 --steps 10 --warmup 1 "$@" > $out/bench_pmc$i.log 2>&1
done
python - <<PY
import csv, glob, collections
per = collections.defaultdict(dict)
for f in glob.glob("$raw/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        per[(int(row["Dispatch_Id"]), row["Kernel_Name"].split("(")[0][:48])][row["Counter_Name"]] = float(row["Counter_Value"])
# dispatch ids differ between passes only by a constant if the launch sequence is identical: group by order of appearance per pass
rows = collections.defaultdict(dict)
for f in sorted(glob.glob("$raw/pmc*/**/*counter_collection.csv", recursive=True)):
    seq = collections.OrderedDict()
    for row in csv.DictReader(open(f)):
        seq.setdefault(int(row["Dispatch_Id"]), {})[row["Counter_Name"]] = (float(row["Counter_Value"]), row["Kernel_Name"].split("(")[0][:48])
    for j, (did, d) in enumerate(sorted(seq.items())):
        for c, (v, name) in d.items():
            rows[j][c] = v; rows[j]["name"] = name
with open("$out/per_dispatch.txt", "w") as fh:
    for j in sorted(rows):
        r = rows[j]
        fh.write(f"{j:3d} {r.get('name')}\n    " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(r.items()) if k != "name") + "\n")
print(open("$out/per_dispatch.txt").read()[:6000])
PY
