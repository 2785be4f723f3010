#!/bin/bash
# Usage (GPU box): tools/pmc_fetch.sh <tag> [bench args]: FETCH_SIZE / WRITE_SIZE per k_screen_quad dispatch (two passes).
tag=$1; shift
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmcf_$tag; raw=/tmp/pmcf_raw_$tag
rm -rf $raw; mkdir -p $out $raw; cd /tmp
i=0
for grp in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --kernel-include-regex "k_screen_quad" --output-format csv -d $raw/pmc$i -o pmc -- python $root/bench.py --no-pmc --no-regimes --cpu-sample 0 --steps 10 --warmup 1 "$@" > $out/bench_pmc$i.log 2>&1
done
python - <<PY
import csv, glob, collections
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
        f, w = r.get("FETCH_SIZE", 0), r.get("WRITE_SIZE", 0)
        cyc = r.get("GRBM_GUI_ACTIVE", 0) / 8
        fh.write(f"{j:3d} {r.get('name'):50s} fetch x2 {f * 2 * 1024 / 1e9:7.2f} GB  write {w * 1024 / 1e9:6.2f} GB  cycles/XCD {cyc:.4g}\n")
print(open("$out/per_dispatch.txt").read()[:3000])
PY
