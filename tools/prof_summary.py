#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into one small text/JSON summary."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

raw, out = sys.argv[1], sys.argv[2]
summary = {}
per = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(raw, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            per[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in per.items():
    summary[k] = {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, d2 in [(c, d[c]) for c in d] for v in [d2]}
with open(os.path.join(out, "pmc_summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1, sort_keys=True)
lines = []
for k in sorted(summary):
    lines.append(k)
    for c in sorted(summary[k]):
        lines.append(f"    {c:28s} {summary[k][c]['mean_per_dispatch']:.6g}  (n={summary[k][c]['dispatches']})")
with open(os.path.join(out, "pmc_summary.txt"), "w") as fh:
    fh.write("\n".join(lines) + "\n")
print("\n".join(lines))
