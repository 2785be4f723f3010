#!/usr/bin/env python3
"""Print the PMC summary of selected kernels: tools/show_pmc.py gpurun_out/prof_<tag> [substr ...]"""
import csv, json, sys
d = sys.argv[1]
subs = sys.argv[2:] or ["k_assign_tile"]
s = json.load(open(f"{d}/pmc_summary.json"))
for k in s:
    if any(x in k for x in subs):
        print(k)
        for c, v in sorted(s[k].items()):
            print(f"   {c:26s} {v['mean_per_dispatch']:.5g}")
try:
    for row in csv.DictReader(open(f"{d}/kernel_stats.csv")):
        if any(x in row["Name"] for x in subs) or row["Name"].startswith(("k_", "void k_")):
            print(f"{row['Name'][:50]:50s} calls={row['Calls']:>4s} avg_us={float(row['AverageNs'])/1e3:10.1f}")
except Exception as e:
    print("no kernel stats", e)
