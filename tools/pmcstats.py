#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel: tools/pmcstats.py <dir>"""
import csv, glob, os, sys, collections
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[k]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in acc.items():
    if "ofdg" not in k: continue
    print(k, "n=%d" % len(d["dur_us"]))
    for c, v in sorted(d.items()):
        v = v[len(v) // 4:]  # skip warm-up quarter
        print("   %-24s %14.1f" % (c, sum(v) / len(v)))
