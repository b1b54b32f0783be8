#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel: tools/pmcstats.py <dir> [kernel-name-filter]"""
import csv, glob, os, sys, collections
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True))[-1]
flt = sys.argv[2] if len(sys.argv) > 2 else "ofdg"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r and r["Start_Timestamp"]:
        acc[k]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in acc.items():
    if flt not in k: continue
    print(k, "n=%d" % max(len(v) for v in d.values()))
    for c, v in sorted(d.items()):
        v = v[len(v) // 4:]  # skip warm-up quarter
        print("   %-40s %16.1f" % (c, sum(v) / len(v)))
