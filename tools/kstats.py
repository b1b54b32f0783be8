#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: tools/kstats.py <dir-or-csv>"""
import csv, glob, os, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[-1]
for r in csv.DictReader(open(f)):
    print("%-44s calls=%-5s avg_us=%8.1f min=%8.1f max=%8.1f" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                 float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
