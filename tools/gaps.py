#!/usr/bin/env python3
"""Per hardware queue: mean idle gap between consecutive kernels of a rocprofv3 kernel trace (tools/gaps.py <dir>)."""
import csv, glob, os, sys, statistics as st
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("ofdg::", ""), r["Queue_Id"]) for r in rows)
byq = {}
for k in ks:
    byq.setdefault(k[3], []).append(k)
for q, v in byq.items():
    gaps, durs = {}, {}
    for a, b in zip(v, v[1:]):
        gaps.setdefault((a[2], b[2]), []).append((b[0] - a[1]) / 1e3)
    for a in v:
        durs.setdefault(a[2], []).append((a[1] - a[0]) / 1e3)
    print("queue", q, "kernels", len(v))
    for k, x in gaps.items():
        if len(x) > 20:
            print("   gap %-28s -> %-28s %6.1f us  (n=%d)" % (k[0], k[1], st.mean(x), len(x)))
    for k, x in durs.items():
        if len(x) > 20:
            print("   dur %-28s %6.1f us" % (k, st.mean(x)))
