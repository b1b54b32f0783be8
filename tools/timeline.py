#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 kernel trace: tools/timeline.py <dir> — compose duration, gap and period,
and one window of consecutive steps."""
import csv, glob, os, sys, statistics as st
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("ofdg::", "")) for r in csv.DictReader(open(f)))
comp = [k for k in ks if "compose" in k[2]]
n0 = len(comp) // 3
durs = [(c[1] - c[0]) / 1e3 for c in comp[n0:]]
gaps = [(comp[i + 1][0] - comp[i][1]) / 1e3 for i in range(n0, len(comp) - 1)]
per = [(comp[i + 1][0] - comp[i][0]) / 1e3 for i in range(n0, len(comp) - 1)]
print("compose: duration %.1f us, gap to the next %.1f us, period %.1f us" % (st.mean(durs), st.mean(gaps), st.mean(per)))
t0 = comp[n0 + 10][0]
for k in ks:
    if comp[n0 + 10][0] - 1000 <= k[0] <= comp[n0 + 12][1]:
        print("%8.1f %8.1f  %s" % ((k[0] - t0) / 1e3, (k[1] - t0) / 1e3, k[2]))
