#!/usr/bin/env python3
"""Timeline of the 20 timed steps of `bench.py --steps 20 --warmup 5` from a rocprofv3 kernel trace:
tools/timeline20.py <dir> [warmup=5] [steps=20]"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("ofdg::", ""), r.get("Queue_Id", "")) for r in csv.DictReader(open(f)) if "ofdg::" in r["Kernel_Name"] and "pool_synth" not in r["Kernel_Name"])
comp = [k for k in ks if "compose" in k[2]]
first = comp[W]          # first timed compose
# the timed region starts with the first sampler launched after the last warm-up compose ended
t_start = min(k[0] for k in ks if k[0] > comp[W - 1][1])
t_end = comp[W + K - 1][1]
t_end = max(k[1] for k in comp[W:W + K])
print("timed region (first kernel start .. last compose end): %.1f us = %.1f us/step" % ((t_end - t_start) / 1e3, (t_end - t_start) / 1e3 / K))
busy = 0
for k in ks:
    if k[0] >= t_start and k[1] <= t_end + 1:
        print("%8.1f %8.1f  %-28s q%s" % ((k[0] - t_start) / 1e3, (k[1] - t_start) / 1e3, k[2], k[3]))
