"""Summarises rocprofv3 csv output (kernel stats + PMC passes) per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
ks = os.path.join(out, "kernel_stats.csv")
if os.path.exists(ks):
    print("== kernel stats (rocprofv3 --kernel-trace --stats)")
    for row in csv.DictReader(open(ks)):
        print("  {Name:60.60s} calls={Calls:>5s} avg_ns={AverageNs:>12s} total_ns={TotalDurationNs:>14s} pct={Percentage}".format(**row))
for p in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(p):
        continue
    files = glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"][:48]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
    print("==", os.path.basename(p), "(per-dispatch averages)")
    for k in agg:
        print("  ", k)
        for c in sorted(agg[k]):
            print("      %-24s %16.1f" % (c, agg[k][c] / cnt[k][c]))
