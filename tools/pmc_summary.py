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

# HBM traffic of the dominant kernel per launch: FETCH_SIZE and WRITE_SIZE count KB; on gfx950
# FETCH_SIZE tallies the 128-byte read requests of wide streaming loads at 64 bytes, so it is
# doubled.  Written as JSON so that bench.py can quote it next to the algorithmic bytes.
import json

def _avg(passdir, counter, kernel_sub):
    files = glob.glob(os.path.join(out, passdir, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return None
    # one row per dispatch and counter instance: sum the instances of a dispatch, average the dispatches
    per = defaultdict(float)
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] == counter and kernel_sub in row["Kernel_Name"]:
            per[row["Dispatch_Id"]] += float(row["Counter_Value"])
    return sum(per.values()) / len(per) if per else None

# Both kernels of a parse: {"sizing": the tile kernel's sizing instance, "write": the emit kernel}.
# usage: pmc_summary.py <dir> [sizing kernel substring] [write kernel substring]
subs = {"sizing": sys.argv[2] if len(sys.argv) > 2 else "tile_kernel<0>", "write": sys.argv[3] if len(sys.argv) > 3 else "emit_kernel"}
kernels = {}
for role, sub in subs.items():
    f, w = _avg("pmc3", "FETCH_SIZE", sub), _avg("pmc4", "WRITE_SIZE", sub)
    if f is not None and w is not None:
        kernels[role] = {"kernel": sub, "fetch_bytes": 2 * f * 1024, "write_bytes": w * 1024, "traffic_bytes": 2 * f * 1024 + w * 1024}
if kernels:
    t = {"kernels": kernels,
         "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs; KB units; FETCH_SIZE doubled (gfx950)"}
    # the batch the profiled bench run parsed (its JSON line in trace.log)
    try:
        for line in open(os.path.join(out, "trace.log")):
            if line.startswith("{") and "bytes_per_gpu" in line:
                t["workload_bytes"] = json.loads(line)["config"]["bytes_per_gpu"]
    except (OSError, ValueError, KeyError):
        pass
    t["source"] = out
    json.dump(t, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print("== traffic per launch:", json.dumps(t))
