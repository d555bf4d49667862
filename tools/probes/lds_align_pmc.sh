#!/bin/bash
# Calibrates the LDS counters against the microbenchmark (known instruction counts and cycles).
set -u
out=gpurun_out/lds_align_pmc
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/p -- $root/tools/probes/lds_align > $out/run.log 2>&1 || { echo failed; tail -5 $out/run.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/lds_align_pmc/p/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    rows.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
# every configuration is launched twice (10 iterations, then 2000): keep the second
n = 0
for i in ids[1::2]:
    d = rows[i]
    if n < 27:
        print(n, d["k"][:40], {k: int(v) for k, v in d.items() if k != "k"})
    n += 1
PY
