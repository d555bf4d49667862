#!/bin/bash
# Runs on the GPU box: kernel trace + two PMC passes of a short bench run.
# usage: tools/profile.sh <tag> [bench args...]
set -u
tag=${1:-r01}; shift || true
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --cpu-sample-mb 0 --verify 0 "$@" > $out/trace.log 2>&1 || { echo trace failed; tail -5 $out/trace.log; exit 1; }
find $out/trace -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc1 -- python3 $root/bench.py --steps 2 --warmup 1 --cpu-sample-mb 0 --verify 0 "$@" > $out/pmc1.log 2>&1 || { echo pmc1 failed; tail -5 $out/pmc1.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/pmc2 -- python3 $root/bench.py --steps 2 --warmup 1 --cpu-sample-mb 0 --verify 0 "$@" > $out/pmc2.log 2>&1 || { echo pmc2 failed; tail -5 $out/pmc2.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc3 -- python3 $root/bench.py --steps 2 --warmup 1 --cpu-sample-mb 0 --verify 0 "$@" > $out/pmc3.log 2>&1 || { echo pmc3 failed; tail -5 $out/pmc3.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc4 -- python3 $root/bench.py --steps 2 --warmup 1 --cpu-sample-mb 0 --verify 0 "$@" > $out/pmc4.log 2>&1 || { echo pmc4 failed; tail -5 $out/pmc4.log; exit 1; }
python3 $root/tools/pmc_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
