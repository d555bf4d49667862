#!/bin/bash
# Counters of the formatting kernel: two rocprofv3 --pmc passes over tools/format_bench.py.
set -u
out=gpurun_out/format_pmc
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $out/pmc1 -- python3 $root/tools/format_bench.py "$@" > $out/pmc1.log 2>&1 || { echo "pass 1 failed"; tail -5 $out/pmc1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc2 -- python3 $root/tools/format_bench.py "$@" > $out/pmc2.log 2>&1 || { echo "pass 2 failed"; tail -5 $out/pmc2.log; exit 1; }
python3 $root/tools/pmc_summary.py $out | grep -A12 "format_kernel"
