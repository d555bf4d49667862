#!/bin/bash
# Runs on the GPU box: two SQ counter passes of a short parse loop; per-kernel averages.
# usage: tools/pmc_quick.sh [xml|csv]
set -u
export TMPDIR=/tmp
root=$(pwd)
out=gpurun_out/pmc_quick
rm -rf $out; mkdir -p $out/pmc1 $out/pmc2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc1 -- python3 $root/tools/steps.py ${1:-xml} device 1073741824 3 > $out/log1.txt 2>&1 || { echo pass 1 failed; tail -5 $out/log1.txt; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc2 -- python3 $root/tools/steps.py ${1:-xml} device 1073741824 3 > $out/log2.txt 2>&1 || { echo pass 2 failed; tail -5 $out/log2.txt; exit 1; }
python3 $root/tools/pmc_summary.py $out 2>/dev/null | grep -A9 "tile_kernel<false>\|emit_kernel" | grep -v "^--"
