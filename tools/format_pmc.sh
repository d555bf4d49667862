#!/bin/bash
# Runs on the GPU box: kernel trace and counter passes of the formatting kernel over
# tools/format_bench.py (args: bytes, maxcount, LDS budget).
set -u
out=gpurun_out/format_pmc
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
pass() { # name, then the rocprofv3 options
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 "$@" --output-format csv -d $out/$name -- python3 $root/tools/format_bench.py "${ARGS[@]}" > $out/$name.log 2>&1 || { echo "$name failed"; tail -5 $out/$name.log; exit 1; }
}
ARGS=("$@")
pass trace --kernel-trace --stats
find $out/trace -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
pass pmc1 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVES
pass pmc2 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES
pass pmc3 --pmc FETCH_SIZE
pass pmc4 --pmc WRITE_SIZE
python3 $root/tools/pmc_summary.py $out > $out/summary.txt 2>&1
grep -v "copyBuffer\|^      SQ.* 0.0$" $out/summary.txt | grep -B1 -A12 "format_kernel\|kernel stats"
