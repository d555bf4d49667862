#!/bin/bash
# Instruction counts per phase: the ablation build under rocprofv3 counters, one run per skip mask.
set -u
out=gpurun_out/ablate_pmc
mkdir -p $out
export TMPDIR=/tmp
export MDC_ABLATE_CHILD=1
root=$(pwd)
for skip in ${@:-0 1 2 3}; do
  export MDC_DEBUG_SKIP=$skip
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM --output-format csv -d $out/skip$skip -- python3 $root/tools/ablate.py > $out/skip$skip.log 2>&1 || { echo "skip $skip failed"; tail -5 $out/skip$skip.log; exit 1; }
  mkdir -p $out/s$skip; rm -rf $out/s$skip/pmc1; mv $out/skip$skip $out/s$skip/pmc1
  echo "=== skip=$skip"; python3 $root/tools/pmc_summary.py $out/s$skip | grep -A9 "tile_kernel"
done
