#!/bin/bash
# Runs on the GPU box: one SQ counter pass of a short parse loop per library build named on the
# command line (files under dsp_amd/), for side-by-side instruction counts.
# usage: tools/pmc_ab.sh <lib.so> [<lib.so> ...]
set -u
export TMPDIR=/tmp
root=$(pwd)
for lib in "$@"; do
    out=gpurun_out/pmc_ab/$lib
    rm -rf $out; mkdir -p $out/pmc1
    export MDC_LIB=$lib
    timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $out/pmc1 -- python3 $root/tools/steps.py xml device 1073741824 3 > $out/log.txt 2>&1 || { echo "$lib: pass failed"; tail -5 $out/log.txt; exit 1; }
    echo "== $lib"
    python3 $root/tools/pmc_summary.py $out 2>/dev/null | grep -A9 "tile_kernel<false>\|emit_kernel" | grep -v "^--"
done
