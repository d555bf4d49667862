#!/bin/bash
# Runs on the GPU box: kernel trace of a short parse loop, per-kernel average durations.
# usage: tools/trace_once.sh [xml|csv]
set -u
export TMPDIR=/tmp
root=$(pwd)
out=gpurun_out/trace_once
rm -rf $out; mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/steps.py ${1:-xml} device 1073741824 8 > $out/log.txt 2>&1 || { echo trace failed; tail -5 $out/log.txt; exit 1; }
find $out -name '*kernel_stats.csv' -exec cat {} \; | cut -d, -f1-4 | head -12
