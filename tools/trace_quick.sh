#!/bin/bash
# quick per-kernel timing at one size: tools/trace_quick.sh LOG2N [extra bench args]
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
LG=${1:-20}; shift
OUT=$REPO/gpurun_out/trace_$LG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 5 --warmup 1 --log2n $LG --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
cat $OUT/*/*_kernel_stats.csv
