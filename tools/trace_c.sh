#!/bin/bash
# per-kernel timing of ONE serialised MSM at a forced window size: tools/trace_c.sh LOG2N C
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
LG=${1:-26}; C=${2:-16}
OUT=$REPO/gpurun_out/trace_c$C; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/tools/run_once.py $LG $C > $OUT/log.txt 2>&1
cat $OUT/log.txt | tail -2
python3 $REPO/tools/kstats.py $OUT/*/*_kernel_stats.csv | head -${3:-14}
