#!/bin/bash
# A/B timing of ab_builds/libmsm_<name>.so against the in-tree build: tools/ab_run.sh OUTTAG LOG2N name1 name2 ...
TAG=$1; LG=$2; shift 2
mkdir -p gpurun_out/$TAG
LIBS="-"; for n in "$@"; do LIBS="$LIBS ab_builds/libmsm_$n.so"; done
{ echo "== serial"; AB_SERIAL=1 AB_REPS=${AB_REPS:-1} python tools/ab_time.py $LG $LIBS; echo "== overlapped"; AB_SERIAL=0 AB_REPS=${AB_REPS:-2} python tools/ab_time.py $LG $LIBS; } > gpurun_out/$TAG/ab.txt 2>&1
cat gpurun_out/$TAG/ab.txt
