#!/bin/bash
# chunk-order geometry at a big window (tuning builds: make ab NAME=t4k EXTRA="-DMSM_TUNING -DMSM_CO_MAX_KEYS=129"):
# tools/chunk_sweep.sh LOG2N C "libs" "chunk logs"
LG=${1:-26}; C=${2:-22}
LIBS=${3:-ab_builds/libmsm_t4k.so ab_builds/libmsm_t8k.so}; CLS=${4:-22 21 20}
for lib in $LIBS; do for cl in $CLS; do
  echo "== $lib chunk_log=$cl"
  MSM_CHUNK_LOG=$cl MSM_C=$C AB_SERIAL=1 AB_REPS=1 timeout 600 python tools/ab_time.py $LG $lib
done; done
