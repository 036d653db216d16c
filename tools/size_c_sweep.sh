#!/bin/bash
# overlapped MSM time against the window size at several input sizes: tools/size_c_sweep.sh "24 25 27" "16 19 22" [lib]
SIZES=${1:-"24 25"}; CS=${2:-"16 19 22"}; LIB=${3:--}
for lg in $SIZES; do for c in $CS; do
  echo "== 2^$lg c=$c"
  MSM_C=$c AB_SERIAL=0 AB_REPS=1 timeout 600 python tools/ab_time.py $lg $LIB
done; done
