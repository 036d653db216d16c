#!/bin/bash
# accumulate / total time against the window size c at one input size: tools/csweep.sh LOG2N "c1 c2 ..." [lib]
LG=${1:-26}; CS=${2:-"16 17 18 19 20 22"}; LIB=${3:--}
mkdir -p gpurun_out/csweep
for c in $CS; do
  echo "== c=$c"
  MSM_C=$c AB_SERIAL=1 AB_REPS=1 python tools/ab_time.py $LG $LIB
  MSM_C=$c AB_SERIAL=0 AB_REPS=1 python tools/ab_time.py $LG $LIB
done 2>&1 | tee gpurun_out/csweep/c_$LG.txt
