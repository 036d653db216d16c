#!/bin/bash
# one-knob sweeps of the tuning build (make ab NAME=tune EXTRA=-DMSM_TUNING): tools/knob_sweep.sh LOG2N KNOB "v1 v2 ..."
LG=$1; KNOB=$2; VALS=$3
mkdir -p gpurun_out/knobs
for v in $VALS; do
  echo "== $KNOB=$v"
  env $KNOB=$v AB_SERIAL=0 AB_REPS=1 python tools/ab_time.py $LG ab_builds/libmsm_tune.so
done 2>&1 | tee -a gpurun_out/knobs/${KNOB}_$LG.txt
