#!/bin/bash
# window groups at small sizes (tuning build): tools/groups_small.sh "19 20 21"
for lg in ${1:-19 20 21}; do for g in 1 2; do for st in 0 150 400; do
  [ $g = 1 ] && [ $st != 0 ] && continue
  echo "== 2^$lg groups=$g stagger_us=$st"
  MSM_GROUPS=$g MSM_STAGGER_US=$st AB_SERIAL=0 AB_REPS=1 python tools/ab_time.py $lg ab_builds/libmsm_tune.so
done; done; done
