for lg in 16 18 20; do
  echo "== lg=$lg"
  for v in "" "MSM_PBL=4" "MSM_PBL=16" "MSM_PBL=2" "MSM_FINISH_MAX=8" "MSM_FINISH_MAX=16" "MSM_FINISH_MAX=64" "MSM_TAIL_MIN=262144" "MSM_TAIL_MIN=1048576" "MSM_TAIL_MIN=8388608" "MSM_TC=2" "MSM_TC=8" "MSM_TC=16"; do
    printf "%-22s " "$v"; env $v AB_REPS=1 python tools/ab_time.py $lg - | python -c "import sys,json; d=json.loads(sys.stdin.read().split(' ',1)[1]); print(round(d['ms'],3), d['phase'])"
  done
done
