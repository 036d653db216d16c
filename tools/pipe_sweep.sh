# range splits of the pipelined host-scalar msm_run (tuning build): MSM_PIPE_64 = where the ranges end, in 64ths of the points
# (MSM_PIPE_SH = ends at n >> shift is the older knob): tools/pipe_sweep.sh [LOG2N] ["4,12 4,13 ..."]
LIST=${2:-"4,12 4,13 4,14 3,12 5,13 4,11 6,16 2,10"}
for sh in $LIST; do echo "== MSM_PIPE_64=$sh"; MSM_PIPE_64=$sh MSM_HIP_LIB=$PWD/ab_builds/libmsm_tune.so timeout 300 python3 tools/host_scalars_time.py ${1:-26} 2>&1 | tail -2; done
