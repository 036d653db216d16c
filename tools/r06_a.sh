#!/bin/bash
# round 6, first GPU call: new table tests, 2^26 timeline, tail-round / finish knobs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_tables.py -m gpu -q -x > gpurun_out/a_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/a_pytest.log
bash tools/timeline.sh 26 > gpurun_out/a_timeline26.txt 2>&1; tail -80 gpurun_out/a_timeline26.txt
for kv in "X=0" "MSM_TAIL_MIN=4194304" "MSM_TAIL_MIN=2097152" "MSM_TAIL_MIN=1" "MSM_TAIL_MIN=1 MSM_FINISH_MAX=2" "MSM_TAIL_MIN=1 MSM_FINISH_MAX=1" "MSM_PBL=8" "MSM_PBL=32" "MSM_TC=16" "MSM_TC=64"; do
  echo "== $kv"
  env $kv AB_SERIAL=0 AB_REPS=1 python3 tools/ab_time.py 26 ab_builds/libmsm_tune.so
done 2>&1 | tee gpurun_out/a_knobs.txt
