#!/bin/bash
# round 6: GPU check of a work-in-progress build -- new tests first, then the whole suite, then a short headline run
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_skew.py tests/test_gpu_operators.py tests/test_gpu_tables.py -m gpu -q -x > gpurun_out/b_new.log 2>&1; echo "new tests rc=$?"; tail -15 gpurun_out/b_new.log
timeout 2400 python3 -m pytest tests -m gpu -q -x --deselect tests/test_gpu_skew.py > gpurun_out/b_all.log 2>&1; echo "all rc=$?"; tail -8 gpurun_out/b_all.log
timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-pcie > gpurun_out/b_bench.json 2> gpurun_out/b_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/b_bench.err
python3 - <<'P'
import json
try:
    b=json.loads(open('gpurun_out/b_bench.json').read().strip().splitlines()[-1])
    x=b['roofline']['exclusive']
    print('ms_per_step', b['ms_per_step'], 'median', b['median_ms'], 'verified', b['verified'])
    print('exclusive phases', x['phase_ms']); print('overlapped phases', b['phase_ms'])
except Exception as e: print('no bench line', e)
P
