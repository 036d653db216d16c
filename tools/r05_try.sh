#!/bin/bash
# quick GPU check of a work-in-progress build: parity tests, then a short headline run
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests -m gpu -q ${PYTEST_ARGS} > gpurun_out/try_pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/try_pytest.log
timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs ${BENCH_ARGS} > gpurun_out/try_bench.json 2> gpurun_out/try_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/try_bench.err
python3 - <<'P'
import json
try:
    b=json.loads(open('gpurun_out/try_bench.json').read().strip().splitlines()[-1])
    x=b['roofline']['exclusive']
    print('ms_per_step', b['ms_per_step'], 'median', b['median_ms'], 'verified', b['verified'])
    print('exclusive phases', x['phase_ms']); print('overlapped phases', b['phase_ms'])
    print('pcie', (b.get('pcie_inclusive') or {}).get('median_ms'))
except Exception as e: print('no bench line', e)
P
