#!/bin/bash
# round-5 opening measurements: PMC passes at 2^20 for BASELINE configs 2 and 4, a kernel trace at 2^20, a short headline run
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/pmc_headline.sh r05 20 > gpurun_out/r05_pmc20.log 2>&1
bash tools/pmc_headline.sh r05 20 --curve ed377 > gpurun_out/r05_pmc20_ed.log 2>&1
bash tools/trace_quick.sh 20 --no-other-configs --no-pcie > gpurun_out/r05_trace20.txt 2>&1
python3 tools/gaps.py $(ls gpurun_out/trace_20/*/*_kernel_trace.csv | head -1) > gpurun_out/r05_gaps20.txt 2>&1
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/r05_bench26_base.json 2> gpurun_out/r05_bench26_base.err
tail -c 1500 gpurun_out/r05_bench26_base.json
