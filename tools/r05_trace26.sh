#!/bin/bash
# kernel trace of a few headline MSMs -> gpurun_out/r05_kstats26.txt (per-kernel table) + the raw stats csv
cd "$(dirname "$0")/.."
REPO=$PWD; export TMPDIR=/tmp
OUT=$REPO/gpurun_out/trace_26; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 3 --warmup 1 --log2n 26 --no-cpu-baseline --no-other-configs --no-pcie --no-c16 --no-tables-leg ${BENCH_ARGS} > $OUT/log.txt 2>&1
cd $REPO
cp $(ls $OUT/*/*_kernel_stats.csv | head -1) gpurun_out/r05_kstats26.csv
find $OUT -name "*_kernel_trace.csv" -size +30M -delete
cut -c1-60,200- gpurun_out/r05_kstats26.csv | head -5
python3 - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r05_kstats26.csv')))
for r in rows[:28]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_ms {float(r['AverageNs'])/1e6:8.3f}")
P
tail -2 $OUT/log.txt | cut -c1-300
