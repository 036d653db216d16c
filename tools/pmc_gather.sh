#!/bin/bash
# Memory-system counters of the accumulation kernels for one window size: tools/pmc_gather.sh TAG LOG2N C
TAG=$1; LG=$2; CC=$3
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmcg_$TAG
mkdir -p $OUT
cd /tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_LEVEL_sum GRBM_GUI_ACTIVE" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $REPO/tools/run_once.py $LG $CC > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_batch_add" not in k: continue
            mode = "gather" if "ELi0E" in k else "regular" if "ELi1E" in k else "search"
            agg[mode][r["Counter_Name"]] += float(r["Counter_Value"])
        for m in agg:
            print("$TAG", m, dict(agg[m]))
PY
