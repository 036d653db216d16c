#!/bin/bash
# Counters of the bucket-reduction kernels at one size: tools/pmc_reduce.sh TAG LOG2N
TAG=$1; LG=$2
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmcr_$TAG
mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $REPO/tools/run_once.py $LG > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for name in ("k_bit_tree", "k_bucket_reduce", "k_bucket_finish"):
                if name in k:
                    agg[name + "#" + r["Dispatch_Id"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for m in sorted(agg):
            print(m, {k: sum(v) for k, v in agg[m].items()})
PY
