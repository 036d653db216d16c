#!/bin/bash
# SQ counter passes over one 2^24 MSM (own runs: --pmc only).  usage: tools/pmc_quick.sh <tag> [log2n]
TAG=${1:-x}; LG=${2:-24}
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/sq1 -- python3 $REPO/bench.py --steps 1 --warmup 0 --log2n $LG --no-cpu-baseline > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -- python3 $REPO/bench.py --steps 1 --warmup 0 --log2n $LG --no-cpu-baseline > $OUT/sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- python3 $REPO/bench.py --steps 1 --warmup 0 --log2n $LG --no-cpu-baseline > $OUT/grbm.log 2>&1
python3 - <<PY
import csv, glob, collections
for kind in ("sq1", "sq2", "grbm"):
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % kind)
    if not fs: print(kind, "no csv"); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "batch_add" not in k and "scatter" not in k: continue
        d = agg.setdefault(k, collections.OrderedDict())
        d.setdefault("launches", set()).add(r["Dispatch_Id"])
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k, d in agg.items():
        d["launches"] = len(d["launches"])
        print(kind, k, dict(d))
PY
find $OUT -name "*.csv" -size +5M -delete
