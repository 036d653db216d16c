#!/bin/bash
# SQ counter pass over the accumulation kernels (own run: --pmc only, no tracing domains).
TAG=${1:-r01}
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/sq1 -- python3 $REPO/bench.py --steps 1 --warmup 0 --log2n 24 --no-cpu-baseline > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -- python3 $REPO/bench.py --steps 1 --warmup 0 --log2n 24 --no-cpu-baseline > $OUT/sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 $REPO/bench.py --steps 1 --warmup 0 --log2n 24 --no-cpu-baseline > $OUT/tcc.log 2>&1
ls -laR $OUT | head -40
