#!/bin/bash
# PMC passes over the HEADLINE configuration (2^26 by default), each counter group in its own rocprofv3 run (--pmc only, no
# trace domains), as MI355X_MICROARCH.md prescribes.  usage: tools/pmc_headline.sh TAG [LOG2N] [extra bench.py arguments, e.g. --c 16]
# Every run holds two MSMs (the timed step and the serialised "exclusive" one); tools/collect_pmc.py TAG LOG2N turns the
# counter files into profiles/<TAG>_pmc_2p<LOG2N>.json with per-pair-addition figures.
TAG=${1:-r06}; LG=${2:-26}; EXTRA="${@:3}"
CURVE=bls12-377; SFX=""
case "$EXTRA" in *"--curve ed377"*) CURVE=ed377; SFX=_ed377;; esac
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_${TAG}${SFX}_2p$LG
rm -rf $OUT; mkdir -p $OUT
PM="python3 $REPO/bench.py --steps 1 --warmup 0 --log2n $LG --no-cpu-baseline --no-verify --no-other-configs --no-pcie --no-c16 --no-tables-leg --no-skewed $EXTRA"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PM > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $PM > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- $PM > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- $PM > $OUT/pmc_grbm.log 2>&1
cd $REPO
python3 tools/collect_pmc.py $TAG $LG $CURVE
find $OUT -name "*.csv" -size +20M -delete
tail -2 $OUT/pmc_fetch.log | cut -c1-600
