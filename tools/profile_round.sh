#!/bin/bash
# Profiling recipe for one round (run on the GPU box through gpurun):  tools/profile_round.sh r02
# Writes rocprofv3 summaries under gpurun_out/prof_<tag>/; tools/collect_profiles.py copies the ones to keep into profiles/.
set -x
TAG=${1:-r06}
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
python bench.py --steps 10 --warmup 5 > $OUT/bench_2p26.json 2> $OUT/bench_2p26.err
python bench.py --steps 10 --warmup 5 --log2n 20 --no-cpu-baseline --no-other-configs > $OUT/bench_2p20.json 2> $OUT/bench_2p20.err
python bench.py --curve ed377 --steps 10 --warmup 5 --log2n 20 > $OUT/bench_ed377_2p20.json 2> $OUT/bench_ed.err
python bench.py --curve bls12-381 --steps 10 --warmup 5 > $OUT/bench_bls381_2p26.json 2> $OUT/bench_381.err
python bench.py --curve bls12-381 --steps 10 --warmup 5 --log2n 20 > $OUT/bench_bls381_2p20.json 2>> $OUT/bench_381.err
python tools/cpu_series.py $OUT/cpu_baseline.json > $OUT/cpu_series.log 2>&1
for u in ubench_int2; do [ -x tools/$u ] && ./tools/$u > $OUT/$u.txt 2>&1; done
[ -x tools/ubench_gather ] && { ./tools/ubench_gather 16 512; for b in 256 1024 2048; do echo "workgroups $b"; ./tools/ubench_gather 16 $b | grep -E "range    16384 MB  wave window (       0|     256) MB"; done; } > $OUT/ubench_gather.txt 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace26 -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-other-configs --no-skewed > $OUT/trace26.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace20 -- python3 $REPO/bench.py --steps 5 --warmup 1 --log2n 20 --no-cpu-baseline --no-verify --no-other-configs --no-skewed > $OUT/trace20.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ed20 -- python3 $REPO/bench.py --curve ed377 --steps 5 --warmup 1 --log2n 20 > $OUT/trace_ed20.log 2>&1
cd $REPO
# the sharded path as the driver runs it, here with both ranks on the one GPU of the box (gloo): self-launched, both splits
python bench.py --gpus 2 --dist-backend gloo --log2n 22 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_2rank_gloo_2p22.json 2> $OUT/bench_2rank.err
[ -f montgomery_amd/msm_hip.node ] && node js/bench-msm.js 20 > $OUT/js_bench_2p20.txt 2>&1
# PMC passes at the headline size (own runs, --pmc only): profiles/<tag>_pmc_2p26.json through tools/collect_pmc.py
tools/pmc_headline.sh $TAG 26 > $OUT/pmc_headline.log 2>&1
tools/pmc_headline.sh $TAG 20 > $OUT/pmc_2p20.log 2>&1
tools/pmc_headline.sh $TAG 20 --curve ed377 > $OUT/pmc_ed377_2p20.log 2>&1
python3 tools/shard_time.py 26 > $OUT/shard_proxy.txt 2>&1
for lg in 20 22 24 26; do python3 tools/skew_time.py $lg; done > $OUT/skew_time.txt 2>&1
find $OUT -name "*.csv" -size +20M -delete
ls -laR $OUT | head -80
