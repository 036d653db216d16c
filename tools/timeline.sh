#!/bin/bash
# kernel timeline of one warmed-up MSM: tools/timeline.sh LOG2N [C] [CURVE_ID]   (curve ids of include/msm_hip.h: 0 BLS12-377, 1 Ed-377, ...)
cd "$(dirname "$0")/.."
REPO=$PWD; export TMPDIR=/tmp
OUT=$REPO/gpurun_out/tl_$1_${3:-0}; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && RUN_TWICE=1 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/run_once.py $1 ${2:-0} ${3:-0} > $OUT/log.txt 2>&1)
python3 - $OUT <<'P'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("msm::", "")) for r in rows))
starts = [i for i, e in enumerate(ev) if "k_digits" in e[2] or "k_te_digits" in e[2]]
seg = ev[starts[len(starts) // 2]:]
t0 = seg[0][0]; prev = t0
for s, e, k in seg:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {k[:60]}")
    prev = e
print(f"span {(seg[-1][1] - t0) / 1e3:.1f} us, busy {sum(e - s for s, e, _ in seg) / 1e3:.1f} us")
P
tail -1 $OUT/log.txt
