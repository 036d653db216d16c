#!/bin/bash
# kernel timeline of one warmed-up OVERLAPPED MSM (two window groups on two streams): tools/timeline2.sh LOG2N
cd "$(dirname "$0")/.."
REPO=$PWD; export TMPDIR=/tmp
OUT=$REPO/gpurun_out/tl2_$1; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/run2.py <<P
import sys
sys.path.insert(0, "$REPO")
from montgomery_amd.api import MsmContext
n = 1 << $1
ctx = MsmContext(0)
ctx.generate_points(n, seed=7)
dev, _ = ctx.generate_scalars(n, seed=9)
for i in range(3):
    r, info = ctx.run_device(dev, n, no_tables=True)
print(info["phase_ms"])
P
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 /tmp/run2.py > $OUT/log.txt 2>&1)
python3 - $OUT <<'P'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("msm::", ""), r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows))
starts = [i for i, e in enumerate(ev) if "k_digits" in e[2]]
seg = ev[starts[-2]:]
t0 = seg[0][0]
qs = sorted(set(e[3] for e in seg))
for s, e, k, q in seg:
    if e - s < 150000 and "batch_add" not in k: continue
    print(f"{(s - t0) / 1e6:9.3f} .. {(e - t0) / 1e6:9.3f} ms  q{qs.index(q)}  {(e - s) / 1e6:8.3f} ms  {k[:44]}")
print(f"span {(seg[-1][1] - t0) / 1e6:.2f} ms")
P
tail -1 $OUT/log.txt
