#!/bin/bash
# per-kernel durations of the sort phase for several builds of the library: tools/ab_kernels.sh LOG2N name1 name2 ...
# ('-' = the in-tree build; others ab_builds/libmsm_<name>.so).  Two serialised MSMs per build under a kernel trace; prints
# the average duration of every sort kernel of the SECOND (warmed-up) MSM.
cd "$(dirname "$0")/.."
REPO=$PWD; export TMPDIR=/tmp
LG=$1; shift
for n in "$@"; do
  OUT=$REPO/gpurun_out/abk_$n; rm -rf $OUT; mkdir -p $OUT
  if [ "$n" = "-" ]; then unset MSM_HIP_LIB; else export MSM_HIP_LIB=$REPO/ab_builds/libmsm_$n.so; fi
  (cd /tmp && RUN_TWICE=1 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/run_once.py $LG ${MSM_C:-0} > $OUT/log.txt 2>&1)
  python3 - "$n" $OUT <<'P'
import csv, glob, re, sys, collections
name, out = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("msm::", "")) for r in rows))
starts = [i for i, e in enumerate(ev) if "k_digits" in e[2] or "k_te_digits" in e[2]]
half = len(starts) // 2
seg = ev[starts[half]:]
agg = collections.OrderedDict()
for s, e, k in seg:
    a = agg.setdefault(k, [0, 0]); a[0] += e - s; a[1] += 1
keys = [k for k in agg if any(t in k for t in ("digits", "k_bin", "k_hist", "k_radix", "k_scatter", "colscan", "pscan", "vscan", "bucket_max", "coarse"))]
print(f"{name:12s} " + "  ".join(f"{k.split('<')[0][2:]}={agg[k][0]/agg[k][1]/1e6:.3f}" for k in keys) + f"  | sort+digits total {sum(agg[k][0] for k in keys)/1e6:.2f} ms, MSM span {(seg[-1][1]-seg[0][0])/1e6:.1f} ms")
P
done
