"""Launch-gap analysis of one MSM from a rocprofv3 kernel trace: python tools/gaps.py KERNEL_TRACE.csv
Prints, for the LAST MSM in the trace (from its k_digits to its last kernel): wall span, summed kernel time, and the
largest gaps with the kernels on either side."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("msm::", "")) for r in rows))
starts = [i for i, e in enumerate(ev) if "k_digits" in e[2] or "k_te_digits" in e[2]]
i0 = starts[-1]
seg = ev[i0:]
span = seg[-1][1] - seg[0][0]
busy = sum(e - s for s, e, _ in seg)
print(f"kernels {len(seg)}  span {span/1e3:.1f} us  busy {busy/1e3:.1f} us  gaps {100*(span-busy)/span:.1f} %")
gaps = sorted(((seg[i + 1][0] - seg[i][1], seg[i][2], seg[i + 1][2]) for i in range(len(seg) - 1)), reverse=True)
for g, a, b in gaps[:8]:
    print(f"  {g/1e3:8.1f} us between {a[:32]} -> {b[:32]}")
