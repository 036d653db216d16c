#!/usr/bin/env python3
"""Static opcode histogram of one kernel's gfx950 assembly (hipcc -S --cuda-device-only output).

usage: isa_hist.py file.s kernel_substring [--blocks]
Prints, per basic block (label to label), the number of VALU / SALU / VMEM / LDS instructions and the
v_mad_u64_u32 count, then an opcode histogram of the whole kernel.  Used for profiles/r02_isa_*.txt.
"""
import collections, re, sys

def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (l.startswith("_Z") and key in l.split(":")[0] and ":" in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks = []
    cur = ["<entry>", collections.Counter()]
    total = collections.Counter()
    for l in lines[start + 1:end + 1]:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                blocks.append(cur)
                cur = [m.group(1) + ("  " + s.split(";", 1)[1].strip() if ";" in s else ""), collections.Counter()]
            continue
        if s.startswith(";;#"):
            continue
        op = s.split()[0]
        if op.endswith(":"):
            continue
        cur[1][op] += 1
        total[op] += 1
    blocks.append(cur)
    def cls(op):
        if op.startswith("v_"): return "VALU"
        if op.startswith("s_"): return "SALU"
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM"
        if op.startswith("ds_"): return "LDS"
        return "other"
    if show_blocks:
        for name, c in blocks:
            k = collections.Counter()
            for op, n in c.items(): k[cls(op)] += n
            print(f"{name[:70]:70s} VALU {k['VALU']:5d} (mad64 {c['v_mad_u64_u32']+c['v_mad_i64_i32']:4d})  SALU {k['SALU']:4d}  VMEM {k['VMEM']:3d}  LDS {k['LDS']:3d}")
    k = collections.Counter()
    for op, n in total.items(): k[cls(op)] += n
    print("kernel total:", dict(k))
    for op, n in total.most_common(40):
        print(f"  {op:28s} {n}")

main()
