#!/usr/bin/env python3
"""Straight-line segments of one kernel's assembly: every stretch of instructions between two labels / branches, with its
VALU and v_mad_u64_u32 counts and the branch that ends it.  The wave-uniform side branches of k_batch_add (identity
operands, P + P, the rare excess over p) show up as segments "after" an s_cbranch_*z; adding up the others gives the
instructions a wave really issues per step (isa_hist.py counts whole labelled blocks, rare tails included).

usage: isa_path.py file.s kernel_substring [--sum SEG,SEG,...]   (SEG = index printed in the first column)"""
import re, sys


def main():
    path, key = sys.argv[1:3]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    segs = []   # [name, valu, mad, vmem, salu, terminator]
    cur = ["<entry>", 0, 0, 0, 0, ""]
    for i in range(start + 1, end + 1):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        s = l.split(";")[0].strip()
        if m:
            segs.append(cur)
            cur = [m.group(1), 0, 0, 0, 0, ""]
            continue
        if not s or s.startswith("."):
            continue
        op = s.split()[0]
        if op.startswith("v_"): cur[1] += 1
        if op == "v_mad_u64_u32": cur[2] += 1
        if op.startswith(("global_", "buffer_", "scratch_", "flat_", "ds_")): cur[3] += 1
        if op.startswith("s_") and not op.startswith(("s_cbranch", "s_branch")): cur[4] += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            cur[5] = s
            segs.append(cur)
            cur = [cur[0].split()[0] + " (after the branch)", 0, 0, 0, 0, ""]
    segs.append(cur)
    segs = [s for s in segs if s[1] or s[3] or s[5]]
    for n, (name, valu, mad, vmem, salu, term) in enumerate(segs):
        print(f"{n:3d} {name:34s} VALU {valu:5d}  mad64 {mad:4d}  other {valu - mad:4d}  VMEM {vmem:3d}  SALU {salu:3d}   {term}")
    if "--sum" in sys.argv:
        pick = [int(x) for x in sys.argv[sys.argv.index("--sum") + 1].split(",")]
        v = sum(segs[i][1] for i in pick); m = sum(segs[i][2] for i in pick)
        print(f"sum over {pick}: VALU {v}, v_mad_u64_u32 {m}, other VALU {v - m}, VMEM {sum(segs[i][3] for i in pick)}, SALU {sum(segs[i][4] for i in pick)}")


main()
