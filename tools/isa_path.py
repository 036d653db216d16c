#!/usr/bin/env python3
"""Main-path opcode count of one loop of a kernel's assembly: starts at a label, follows every forward conditional
branch as TAKEN (the wave-uniform side branches of k_batch_add guard rare cases and are skipped by their
s_cbranch_*z), stops when it is back at the label.   usage: isa_path.py file.s kernel_substring label [--scc-falls] [-v]"""
import collections, re, sys

def main():
    path, key, label = sys.argv[1:4]
    verbose = "-v" in sys.argv
    scc_falls = "--scc-falls" in sys.argv   # treat s_cbranch_scc* as not taken
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    pos = {l.split(":")[0]: i for i, l in enumerate(lines[start:end], start) if re.match(r"^\.LBB\d+_\d+:", l)}
    i = pos[label]
    hist = collections.Counter()
    n = 0
    while i < end:
        s = lines[i].split(";")[0].strip()
        i += 1
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        op = s.split()[0]
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", s)
        b = re.match(r"s_branch\s+(\.LBB\d+_\d+)", s)
        hist[op] += 1
        n += 1
        if verbose: print(s)
        if m and "scc" in s.split()[0] and scc_falls:
            continue         # uniform loop control (i > 0, steps left): not the rare-case guards
        if m or b:
            tgt = pos[(m or b).group(1)]
            if tgt <= pos[label] or n > 20000:      # back at the header: one trip done
                break
            i = tgt
        elif i < end and lines[i].startswith(label + ":"):
            break
    valu = sum(c for o, c in hist.items() if o.startswith("v_"))
    mad = hist["v_mad_u64_u32"]
    print(f"{label}: {n} instructions, VALU {valu}, v_mad_u64_u32 {mad}, other VALU {valu - mad}, "
          f"VMEM {sum(c for o, c in hist.items() if o.startswith(('global_', 'buffer_', 'scratch_', 'flat_')))}, "
          f"SALU {sum(c for o, c in hist.items() if o.startswith('s_'))}")
    for o, c in hist.most_common(40):
        print(f"  {o:28s} {c}")

main()
