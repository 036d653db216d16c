#!/usr/bin/env python3
"""Approximate VGPR liveness over one kernel's assembly, treating a line range as straight-line code.

usage: isa_live.py file.s kernel_substring [first_line last_line]   (line numbers relative to the kernel's first line)
Backward pass over the instructions: a register is live from its definition to its last use.  Branches are ignored
(rarely taken side blocks are read as if they ran), so the numbers are an upper estimate; good enough to see WHERE
the pressure of a long unrolled body peaks and what is live there."""
import re, sys

def regs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(3) is not None: out.append(int(m.group(3)))
        else: out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(body)
    ins = []
    for n in range(lo, hi):
        s = body[n].strip()
        if not s or s.startswith((";", ".")) or s.endswith(":"): continue
        s = s.split(";")[0].strip()
        if not s: continue
        op, _, rest = s.partition(" ")
        ops = [x.strip() for x in rest.split(",")] if rest else []
        if not (op.startswith(("v_", "global_", "ds_", "scratch_", "buffer_", "flat_"))): 
            continue
        if op.startswith(("global_store", "ds_write", "scratch_store", "buffer_store")) or op.startswith("v_cmp"):
            d, u = [], [r for o in ops for r in regs(o)]
        elif op.startswith("v_cmpx"):
            d, u = [], [r for o in ops for r in regs(o)]
        else:
            d = regs(ops[0]) if ops else []
            u = [r for o in ops[1:] for r in regs(o)]
            if op in ("v_mad_u64_u32", "v_mad_i64_i32", "v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_subbrev_co_u32") and len(ops) > 1 and not regs(ops[1]):
                pass
        ins.append((n, op, set(d), set(u)))
    live = set()
    trace = []
    for n, op, d, u in reversed(ins):
        live -= d
        live |= u
        trace.append((n, op, len(live), frozenset(live)))
    trace.reverse()
    mx = max(trace, key=lambda x: x[2])
    print("instructions", len(ins), "max live", mx[2], "at line", mx[0], mx[1])
    step = max(1, len(trace) // 60)
    for k in range(0, len(trace), step):
        n, op, c, _ = trace[k]
        print(f"  line {n:6d} {op:24s} live {c}")
    print("live at max:", sorted(mx[3]))

main()
