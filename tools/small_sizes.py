"""Window-size sweep at small and medium N (retuning pick_window): python tools/small_sizes.py [curve id]"""
import sys, time
sys.path.insert(0, "/root/repo")
from montgomery_amd.api import MsmContext
curve = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = MsmContext(curve)
import os
cands = tuple(int(x) for x in os.environ['CANDS'].split(',')) if os.environ.get('CANDS') else ((4, 6, 8, 10, 11, 13, 16) if curve != 1 else (4, 6, 7, 9, 12, 14, 16))
for lg in (4, 6, 8, 10, 12, 13, 14, 15, 16, 17, 18, 19):
    n = 1 << lg
    ctx.generate_points(n, seed=7)
    dev, _ = ctx.generate_scalars(n, seed=9)
    row = []
    for c in (None,) + cands:
        ctx.run_device(dev, n, c=c)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); r, info = ctx.run_device(dev, n, c=c); best = min(best, time.perf_counter() - t)
        row.append(f"{'def' if c is None else c}({info['c']}):{best*1e3:.2f}")
    print(f"2^{lg:2d}", "  ".join(row), flush=True)
