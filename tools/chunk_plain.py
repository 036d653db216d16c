"""One size on the plain path (no window tables), best of five warmed-up calls: python tools/chunk_plain.py LOG2N"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montgomery_amd.api import MsmContext
lg = int(sys.argv[1]); n = 1 << lg
ctx = MsmContext(); ctx.generate_points(n, seed=7); dev, _ = ctx.generate_scalars(n, seed=9)
for i in range(2): ctx.run_device(dev, n, no_tables=True)
ts = []
for i in range(5):
    t = time.perf_counter(); r, info = ctx.run_device(dev, n, no_tables=True); ts.append((time.perf_counter() - t) * 1e3)
print(lg, "plain c", info["c"], "min %.3f ms" % min(ts))
