"""2^29 points (a 137 GB row table): the default window against a forced one.  python tools/huge_check.py [LOG2N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montgomery_amd.api import MsmContext

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 29
n = 1 << lg
ctx = MsmContext()
t = time.time(); ctx.generate_points(n, seed=5); print("gen points", lg, round(time.time() - t, 2), "s", flush=True)
dev, _ = ctx.generate_scalars(n, seed=6)
out = []
for c in (None, None, 16):
    t = time.time(); r, info = ctx.run_device(dev, n, c=c); dt = time.time() - t
    out.append(r.as_tuple()); print(lg, "c", info["c"], "K", info["K"], round(dt * 1e3, 1), "ms", hex(r.x)[:14], {k: round(v, 1) for k, v in info["phase_ms"].items()}, flush=True)
print("independent of c:", len(set(out)) == 1)
