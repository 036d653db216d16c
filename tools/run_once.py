"""One MSM on generated inputs (profiling target): python3 tools/run_once.py LOG2N [C] [CURVE_ID]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montgomery_amd.api import MsmContext

lg = int(sys.argv[1]); n = 1 << lg
c = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) else None
ctx = MsmContext(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
ctx.generate_points(n, seed=7)
dev, _ = ctx.generate_scalars(n, seed=9)
if os.environ.get("RUN_TWICE"):   # a warmed-up second call (workspace already allocated)
    ctx.run_device(dev, n, c=c, serial=True)
r, info = ctx.run_device(dev, n, c=c, serial=True)
print(hex(r.x)[:18], info["c"], info["K"], {k: round(v, 2) for k, v in info["phase_ms"].items()})
