"""Window size against time on the plain path for one curve: python tools/plan_sweep_curve.py CURVE_ID "sizes" "windows" """
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montgomery_amd.api import MsmContext
curve = int(sys.argv[1]); sizes = [int(x) for x in sys.argv[2].split()]; cs = [int(x) for x in sys.argv[3].split()]
ctx = MsmContext(curve)
key = (lambda r: (r.x, r.y)) if curve == 1 else (lambda r: r.as_tuple())
for lg in sizes:
    n = 1 << lg
    ctx.generate_points(n, seed=7); dev, _ = ctx.generate_scalars(n, seed=9)
    ref, row = None, []
    for c in cs:
        ctx.run_device(dev, n, c=c, no_tables=True); ts = []
        for _ in range(5):
            t = time.perf_counter(); r, i = ctx.run_device(dev, n, c=c, no_tables=True); ts.append((time.perf_counter() - t) * 1e3)
        ref = ref or key(r); assert key(r) == ref
        row.append(f"c={c} K={i['K']}: {min(ts):7.3f}")
    print(f"curve {curve} 2^{lg}: " + " | ".join(row), flush=True)
