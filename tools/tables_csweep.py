"""Window size against time on window tables (msm_precompute builds them for the c asked for): python tools/tables_csweep.py CURVE LOG2N c1 c2 ..."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montgomery_amd.api import MsmContext
curve, lg = int(sys.argv[1]), int(sys.argv[2])
n = 1 << lg
ctx = MsmContext(curve)
ctx.generate_points(n, seed=7)
dev, _ = ctx.generate_scalars(n, seed=9)
ref, _ = ctx.run_device(dev, n, no_tables=True)
key = (lambda r: (r.x, r.y)) if curve == 1 else (lambda r: r.as_tuple())
def best(f, reps=8):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r, i = f(); ts.append((time.perf_counter() - t) * 1e3)
    return min(ts), r, i
ms, r, i = best(lambda: ctx.run_device(dev, n, no_tables=True))
print(f"plain default c={i['c']} K={i['K']}: {ms:.3f} ms", {k: round(v, 3) for k, v in i["phase_ms"].items()}, flush=True)
for c in [int(x) for x in sys.argv[3:]]:
    t = time.perf_counter(); info = ctx.precompute(n, c=c); tb = time.perf_counter() - t
    ms, r, i = best(lambda: ctx.run_device(dev, n, c=c))
    assert key(r) == key(ref), c
    print(f"tables c={c} K={i['K']} tables={i['tables']} ({info[2] / 2**30:.2f} GiB, built in {tb:.2f} s): {ms:.3f} ms", {k: round(v, 3) for k, v in i["phase_ms"].items()}, flush=True)
    msp, r, i = best(lambda: ctx.run_device(dev, n, c=c, no_tables=True), 4)
    print(f"   plain c={c}: {msp:.3f} ms", flush=True)
