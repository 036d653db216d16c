import sys, time
sys.path.insert(0, "/root/repo")
from montgomery_amd.api import MsmContext
n = 1 << 26
ctx = MsmContext()
ctx.generate_points(n, seed=7)
dev, _ = ctx.generate_scalars(n, seed=9)
c, K = ctx.plan(n)
for (lo, hi) in ((0, 1), (7, 8), (0, 2), (0, 4), (0, 8)):
    ctx.window_sums(dev, n, lo, hi, c=c, on_device=True)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); parts, info = ctx.window_sums(dev, n, lo, hi, c=c, on_device=True); best = min(best, time.perf_counter() - t)
    print(f"windows [{lo},{hi}) of {K}: {best*1e3:.1f} ms", {k: round(v, 1) for k, v in info["phase_ms"].items()})
