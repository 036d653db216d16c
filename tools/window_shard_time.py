import sys, time
sys.path.insert(0, "/root/repo")
from montgomery_amd.api import MsmContext
n = 1 << 26
ctx = MsmContext()
ctx.generate_points(n, seed=7)
dev, _ = ctx.generate_scalars(n, seed=9)
def best_of(f, reps=3):
    f(); best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); f(); best = min(best, time.perf_counter() - t)
    return best * 1e3
full = best_of(lambda: ctx.run_device(dev, n, no_tables=True))
print("full", round(full, 1))
for (c, lo, hi) in ((16, 0, 4), (16, 4, 8), (21, 0, 3), (21, 3, 6), (21, 0, 2), (21, 2, 4), (21, 4, 6), (16, 0, 2), (16, 6, 8), (16, 0, 1), (16, 7, 8)):
    t = best_of(lambda: ctx.window_sums(dev, n, lo, hi, c=c, on_device=True))
    tm = best_of(lambda: ctx.window_sums(dev, n, lo, hi, c=c, on_device=True, merged=True))
    print(f"c={c} windows [{lo},{hi}): {t:.1f} ms  (merged sums {tm:.1f})  x{full / t:.2f}")
