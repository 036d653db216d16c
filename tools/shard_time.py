"""Per-rank work of a G-GPU run of the 2^26 MSM, timed on ONE GPU: the window shard (K / G windows over all points), the
points shard (all K windows over n / G points) and the bucket shard (all K windows over all points, 1 / G of every window's
buckets), for G = 2, 4, 8.  The proxy behind montgomery_amd.distributed.choose_split: ONE GPU, one rank at a time.
usage: python tools/shard_time.py [LOG2N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montgomery_amd.api import MsmContext
from montgomery_amd.distributed import choose_window
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << lg
ctx = MsmContext()
ctx.generate_points(n, seed=7)
dev, _ = ctx.generate_scalars(n, seed=9)
c, K = ctx.plan(n, no_tables=True)

def best_of(f, reps=3):
    f()
    best, info = 1e9, None
    for _ in range(reps):
        t = time.perf_counter(); _, info = f(); best = min(best, time.perf_counter() - t)
    return best * 1e3, info

full, _ = best_of(lambda: ctx.run_device(dev, n, no_tables=True))
print(f"2^{lg}, c = {c}, K = {K}: full MSM {full:.1f} ms")
plan = lambda m, cc, no_tables=False, merged=False: ctx.plan(m, cc, no_tables=no_tables, merged=merged)
for G in (2, 4, 8):
    # the windows each sharding runs with (choose_window: K the ranks divide / the pick for a rank's share of the points)
    cw, Kw = choose_window(plan, n, G, "windows")
    cp, Kp = choose_window(plan, n, G, "points")
    w = -(-Kw // G)
    # the slowest window shard: the lowest windows are full, the top one is lighter
    tw, iw = best_of(lambda: ctx.window_sums(dev, n, 0, w, c=cw, on_device=True))
    tt, _ = best_of(lambda: ctx.window_sums(dev, n, Kw - w, Kw, c=cw, on_device=True))
    m = n // G
    first = (G - 1) * m
    tp, ip = best_of(lambda: ctx.window_sums(dev + 32 * first, m, 0, Kp, c=cp, on_device=True, point_lo=first))
    # the same shard on the window tables of its range of the points (round 6), where they fit the limit (10 % of the device)
    ctx.precompute(m, c=cp, point_lo=first)
    tpt, ipt = best_of(lambda: ctx.window_sums(dev + 32 * first, m, 0, Kp, c=cp, on_device=True, point_lo=first, merged=True))
    on_tab = bool(ipt["tables"])
    ctx.precompute(4096, point_lo=0)   # (tables of a tiny range from here on: the other shards run the plain path)
    # bucket shard: the single-GPU plan, the slowest of the first and the last range of the buckets
    tb, ib = max((best_of(lambda g=g: ctx.window_sums(dev, n, 0, K, c=c, on_device=True, bucket_shard=(g, G))) for g in (0, G - 1)), key=lambda x: x[0])
    print(f"G = {G}: window shard (c = {cw}, {w} of {Kw} windows) {tw:.1f} ms (top windows {tt:.1f})  -> x{full / tw:.2f}   "
          f"points shard (c = {cp}) {tp:.1f} ms -> x{full / tp:.2f}   " + (f"on range tables {tpt:.1f} ms -> x{full / tpt:.2f}   " if on_tab else
          f"(range tables do not fit the limit; merged sums {tpt:.1f} ms)   ") + f"bucket shard (c = {c}, K = {K}) {tb:.1f} ms -> x{full / tb:.2f}")
    print("   bucket phases", {k: round(v, 1) for k, v in ib["phase_ms"].items()})
    print("   window phases", {k: round(v, 1) for k, v in (iw or {"phase_ms": {}})["phase_ms"].items()})
    print("   points phases", {k: round(v, 1) for k, v in ip["phase_ms"].items()})
