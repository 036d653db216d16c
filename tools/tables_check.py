import sys, time, json
sys.path.insert(0, '/root/repo')
from montgomery_amd.api import MsmContext
from oracle import msm_oracle as O
C = O.BLS12_377
for curve, lgs in ((0, (13, 16, 20, 22)), (1, (14, 20)), (2, (16,)), (3, (16,))):
    for lg in lgs:
        n = 1 << lg
        ctx = MsmContext(curve)
        ctx.generate_points(n, seed=7 + lg)
        dev, _ = ctx.generate_scalars(n, seed=9 + lg)
        plain, ip = ctx.run_device(dev, n, no_tables=True)
        assert not ip["tables"]
        if curve == 1: ctx.precompute()   # the Edwards path builds tables only when asked to
        t0 = time.perf_counter(); tab, it = ctx.run_device(dev, n); build = time.perf_counter() - t0
        assert it["tables"], (curve, lg, it)
        tabs = []
        for _ in range(6):
            t = time.perf_counter(); r, i2 = ctx.run_device(dev, n); tabs.append((time.perf_counter() - t) * 1e3)
        pl = []
        for _ in range(6):
            t = time.perf_counter(); r2, i3 = ctx.run_device(dev, n, no_tables=True); pl.append((time.perf_counter() - t) * 1e3)
        ok = (tab.as_tuple() if curve != 1 else (tab.x, tab.y)) == (plain.as_tuple() if curve != 1 else (plain.x, plain.y))
        print(json.dumps({"curve": curve, "lg": lg, "equal": ok, "c": it["c"], "K": it["K"], "tables": ctx.tables_info(), "first_call_s": round(build, 3),
                          "tables_ms": round(min(tabs), 3), "plain_ms": round(min(pl), 3), "phase_tables": {k: round(v, 3) for k, v in i2["phase_ms"].items()},
                          "phase_plain": {k: round(v, 3) for k, v in i3["phase_ms"].items()}}), flush=True)
        assert ok
        ctx.close()
