"""Window size against time at several input sizes, plain path and window tables: python tools/plan_sweep.py "21 22 23" "16 18 21" [tables]
(BLS12-377; best of 6 warmed-up calls; every result compared with the first one of its size)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montgomery_amd.api import MsmContext

sizes = [int(x) for x in sys.argv[1].split()]
cs = [int(x) for x in sys.argv[2].split()]
with_tables = len(sys.argv) > 3
ctx = MsmContext()
for lg in sizes:
    n = 1 << lg
    ctx.generate_points(n, seed=7)
    dev, _ = ctx.generate_scalars(n, seed=9)
    ref = None
    row = []
    for c in cs:
        def best(f, reps=6):
            f(); ts = []
            for _ in range(reps):
                t = time.perf_counter(); r, i = f(); ts.append((time.perf_counter() - t) * 1e3)
            return min(ts), r, i
        ms, r, i = best(lambda: ctx.run_device(dev, n, c=c, no_tables=True))
        ref = ref or r.as_tuple()
        assert r.as_tuple() == ref
        cell = f"c={c} K={i['K']}: {ms:7.3f} (sort {i['phase_ms']['sort']:.2f})"
        if with_tables:
            try:
                ctx.precompute(n, c=c)
                mt, r, i = best(lambda: ctx.run_device(dev, n, c=c))
                assert r.as_tuple() == ref and i["tables"]
                cell += f" tables {mt:7.3f}"
            except Exception as e:   # (tables beyond msm_set_tables_limit)
                cell += " tables -"
        row.append(cell)
    print(f"2^{lg}: " + " | ".join(row), flush=True)
