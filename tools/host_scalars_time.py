"""One MSM with the scalars handed over as a pageable HOST buffer (the PCIe-inclusive call): python tools/host_scalars_time.py [LOG2N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montgomery_amd.api import MsmContext

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << lg
ctx = MsmContext()
ctx.generate_points(n, seed=7)
dev, host = ctx.generate_scalars(n, seed=9, to_host=True, raw=True)
ctx.run_device(dev, n)
for i in range(4):
    t = time.perf_counter(); r, info = ctx.run_device(dev, n); d0 = time.perf_counter() - t
    t = time.perf_counter(); r2, info2 = ctx.run(host); d1 = time.perf_counter() - t
    assert r.as_tuple() == r2.as_tuple()
    print(f"resident {d0 * 1e3:7.1f} ms   host scalars {d1 * 1e3:7.1f} ms   upload phase {info2['phase_ms']['upload']:6.1f} ms", flush=True)
