import sys, os, time
sys.path.insert(0, "/root/repo")
from montgomery_amd.api import MsmContext
lg = int(sys.argv[1]); n = 1 << lg
ctx = MsmContext()
ctx.generate_points(n, seed=7)
dev, _ = ctx.generate_scalars(n, seed=9)
ctx.run_device(dev, n)
r, info = ctx.run_device(dev, n, serial=True)
t = time.perf_counter(); r2, info2 = ctx.run_device(dev, n); dt = time.perf_counter() - t
print("mult", os.environ.get("MSM_SORTB_MULT"), "serial sort ms", round(info["phase_ms"]["sort"], 2), "acc", round(info["phase_ms"]["accumulate"], 1), "overlapped total ms", round(dt * 1e3, 1))
