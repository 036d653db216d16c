"""Cost of the first MSM call of a process (context, buffers) against the later ones: python tools/cold_start.py LOG2N"""
import sys, time
sys.path.insert(0, "/root/repo")
from montgomery_amd.api import MsmContext
n = 1 << int(sys.argv[1])
t = time.perf_counter(); ctx = MsmContext(); print("ctx create %.1f ms" % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter(); ctx.generate_points(n, seed=7); print("gen points %.1f ms" % ((time.perf_counter() - t) * 1e3))
dev, _ = ctx.generate_scalars(n, seed=9)
for i in range(3):
    t = time.perf_counter(); r, info = ctx.run_device(dev, n); print("msm call %d: %.1f ms" % (i, (time.perf_counter() - t) * 1e3), {k: round(v, 1) for k, v in info["phase_ms"].items()})
