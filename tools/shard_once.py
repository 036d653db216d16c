"""One single-window shard of the 2^26 MSM, three times (profiling target for the per-rank timeline of an 8-GPU run)."""
import sys
sys.path.insert(0, "/root/repo")
from montgomery_amd.api import MsmContext
n = 1 << 26
ctx = MsmContext()
ctx.generate_points(n, seed=7)
dev, _ = ctx.generate_scalars(n, seed=9)
c, K = ctx.plan(n)
for _ in range(3):
    parts, info = ctx.window_sums(dev, n, 3, 4, c=c, on_device=True)
print({k: round(v, 2) for k, v in info["phase_ms"].items()})
