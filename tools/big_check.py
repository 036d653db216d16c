import sys, time
sys.path.insert(0, "/root/repo")
from montgomery_amd.api import MsmContext
for lg in (27, 28):
    n = 1 << lg
    ctx = MsmContext()
    t = time.time(); ctx.generate_points(n, seed=5); print("gen points", lg, round(time.time() - t, 2), "s", flush=True)
    dev, _ = ctx.generate_scalars(n, seed=6)
    out = []
    for c in (None, 13, 19):
        t = time.time(); r, info = ctx.run_device(dev, n, c=c); dt = time.time() - t
        out.append(r.as_tuple()); print(lg, "c", info["c"], "K", info["K"], round(dt * 1e3, 1), "ms", hex(r.x)[:14], flush=True)
    print("independent of c:", out[0] == out[1] == out[2], flush=True)
    ctx.close()
