import sys, time
sys.path.insert(0, "/root/repo")
from oracle import msm_oracle as O
from montgomery_amd.api import MsmContext
C = O.BLS12_377
ctx = MsmContext()
G = (C.gx, C.gy)
for lg, cs in ((16, (17, 18, 20, 22)), (20, (19, 22))):
    n = 1 << lg
    a = O.scalars_from_bytes(ctx.generate_points(n, seed=5, want_scalars=True))
    dev, sb = ctx.generate_scalars(n, seed=6, to_host=True)
    s = O.scalars_from_bytes(sb)
    exp = O.aff_scale(sum(x * y for x, y in zip(a, s)) % C.q, G, C.p)
    for c in cs:
        r, info = ctx.run_device(dev, n, c=c)
        print(lg, c, r.as_tuple() == exp, info["phase_ms"]["total"], info["rounds"], info["max_bucket"], flush=True)
