import os, sys
sys.path.insert(0, "/root/repo")
from montgomery_amd.api import MsmContext
from montgomery_amd.distributed import combine_host
from oracle import msm_oracle as O
ctx = MsmContext()
ok = True
# small exact check against the oracle, then sizes against the default plan
pts, _ = O.random_points_bls377("fold", 300)
sc = O.prng_ints("fold/s", 300, O.BLS12_377.q)
ctx.set_points(O.points_to_bytes(pts, 48))
exp = O.msm_batched_affine(sc, pts, c=16)
for c in (18, 21):
    r, info = ctx.run(O.scalars_to_bytes(sc), c=c)
    print("n=300 c", c, "K", info["K"], r.as_tuple() == exp); ok &= r.as_tuple() == exp
    # extremes: scalars near 2^253, q-1, sums of carries
for lg in (12, 18, 22, 23, 24):
    n = 1 << lg
    ctx.generate_points(n, seed=lg)
    dev, _ = ctx.generate_scalars(n, seed=lg + 1)
    want, i0 = ctx.run_device(dev, n, c=16)
    for c in (18, 21):
        got, info = ctx.run_device(dev, n, c=c)
        good = got.as_tuple() == want.as_tuple()
        # window shards of the folded plan through the sharded entry + host combine
        K = info["K"]
        parts = b"".join(ctx.window_sums(dev, n, k, k + 1, c=c, on_device=True)[0] for k in range(K))
        good2 = combine_host(parts, K, c) == want.as_tuple()
        print("2^%d c %d K %d" % (lg, c, K), good, good2, info["phase_ms"]["total"]); ok &= good and good2
# adversarial: all scalars = q - 1, 2^126-ish halves
import ctypes
q = O.BLS12_377.q
for val in (q - 1, q - 2, 1, (1 << 252) - 1):
    n = 1 << 12
    ctx.generate_points(n, seed=99)
    sb = O.scalars_to_bytes([val] * n)
    want, _ = ctx.run(sb, c=16)
    for c in (18, 21):
        got, _ = ctx.run(sb, c=c)
        print("const scalar", hex(val)[:10], c, got.as_tuple() == want.as_tuple()); ok &= got.as_tuple() == want.as_tuple()
# msmProjective with the folded c = 23 (254 = 11 * 23 + 1)
n = 1 << 14
ctx.generate_points(n, seed=5)
dev, _ = ctx.generate_scalars(n, seed=6)
want, _ = ctx.run_device(dev, n)
got, info = ctx.run_device(dev, n, c=23, no_glv=True)
print("no_glv c 23 K", info["K"], got.as_tuple() == want.as_tuple()); ok &= got.as_tuple() == want.as_tuple()
print("ALL OK" if ok else "FAILED")
