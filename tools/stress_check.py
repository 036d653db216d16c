"""Robustness checks at sizes the unit tests do not reach (run on the GPU box): ragged large N, skewed scalar
distributions (few distinct scalars -> huge buckets), every curve, two live contexts, forced window sizes."""
import sys, time
sys.path.insert(0, "/root/repo")
from oracle import msm_oracle as O
from montgomery_amd import _lib
from montgomery_amd.api import MsmContext

ok = True
def check(name, cond, extra=""):
    global ok
    ok &= bool(cond)
    print(("PASS " if cond else "FAIL ") + name, extra, flush=True)

for cid, B in ((_lib.CURVE_BLS12_377_G1, O.BLS12_377), (_lib.CURVE_BLS12_381_G1, O.BLS12_381), (_lib.CURVE_PALLAS, O.PALLAS)):
    ctx = MsmContext(cid)
    G = (B.gx, B.gy)
    for n in ((1 << 22) + 12345, (1 << 17) - 1, 3 * (1 << 19) + 7):
        a = O.scalars_from_bytes(ctx.generate_points(n, seed=n & 0xffff, want_scalars=True))
        dev, sb = ctx.generate_scalars(n, seed=99, to_host=True)
        s = O.scalars_from_bytes(sb)
        exp = O.aff_scale(sum(x * y for x, y in zip(a, s)) % B.q, G, B.p)
        for c in (None, 12, 19):
            t = time.time(); r, info = ctx.run_device(dev, n, c=c); dt = time.time() - t
            check(f"{B.label} n={n} c={info['c']}", r.as_tuple() == exp, f"{dt*1e3:.1f} ms")
        # skewed: only 5 distinct scalars -> a handful of enormous buckets per window
        vals = O.prng_ints("skew", 5, B.q)
        sk = [vals[i % 5] for i in range(n)]
        exp2 = O.aff_scale(sum(x * y for x, y in zip(a, sk)) % B.q, G, B.p)
        r, info = ctx.run(O.scalars_to_bytes(sk))
        check(f"{B.label} n={n} 5 distinct scalars", r.as_tuple() == exp2, f"max_bucket={info['max_bucket']} rounds={info['rounds']}")
        # small scalars only (upper windows empty)
        sm = [(i * 2654435761) & 0xFFFFF for i in range(n)]
        exp3 = O.aff_scale(sum(x * y for x, y in zip(a, sm)) % B.q, G, B.p)
        r, info = ctx.run(O.scalars_to_bytes(sm))
        check(f"{B.label} n={n} 20-bit scalars", r.as_tuple() == exp3)
    ctx.close()

# two live contexts on one GPU, interleaved calls
c1, c2 = MsmContext(_lib.CURVE_BLS12_377_G1), MsmContext(_lib.CURVE_BLS12_381_G1)
n = 1 << 16
a1 = O.scalars_from_bytes(c1.generate_points(n, seed=1, want_scalars=True))
a2 = O.scalars_from_bytes(c2.generate_points(n, seed=2, want_scalars=True))
d1, s1 = c1.generate_scalars(n, seed=3, to_host=True)
d2, s2 = c2.generate_scalars(n, seed=4, to_host=True)
for _ in range(3):
    r1, _i = c1.run_device(d1, n)
    r2, _i = c2.run_device(d2, n)
B1, B2 = O.BLS12_377, O.BLS12_381
check("two contexts / 377", r1.as_tuple() == O.aff_scale(sum(x * y for x, y in zip(a1, O.scalars_from_bytes(s1))) % B1.q, (B1.gx, B1.gy), B1.p))
check("two contexts / 381", r2.as_tuple() == O.aff_scale(sum(x * y for x, y in zip(a2, O.scalars_from_bytes(s2))) % B2.q, (B2.gx, B2.gy), B2.p))
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
