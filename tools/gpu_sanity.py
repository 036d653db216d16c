"""First-light check on a GPU box: field ops, GLV, batch add and small MSMs against the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import msm_oracle as O
from montgomery_amd import _lib
from montgomery_amd.api import MsmContext

C = O.BLS12_377
p = C.p
R = 1 << 390
ctx = MsmContext()

def tb(v): return v.to_bytes(48, "little")
def fb(b, i): return int.from_bytes(b[48*i:48*i+48], "little")

vals = O.prng_ints("fp", 300, p) + [0, 1, p - 1, p - 2, 2]
n = len(vals)
a = b"".join(tb(v) for v in vals)
b = b"".join(tb(v) for v in reversed(vals))
out = ctx.test_fp(_lib.OP_MUL, a, b)
bad = sum(fb(out, i) != vals[i] * vals[n-1-i] * pow(R, -1, p) % p for i in range(n))
print("mul mismatches", bad)
out = ctx.test_fp(_lib.OP_SQR, a)
print("sqr mismatches", sum(fb(out, i) != vals[i] * vals[i] * pow(R, -1, p) % p for i in range(n)))
out = ctx.test_fp(_lib.OP_ADD, a, b)
print("add mismatches", sum(fb(out, i) != (vals[i] + vals[n-1-i]) % p for i in range(n)))
out = ctx.test_fp(_lib.OP_SUB, a, b)
print("sub mismatches", sum(fb(out, i) != (vals[i] - vals[n-1-i]) % p for i in range(n)))
nz = [v for v in vals if v]
out = ctx.test_fp(_lib.OP_INV, b"".join(tb(v) for v in nz))
print("inv mismatches", sum(fb(out, i) != pow(nz[i], -1, p) * R * R % p for i in range(len(nz))))
out = ctx.test_fp(_lib.OP_TO_MONT, a)
print("to_mont mismatches", sum(fb(out, i) != vals[i] * R % p for i in range(n)))

g = O.glv_params(C.q, C.lam)
sc = O.prng_ints("glv", 2000, C.q) + [0, 1, C.q - 1, C.lam, C.q // 2]
res = ctx.test_glv(O.scalars_to_bytes(sc))
print("glv mismatches", sum(tuple(r) != O.glv_decompose(s, g) for r, s in zip(res, sc)))

pts, ks = O.random_points_bls377("pts", 64)
G = (C.gx, C.gy)
# batch add incl. edge cases
gs = pts[:20] + [pts[0], pts[1], None, pts[3], None]
hs = pts[20:40] + [pts[0], O.aff_neg(pts[1], p), pts[2], None, None]
enc = lambda P: b"\0" * 96 if P is None else tb(P[0]) + tb(P[1])
out = ctx.test_batch_add(b"".join(map(enc, gs)), b"".join(map(enc, hs)))
bad = 0
for i, (g_, h_) in enumerate(zip(gs, hs)):
    exp = O.aff_add(g_, h_, p)
    got = (fb(out, 2*i), fb(out, 2*i+1))
    if (exp is None and got != (0, 0)) or (exp is not None and got != exp): bad += 1
print("batch add mismatches", bad)

for N, c in [(1, None), (2, None), (3, 4), (64, None), (64, 7), (50, 5)]:
    sc = O.prng_ints(f"s{N}", N, C.q)
    ctx.set_points(O.points_to_bytes(pts[:N], 48), check_curve=True)
    t = time.time()
    r, info = ctx.run(O.scalars_to_bytes(sc), c=c)
    dt = time.time() - t
    exp = O.msm_batched_affine(sc, pts[:N], c=info["c"])
    print(f"msm N={N} c={info['c']} K={info['K']} rounds={info['rounds']} ok={r.as_tuple() == exp} {dt*1e3:.1f} ms")

# zprize KAT
P = O.ZPRIZE_BLS377_POINT
ctx.set_points(O.points_to_bytes([P, P], 48))
r, _ = ctx.run(O.scalars_to_bytes([2, C.q - 1]))
print("2P + (q-1)P == P:", r.as_tuple() == P)

# generated points / scalars at moderate size, checked through known discrete logs
N = 1 << 14
a = ctx.generate_points(N, seed=7, want_scalars=True)
dev, sb = ctx.generate_scalars(N, seed=9, to_host=True)
t = time.time(); r, info = ctx.run_device(dev, N); dt = time.time() - t
ai = O.scalars_from_bytes(a); si = O.scalars_from_bytes(sb)
exp = O.aff_scale(sum(x * y for x, y in zip(ai, si)) % C.q, G, p)
print(f"msm N=2^14 ok={r.as_tuple() == exp} wall {dt*1e3:.1f} ms", info["phase_ms"], info["rounds"], info["n_pairs"])
for lg in (16, 18, 20, 22, 24, 26):
    N = 1 << lg
    ctx.generate_points(N, seed=7)
    dev, _ = ctx.generate_scalars(N, seed=9)
    t0 = time.time(); ctx.run_device(dev, N); print("  first call", round(time.time() - t0, 3), "s", flush=True)
    t = time.time(); r, info = ctx.run_device(dev, N); dt = time.time() - t
    print(f"msm N=2^{lg} wall {dt*1e3:.1f} ms -> {N/dt:.3e} pts/s", {k: round(v, 2) for k, v in info["phase_ms"].items()}, "c", info["c"], "rounds", info["rounds"], "maxb", info["max_bucket"])
