"""Degenerate inputs through the big-window paths (three-pass sort, chunk-ordered round 1, element records): 2^LOG2N (default 23)
copies of one point, then P and -P alternating, at c = 22 and the default window, with random scalars and with ONE scalar
repeated (every entry of a window in one bucket).  Expected sums from the scalar sums alone.  usage: adversarial_big.py [LOG2N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import msm_oracle as O
from montgomery_amd.api import MsmContext

C = O.BLS12_377
P = O.ZPRIZE_BLS377_POINT
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 23)
ctx = MsmContext()
rng = np.random.default_rng(23)
sc = rng.integers(0, 256, size=n * 32, dtype=np.uint8); sc[31::32] &= 0x0F          # < 2^252 < q
words = sc.reshape(n, 32)
vals = None
def scalar_sum(signs=None):
    # sum of the 256-bit little-endian scalars (optionally signed), exact, via 16-bit column sums
    cols = words.astype(np.int64)
    if signs is not None: cols = cols * signs[:, None]
    tot = 0
    for j in range(32): tot += int(cols[:, j].sum()) << (8 * j)
    return tot
ok = True
for name, pts, signs in (("same point", [P] * 2, None), ("P, -P alternating", [P, O.aff_neg(P, C.p)], np.tile(np.array([1, -1], dtype=np.int64), n // 2))):
    pb = O.points_to_bytes(pts, 48) * (n // 2)
    ctx.set_points(pb)
    k = scalar_sum(signs) % C.q
    exp = O.aff_scale(k, P, C.p) if k else None
    for c in (22, None):
        r, info = ctx.run(sc.tobytes(), c=c)
        good = r.as_tuple() == exp
        ok &= good
        print(name, "c", info["c"], "max_bucket", info["max_bucket"], "rounds", info["rounds"], "OK" if good else "MISMATCH", flush=True)
# one scalar repeated: every window has a single bucket that holds all 2 n entries of its digit
s0 = int.from_bytes(bytes(words[0]), "little")
one = bytes(words[0]) * n
for c in (22, None):
    r, info = ctx.run(one, c=c)       # points: P, -P alternating -> the sum is the identity for even n
    good = r.as_tuple() is None or r.as_tuple() == None
    ok &= good
    print("one scalar, P / -P", "c", info["c"], "max_bucket", info["max_bucket"], "rounds", info["rounds"], "OK" if good else "MISMATCH", flush=True)
ctx.set_points(O.points_to_bytes([P, P], 48) * (n // 2))
exp = O.aff_scale(s0 * n % C.q, P, C.p)
for c in (22, None):
    r, info = ctx.run(one, c=c)
    good = r.as_tuple() == exp
    ok &= good
    print("one scalar, one point", "c", info["c"], "max_bucket", info["max_bucket"], "rounds", info["rounds"], "OK" if good else "MISMATCH", flush=True)
sys.exit(0 if ok else 1)
