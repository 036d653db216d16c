"""Randomised differential run of the HIP path against the C oracle (BLS12-377) and the Python oracle (other curves):
random N, window sizes (FUZZ_CS=21,22,23,24 picks them; 0 = the default policy), point multisets with repeats / negations / identities, scalar patterns.  Usage:
    python tools/fuzz_parity.py [seconds] [seed]"""
import random, sys, time
sys.path.insert(0, "/root/repo")
from oracle import msm_oracle as O
from oracle import c_oracle
from montgomery_amd import _lib
from montgomery_amd.api import MsmContext

import os
CS = [int(x) or None for x in os.environ["FUZZ_CS"].split(",")] if os.environ.get("FUZZ_CS") else [None, None, 2, 3, 4, 5, 7, 8, 10, 11, 13, 15, 16, 17, 19, 20]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
c_oracle.load()
curves = [(_lib.CURVE_BLS12_377_G1, O.BLS12_377, 48), (_lib.CURVE_BLS12_381_G1, O.BLS12_381, 48), (_lib.CURVE_PALLAS, O.PALLAS, 32)]
ctxs = {cid: MsmContext(cid) for cid, _, _ in curves}
pools = {cid: O.random_points_bls377(f"fuzz/{B.label}", 300, B)[0] for cid, B, _ in curves}
E = O.ED_ON_BLS12_377
ed_ctx = MsmContext(_lib.CURVE_ED_ON_BLS12_377)
ed_pool = O.random_points_ed377("fuzz/ed", 200)[0]
t0, runs, fails = time.time(), 0, 0
while time.time() - t0 < budget:
    if rnd.random() < 0.2:   # twisted Edwards: msmBasic path, identity is the ordinary point (0, 1)
        n = rnd.choice([1, 2, 3, 17, 64, 100, 255, 300])
        pts = [rnd.choice(ed_pool) if rnd.random() > 0.05 else (0, 1) for _ in range(n)]
        mode = rnd.choice(["uniform", "small", "same", "edge"])
        if mode == "uniform": sc = [rnd.randrange(E.q) for _ in range(n)]
        elif mode == "small": sc = [rnd.randrange(1 << rnd.choice([1, 8, 20])) for _ in range(n)]
        elif mode == "same": sc = [rnd.randrange(E.q)] * n
        else: sc = [rnd.choice([0, 1, 2, E.q - 1, E.q - 2, E.q // 2, 1 << 250]) for _ in range(n)]
        c = rnd.choice([None, 2, 3, 4, 6, 7, 9, 12, 14, 16, 17, 19])
        ed_ctx.set_points(O.points_to_bytes(pts, 32))
        got, info = ed_ctx.run(O.scalars_to_bytes(sc), c=c)
        exp = O.msm_basic_te(sc, pts, c=5)
        runs += 1
        if (got.x, got.y) != exp:
            fails += 1
            print("MISMATCH ed", n, c, mode, info, flush=True)
        continue
    cid, B, nb = rnd.choice(curves)
    big = cid == _lib.CURVE_BLS12_377_G1 and rnd.random() < 0.3
    n = rnd.choice([1, 2, 3, 5, 17, 64, 100, 255, 256, 257, 777, 1000]) if not big else rnd.choice([3000, 5000, 12345])
    pool = pools[cid]
    pts = []
    for _ in range(n):
        r = rnd.random()
        P = rnd.choice(pool)
        if r < 0.05: P = None
        elif r < 0.15 and pts and pts[-1] is not None: P = pts[-1]
        elif r < 0.2 and pts and pts[-1] is not None: P = O.aff_neg(pts[-1], B.p)
        pts.append(P)
    mode = rnd.choice(["uniform", "small", "same", "edge", "top"])
    if mode == "uniform": sc = [rnd.randrange(B.q) for _ in range(n)]
    elif mode == "small": sc = [rnd.randrange(1 << rnd.choice([1, 8, 16, 40])) for _ in range(n)]
    elif mode == "same": sc = [rnd.randrange(B.q)] * n
    elif mode == "edge": sc = [rnd.choice([0, 1, 2, B.q - 1, B.q - 2, B.lam, B.lam + 1, B.q // 2, (1 << 127) - 1, 1 << 127, 1 << 126]) for _ in range(n)]
    else: sc = [B.q - 1 - rnd.randrange(1 << 20) for _ in range(n)]
    c = rnd.choice(CS)
    no_glv = rnd.random() < 0.15
    if no_glv and c is not None and c < 4: c = 4
    ctx = ctxs[cid]
    cb = ctx.coord_bytes
    ctx.set_points(b"".join(b"\0" * (2 * cb) if P is None else P[0].to_bytes(cb, "little") + P[1].to_bytes(cb, "little") for P in pts))
    got, info = ctx.run(O.scalars_to_bytes(sc), c=c, no_glv=no_glv)
    if cid == _lib.CURVE_BLS12_377_G1:
        exp, _ = c_oracle.msm_bls377(O.points_to_bytes([(0, 0) if P is None else P for P in pts], 48), O.scalars_to_bytes(sc), 0)
    else:
        exp = O.msm_batched_affine(sc, pts, B, c=6) if n <= 300 else None
        if exp is None and n > 300: continue
    runs += 1
    if got.as_tuple() != exp:
        fails += 1
        print("MISMATCH", B.label, n, c, mode, no_glv, info, flush=True)
print(f"{runs} runs in {time.time() - t0:.0f} s, {fails} mismatches")
sys.exit(1 if fails else 0)
