"""Randomised skewed scalar sets at mid sizes against the known discrete logs of generated points: random mixtures of zeros,
ones, small values, a few repeated values and uniform scalars, random n in [2^15, 2^22], windows that take every sort path
(one level / radix split and its skew fallback / bin split with slots / window tables).  usage: fuzz_skew.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from montgomery_amd.api import MsmContext
from oracle import c_oracle, msm_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
C = O.BLS12_377
ctx = MsmContext()
t0, runs, fails = time.time(), 0, 0
a = None
while time.time() - t0 < budget:
    lg = int(rng.integers(15, 23))
    n = (1 << lg) - int(rng.integers(0, 1000)) * int(rng.integers(0, 2))
    a = ctx.generate_points(n, seed=int(rng.integers(1, 1 << 30)), want_scalars=True, raw=True)
    for rep in range(3):
        s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        s[:, 31] &= 0x0F
        u = rng.random(n)
        cuts = np.sort(rng.random(4)) * rng.choice([0.2, 0.6, 0.95, 1.0])
        vals = rng.integers(0, 256, size=(int(rng.integers(1, 5)), 32), dtype=np.uint8)
        vals[:, 31] &= 0x0F
        s[u < cuts[0]] = 0
        m = (u >= cuts[0]) & (u < cuts[1]); s[m] = 0; s[m, 0] = 1
        m = (u >= cuts[1]) & (u < cuts[2]); s[m, int(rng.integers(1, 9)):] = 0
        m = (u >= cuts[2]) & (u < cuts[3]); s[m] = vals[rng.integers(0, len(vals), size=int(m.sum()))]
        sb = s.tobytes()
        exp = O.aff_scale(c_oracle.dot_mod(a, sb, n, C.q), (C.gx, C.gy), C.p)
        for c, nt in ((None, False), (None, True), (int(rng.choice([16, 18, 19, 21, 22, 13])), True)):
            r, info = ctx.run(sb, c=c, no_tables=nt)
            runs += 1
            if r.as_tuple() != exp:
                fails += 1
                print("MISMATCH", n, c, nt, [round(float(x), 3) for x in cuts], info, flush=True)
print(f"{runs} runs in {time.time() - t0:.0f} s, {fails} mismatches")
sys.exit(1 if fails else 0)
