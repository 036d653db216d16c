"""Bucket reduction alone, projective (as shipped) against the reference's all-affine reduction, at the bucket counts of the
2^20 and 2^26 configurations and of a c = 22 run: python tools/reduce_ab.py   """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montgomery_amd.api import MsmContext
ctx = MsmContext()
for K, cl in ((8, 15), (3, 18), (3, 21)):
    L = 1 << cl
    n = K * L
    ctx.generate_points(n, seed=5)
    buckets = ctx.get_points(0, n)
    ref, ms0 = ctx.test_bucket_reduce(buckets, K, L, mode=0)
    ref, ms0 = ctx.test_bucket_reduce(buckets, K, L, mode=0)
    line = f"K = {K}, L = 2^{cl}: projective {ms0:.3f} ms;  all-affine"
    for c0 in (1, 2, 3, 4):
        got, ms1 = ctx.test_bucket_reduce(buckets, K, L, mode=1, c0=c0)
        got, ms1 = ctx.test_bucket_reduce(buckets, K, L, mode=1, c0=c0)
        X0, Y0, Z0 = (int.from_bytes(ref[48 * j:48 * j + 48], "little") for j in range(3))
        X1, Y1, Z1 = (int.from_bytes(got[48 * j:48 * j + 48], "little") for j in range(3))
        from oracle.msm_oracle import BLS12_377 as C
        same = (X0 * Z1 - X1 * Z0) % C.p == 0 and (Y0 * Z1 - Y1 * Z0) % C.p == 0
        line += f"  c0={c0}: {ms1:.3f} ms{'' if same else ' MISMATCH'}"
    print(line, flush=True)
