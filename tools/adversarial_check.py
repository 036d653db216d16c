import sys, time
sys.path.insert(0, "/root/repo")
from oracle import msm_oracle as O
from montgomery_amd.api import MsmContext
C = O.BLS12_377
ctx = MsmContext()
P = O.ZPRIZE_BLS377_POINT
n = 1 << 18
pb = O.points_to_bytes([P], 48) * n
ctx.set_points(pb)
# all scalars equal, all points equal: one bucket per window holding everything, doubling at every tree level
s = 0x1234567890abcdef1234567890abcdef1234567890abcdef12345
t = time.time(); r, info = ctx.run(O.scalars_to_bytes([s]) * n); dt = time.time() - t
print("same point same scalar", r.as_tuple() == O.aff_scale(s * n % C.q, P, C.p), round(dt * 1e3, 1), "ms", info["rounds"], info["max_bucket"])
# random scalars, same point
sc = O.prng_ints("adv", n, C.q)
t = time.time(); r, info = ctx.run(O.scalars_to_bytes(sc)); dt = time.time() - t
print("same point random scalars", r.as_tuple() == O.aff_scale(sum(sc) % C.q, P, C.p), round(dt * 1e3, 1), "ms", info["rounds"], info["max_bucket"])
# alternating P, -P with equal scalars -> identity
pts = [P, O.aff_neg(P, C.p)] * (n // 2)
ctx.set_points(O.points_to_bytes(pts, 48))
r, info = ctx.run(O.scalars_to_bytes([s]) * n)
print("P,-P cancel", r.isZero, info["rounds"])
