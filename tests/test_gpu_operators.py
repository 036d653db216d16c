"""Operator-level worst cases on the GPU, through the C ABI (`-m gpu`):
  * fe_mul / fe_sqr on unreduced and all-ones-limb operands             src/field.test.ts:27-155 ([0, 2p) with biased values)
  * proj_add / proj_double / proj_add_mixed, te_add (9M), every edge case  src/curve-projective.test.ts:77-208,
                                                                           src/curve-twisted-edwards.test.ts:55-158
  * the tree kernel's plane-reading modes with 1, 2 and 512 pairs per shared inversion  src/curve-affine.ts:376-522
Everything is compared with Python integers / the oracle's spec arithmetic."""
import pytest

from oracle import msm_oracle as O

pytestmark = pytest.mark.gpu

C377 = O.BLS12_377
M30 = (1 << 30) - 1


def to_limbs(v, nl):
    return [(v >> (30 * i)) & M30 if i < nl - 1 else v >> (30 * i) for i in range(nl)]


def from_limbs(l):
    return sum(int(w) << (30 * i) for i, w in enumerate(l))


def worst_case_values(p, nl, tag):
    vals = [0, 1, p - 1, p, p + 1, 2 * p - 1, 4 * p - 3, 63 * p, 64 * p - 1, (1 << ((64 * p).bit_length() - 1)) - 1]
    vals += [v * k + d for v in O.prng_ints("gpu/raw/" + tag, 120, p) for k, d in ((1, 0), (2, 1), (7, 3), (63, 0))]
    return [v for v in vals if v < 64 * p]


def _curve_ctx(curve_id):
    from montgomery_amd.api import MsmContext

    return MsmContext(curve_id)


@pytest.mark.parametrize("name", ["bls12-377", "bls12-381", "pallas", "ed377"])
def test_multiplier_on_unreduced_operands(name):
    from montgomery_amd import _lib

    cid, p, nl = {"bls12-377": (_lib.CURVE_BLS12_377_G1, O.BLS12_377.p, 13), "bls12-381": (_lib.CURVE_BLS12_381_G1, O.BLS12_381.p, 13),
                  "pallas": (_lib.CURVE_PALLAS, O.PALLAS.p, 9), "ed377": (_lib.CURVE_ED_ON_BLS12_377, O.ED_ON_BLS12_377.p, 9)}[name]
    ctx = _curve_ctx(cid)
    R = 1 << (30 * nl)
    rinv = pow(R, -1, p)
    vals = worst_case_values(p, nl, name)
    a = vals
    b = list(reversed(vals))
    ones = R - 1                      # beyond the contract (a b < 2^12 p^2): no accumulator may wrap, congruence must hold
    a2, b2 = a + [ones, ones, 64 * p - 1], b + [ones, 1, ones]
    mul = ctx.test_fp_raw(_lib.OP_MUL, [to_limbs(v, nl) for v in a2], [to_limbs(v, nl) for v in b2])
    sqr = ctx.test_fp_raw(_lib.OP_SQR, [to_limbs(v, nl) for v in a2], [to_limbs(v, nl) for v in a2])
    for i, (x, y) in enumerate(zip(a2, b2)):
        r = from_limbs(mul[i])
        assert all(w <= M30 for w in mul[i][: nl - 1])
        assert r % p == x * y * rinv % p, (name, i)
        assert r < p + x * y // R + 1
        if 2 * x * y < R * p:             # the contract of field.h in its field-independent form: a b < R p / 2  ->  < 1.5 p
            assert r < p + p // 2
        r = from_limbs(sqr[i])
        assert r % p == x * x * rinv % p and r < p + x * x // R + 1, (name, "sqr", i)
    ctx.close()


def _proj_bytes(P, z, p):
    """a projective representative (x z, y z, z) of the affine point P (None: the identity (0, 1, 0) scaled)"""
    X, Y, Z = (0, z % p or 1, 0) if P is None else (P[0] * z % p, P[1] * z % p, z % p)
    cb = 32 if p.bit_length() <= 256 else 48   # coordinate bytes at the ABI are sized per field
    return X.to_bytes(cb, "little") + Y.to_bytes(cb, "little") + Z.to_bytes(cb, "little")


def _proj_affine(b, p):
    cb = len(b) // 3
    X, Y, Z = (int.from_bytes(b[cb * i : cb * i + cb], "little") for i in range(3))
    if Z == 0:
        return None
    zi = pow(Z, -1, p)
    return (X * zi % p, Y * zi % p)


@pytest.mark.parametrize("name", ["bls12-377", "bls12-381", "pallas"])
def test_projective_operators_with_edge_cases(name):
    from montgomery_amd import _lib

    cid, Cv = {"bls12-377": (_lib.CURVE_BLS12_377_G1, O.BLS12_377), "bls12-381": (_lib.CURVE_BLS12_381_G1, O.BLS12_381),
               "pallas": (_lib.CURVE_PALLAS, O.PALLAS)}[name]
    p = Cv.p
    ctx = _curve_ctx(cid)
    G = (Cv.gx, Cv.gy)
    pts = [O.aff_scale(k, G, p) for k in (1, 2, 3, 5, 7, 11, 1234567, Cv.q - 1, Cv.q - 2)]
    zs = O.prng_ints("gpu/proj/z/" + name, 64, p - 1)
    pairs = []
    for i, P in enumerate(pts):
        for Q in (pts[(i + 1) % len(pts)], P, O.aff_neg(P, p), None):      # generic, equal (-> double), opposite (-> zero), identity
            pairs.append((P, Q))
    pairs += [(None, pts[0]), (None, None)]
    pb = b"".join(_proj_bytes(P, zs[i % 64] + 1, p) for i, (P, _) in enumerate(pairs))
    qb = b"".join(_proj_bytes(Q, zs[(i + 7) % 64] + 1, p) for i, (_, Q) in enumerate(pairs))
    cb = ctx.coord_bytes
    pt = 3 * cb
    out = ctx.test_curve_op(_lib_curve_op("add"), pb, qb)
    for i, (P, Q) in enumerate(pairs):
        assert _proj_affine(out[pt * i : pt * i + pt], p) == O.aff_add(P, Q, p), (name, "add", i)
    out = ctx.test_curve_op(_lib_curve_op("double"), pb, qb)
    for i, (P, _) in enumerate(pairs):
        assert _proj_affine(out[pt * i : pt * i + pt], p) == (None if P is None else O.aff_double(P, p)), (name, "double", i)
    # mixed: Q affine (x, y, ignored), the identity as (0, 0)
    qa = b"".join((b"\0" * pt) if Q is None else (Q[0].to_bytes(cb, "little") + Q[1].to_bytes(cb, "little") + (12345).to_bytes(cb, "little"))
                  for _, Q in pairs)
    out = ctx.test_curve_op(_lib_curve_op("mixed"), pb, qa)
    for i, (P, Q) in enumerate(pairs):
        assert _proj_affine(out[pt * i : pt * i + pt], p) == O.aff_add(P, Q, p), (name, "mixed", i)
    ctx.close()


def _lib_curve_op(which):
    return {"add": 0, "double": 1, "mixed": 2}[which]


def test_twisted_edwards_general_addition_9m():
    """te_add (the 9M unified formula on extended points with arbitrary Z) against the oracle's group law, including
    equal operands, opposite operands and the identity (src/curve-twisted-edwards.test.ts:55-158)."""
    from montgomery_amd import _lib

    E = O.ED_ON_BLS12_377
    p = E.p
    ctx = _curve_ctx(_lib.CURVE_ED_ON_BLS12_377)
    G = O.te_from_affine((E.gx, E.gy), E)
    pts = [O.te_to_affine(O.te_scale(k, G, E), E) for k in (1, 2, 3, 5, 99991, E.q - 1)]
    ident = (0, 1)
    zs = O.prng_ints("gpu/te/z", 40, p - 1)

    def ext_bytes(A, z):
        x, y = A
        X, Y, Z, T = x * z % p, y * z % p, z % p, x * y % p * z % p
        return b"".join(v.to_bytes(32, "little") for v in (X, Y, Z, T))

    def ext_affine(b):
        X, Y, Z, T = (int.from_bytes(b[32 * i : 32 * i + 32], "little") for i in range(4))
        zi = pow(Z, -1, p)
        assert T * Z % p == X * Y % p, "T Z = X Y must hold for the output"
        return (X * zi % p, Y * zi % p)

    def add(A, B):
        return O.te_to_affine(O.te_add(O.te_from_affine(A, E), O.te_from_affine(B, E), E), E)

    pairs = []
    for i, P in enumerate(pts):
        neg = ((p - P[0]) % p, P[1])
        for Q in (pts[(i + 1) % len(pts)], P, neg, ident):
            pairs.append((P, Q))
    pairs.append((ident, ident))
    pb = b"".join(ext_bytes(P, zs[i % 40] + 1) for i, (P, _) in enumerate(pairs))
    qb = b"".join(ext_bytes(Q, zs[(i + 3) % 40] + 1) for i, (_, Q) in enumerate(pairs))
    out = ctx.test_curve_op(0, pb, qb)
    for i, (P, Q) in enumerate(pairs):
        assert ext_affine(out[128 * i : 128 * i + 128]) == add(P, Q), ("te add", i)
    out = ctx.test_curve_op(1, pb, qb)
    for i, (P, _) in enumerate(pairs):
        assert ext_affine(out[128 * i : 128 * i + 128]) == add(P, P), ("te double", i)
    ctx.close()


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("steps", [1, 2, 512])
def test_batch_add_plane_modes_and_step_counts(gpu_ctx, mode, steps):
    """G_i + H_i through MODE_REGULAR / MODE_SEARCH of the tree kernel with 1, 2 and 512 pairs per shared inversion:
    generic pairs with P + P, P - P, identity operands and (search mode) missing second operands mixed in, so that the
    forward and backward sweeps of one lane meet every kind of pair."""
    p = C377.p
    n = 3 * 256 * steps // 2 + 37 if steps < 512 else 512 * 256 + 11      # not a multiple of the lane count: idle lanes too
    n = min(n, 140000)
    base, _ = O.random_points_bls377("gpu/modes", 48)
    g, h = [], []
    for i in range(n):
        P, Q = base[i % 48], base[(i * 7 + 3) % 48]
        k = i % 23
        if k == 5: Q = P                      # doubling
        elif k == 9: Q = O.aff_neg(P, p)      # cancels to the identity
        elif k == 13: Q = None                # identity operand (search mode: "no second operand")
        elif k == 17: P = None
        elif k == 21: P, Q = None, None
        g.append(P)
        h.append(Q)
    enc = lambda P: b"\0" * 96 if P is None else P[0].to_bytes(48, "little") + P[1].to_bytes(48, "little")
    out = gpu_ctx.test_batch_add_mode(b"".join(map(enc, g)), b"".join(map(enc, h)), mode, steps)
    memo = {}
    for i in range(n):
        key = (g[i], h[i])
        if key not in memo:
            memo[key] = O.aff_add(g[i], h[i], p)
        got = out[96 * i : 96 * i + 96]
        exp = memo[key]
        assert got == (b"\0" * 96 if exp is None else exp[0].to_bytes(48, "little") + exp[1].to_bytes(48, "little")), (mode, steps, i)


def test_batch_add_gather_mode_every_kind_of_pair(gpu_ctx):
    """The same mix through MODE_GATHER (operands fetched from point rows by slot): generic pairs, P + P, P - P, a
    missing second operand (copy A: a register select), a missing first operand (copy B: the one patch that fetches
    again) and two missing operands, several per wave."""
    p = C377.p
    n = 5003
    base, _ = O.random_points_bls377("gpu/modes/gather", 40)
    g, h = [], []
    for i in range(n):
        P, Q = base[i % 40], base[(i * 11 + 5) % 40]
        k = i % 19
        if k == 3: Q = P
        elif k == 6: Q = O.aff_neg(P, p)
        elif k == 9: Q = None
        elif k == 12: P = None
        elif k == 15: P, Q = None, None
        g.append(P)
        h.append(Q)
    enc = lambda P: b"\0" * 96 if P is None else P[0].to_bytes(48, "little") + P[1].to_bytes(48, "little")
    out = gpu_ctx.test_batch_add(b"".join(map(enc, g)), b"".join(map(enc, h)))
    memo = {}
    for i in range(n):
        key = (g[i], h[i])
        if key not in memo:
            memo[key] = O.aff_add(g[i], h[i], p)
        exp = memo[key]
        assert out[96 * i : 96 * i + 96] == (b"\0" * 96 if exp is None else exp[0].to_bytes(48, "little") + exp[1].to_bytes(48, "little")), i


@pytest.mark.parametrize("kind", ["double", "cancel", "no_a", "no_b", "none"])
def test_one_degenerate_pair_in_one_wave_of_a_workgroup(gpu_ctx, kind):
    """k_batch_add shares ONE inversion among the four waves of a workgroup (batch_add.h): the running products of a lane column
    are multiplied together, so a zero or a wrong denominator in one wave would poison the other three.  768 generic pairs =
    one workgroup x 3 steps in gather mode (msm_test_batch_add walks three pairs per lane); exactly one of them -- wave 2,
    lane 2, the middle step -- is a doubling, a cancellation, or has an identity operand.  Every sum must match, the other
    waves' included.  The second call has two workgroups and the special pair in the last step."""
    p = C377.p
    base, _ = O.random_points_bls377("gpu/one-wave", 64)
    enc = lambda P: b"\0" * 96 if P is None else P[0].to_bytes(48, "little") + P[1].to_bytes(48, "little")
    for n, at in ((768, 256 + 130), (1536, 2 * 512 + 256 + 3 * 64 + 63)):
        g = [base[i % 64] for i in range(n)]
        h = [base[(i * 5 + 1) % 64] for i in range(n)]
        P = g[at]
        g[at], h[at] = {"double": (P, P), "cancel": (P, O.aff_neg(P, p)), "no_a": (None, P), "no_b": (P, None), "none": (None, None)}[kind]
        out = gpu_ctx.test_batch_add(b"".join(map(enc, g)), b"".join(map(enc, h)))
        memo = {}
        for i in range(n):
            key = (g[i], h[i])
            if key not in memo:
                memo[key] = O.aff_add(g[i], h[i], p)
            assert out[96 * i : 96 * i + 96] == enc(memo[key]), (kind, n, i)


def test_bucket_reduction_projective_and_all_affine(gpu_ctx):
    """P_k = sum_l l B_(k,l) by the projective reduction the MSM uses (mode 0) and by the reference's
    all-affine reduction (reduceBucketsAffine, src/msm-batched-affine-single-thread.ts:522-667: in-place batched-affine
    additions and doublings, every chunk size 2^c0) against the oracle's direct sum.  Buckets include empty ones, a repeated
    point (P + P inside the in-place rounds) and P next to -P (a sum that cancels)."""
    from oracle import msm_oracle as O

    C = O.BLS12_377
    K, L = 3, 64
    pts, _ = O.random_points_bls377("op/reduce", K * L)
    for i in (5, 17, 40, 64 + 9, 128 + 63):
        pts[i] = None                                       # empty buckets
    pts[10] = pts[11] = pts[12]                             # equal neighbours
    pts[64 + 20] = O.aff_neg(pts[64 + 21], C.p)             # P and -P side by side
    pts[128 + 0] = None
    pts[128 + 1] = None
    buckets = b"".join(bytes(96) if P is None else P[0].to_bytes(48, "little") + P[1].to_bytes(48, "little") for P in pts)
    exp = []
    for k in range(K):
        acc = None
        for l in range(1, L + 1):
            B = pts[k * L + l - 1]
            if B is not None:
                acc = O.aff_add(acc, O.aff_scale(l, B, C.p), C.p)
        exp.append(acc)

    def affine(parts):
        out = []
        for k in range(K):
            X, Y, Z = (int.from_bytes(parts[144 * k + 48 * j: 144 * k + 48 * j + 48], "little") for j in range(3))
            out.append(None if Z == 0 else (X * pow(Z, -1, C.p) % C.p, Y * pow(Z, -1, C.p) % C.p))
        return out

    got0, ms0 = gpu_ctx.test_bucket_reduce(buckets, K, L, mode=0)
    assert affine(got0) == exp
    for c0 in (0, 1, 2, 3, 6):                              # c0 = 6: one chunk (the plain running sum), c0 = 0: tree only
        got1, ms1 = gpu_ctx.test_bucket_reduce(buckets, K, L, mode=1, c0=c0)
        assert affine(got1) == exp, c0
    assert ms0 > 0 and ms1 > 0
