"""CPU oracle for the MSM hot path (TEST INFRASTRUCTURE ONLY -- never imported by the product).

Pure-Python-int restatement of the reference's algorithms for the one path this repo accelerates
(SURVEY.md section 8).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; ``montgomery_amd`` itself must not.

Each function cites the reference file:line (relative to the upstream checkout) that it follows.
The reference (TypeScript + run-time generated WebAssembly via wasmati 0.2.0) cannot be compiled
or run in this image (node 12, no tsc, no node_modules), so the oracle is pinned against the
known-answer material the reference's own tests hold -- see tests/test_oracle_kat.py:
  * generators / curve constants        src/concrete/bls12-377.params.ts:11-45,
                                         src/concrete/ed-on-bls12-377.params.ts:5-31
  * lambda / beta endomorphism checks   src/concrete/bls12-377.params.ts:49-63
  * fixed-point MSM identities          scripts/zprize23/submission-test-bls377.ts:6-45,
                                         scripts/zprize23/submission-test.ts:5-21
  * hard-coded field identities         src/bigint/field.test.ts:61-104
  * algebraic MSM identities            src/bigint/msm.test.ts:18-101
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

# --------------------------------------------------------------------------------------
# constants: src/concrete/bls12-377.params.ts:11-34, src/concrete/ed-on-bls12-377.params.ts:5-22
# --------------------------------------------------------------------------------------


@dataclass(frozen=True)
class WeierstrassParams:
    label: str
    p: int
    q: int
    h: int
    b: int
    gx: int
    gy: int
    lam: int
    beta: int
    n_bytes: int  # packed bytes per coordinate on the wire (src/parallel.ts:103-112)


@dataclass(frozen=True)
class TwistedEdwardsParams:
    label: str
    p: int
    q: int
    h: int
    d: int
    gx: int
    gy: int
    n_bytes: int


BLS12_377 = WeierstrassParams(
    label="bls12-377",
    p=0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001,
    q=0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001,
    h=0x170B5D44300000000000000000000000,
    b=1,
    gx=0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF,
    gy=0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6,
    lam=0x12AB655E9A2CA55660B44D1E5C37B00114885F32400000000000000000000000,
    beta=0x1AE3A4617C510EABC8756BA8F8C524EB8882A75CC9BC8E359064EE822FB5BFFD1E945779FFFFFFFFFFFFFFFFFFFFFFF,
    n_bytes=48,
)

BLS12_381 = WeierstrassParams(  # src/concrete/bls12-381.params.ts:6-55 (lambda2 = z^2 - 1, beta2)
    label="bls12-381",
    p=0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
    q=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
    h=0x396C8C005555E1568C00AAAB0000AAAB,
    b=4,
    gx=0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    gy=0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    lam=0xD201000000010000 ** 2 - 1,
    beta=0x1A0111EA397FE699EC02408663D4DE85AA0D857D89759AD4897D29650FB85F9B409427EB4F49FFFD8BFD00000000AAAC,
    n_bytes=48,
)

_PALLAS_P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
_PALLAS_Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
PALLAS = WeierstrassParams(  # src/concrete/pasta.params.ts:10-53
    label="pallas",
    p=_PALLAS_P,
    q=_PALLAS_Q,
    h=1,
    b=5,
    gx=1,
    gy=0x1B74B5A30A12937C53DFA9F06378EE548F655BD4333D477119CF7A23CAED2ABB,
    lam=pow(5, (_PALLAS_Q - 1) // 3, _PALLAS_Q),                     # :24
    beta=pow(pow(5, (_PALLAS_P - 1) // 3, _PALLAS_P), 2, _PALLAS_P),  # :33-34
    n_bytes=32,
)

ED_ON_BLS12_377 = TwistedEdwardsParams(
    label="ed-on-bls12-377",
    p=0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001,
    q=0x4AAD957A68B2955982D1347970DEC005293A3AFC43C8AFEB95AEE9AC33FD9FF,
    h=4,
    d=3021,
    gx=0x9F1B5A5BAF6ACF06FED91C9AE9EBFA06068DD2835790980894E2328F3EBCA05,
    gy=0x9A20DF36571AC3CD906B256080BA8454453C177AAF3131BB50A67BF1A806781,
    n_bytes=32,
)

# fixed test points of the ZPrize self-tests
ZPRIZE_BLS377_POINT = (  # scripts/zprize23/submission-test-bls377.ts:6-10
    111871295567327857271108656266735188604298176728428155068227918632083036401841336689521497731900230387779623820740,
    76860045326390600098227152997486448974650822224305058012700629806287380625419427989664237630603922765089083164740,
)
ZPRIZE_ED377_POINT = (  # scripts/zprize23/submission-test.ts:5-10 (x, y, t; z = 1)
    2796670805570508460920584878396618987767121022598342527208237783066948667246,
    8134280397689638111748378379571739274369602049665521098046934931245960532166,
    3446088593515175914550487355059397868296219355049460558182099906777968652023,
)


# --------------------------------------------------------------------------------------
# prime field: src/bigint/field.ts:12-187
# --------------------------------------------------------------------------------------


def ceil_log2(n: int) -> int:
    """`log2` of src/util.ts:138 -- number of bits needed, i.e. ceil(log2(n)) for n >= 1."""
    return (n - 1).bit_length() if n > 1 else 0


def bit_len(n: int) -> int:
    return n.bit_length()


def inv_mod(a: int, p: int) -> int:
    a %= p
    if a == 0:
        raise ZeroDivisionError("inverse of 0")  # wasm traps: src/wasm/inverse.ts:198-199
    return pow(a, -1, p)


def is_square(a: int, p: int) -> bool:
    a %= p
    return a == 0 or pow(a, (p - 1) // 2, p) == 1


def sqrt_mod(a: int, p: int) -> Optional[int]:
    """Tonelli-Shanks (the reference's src/field-sqrt.ts is only used to sample points)."""
    a %= p
    if a == 0:
        return 0
    if not is_square(a, p):
        return None
    m, t = 0, p - 1
    while t % 2 == 0:
        t //= 2
        m += 1
    z = 2
    while is_square(z, p):
        z += 1
    c = pow(z, t, p)
    x = pow(a, (t + 1) // 2, p)
    b = pow(a, t, p)
    while b != 1:
        i, b2 = 0, b
        while b2 != 1:
            b2 = b2 * b2 % p
            i += 1
        g = pow(c, 1 << (m - i - 1), p)
        x = x * g % p
        c = g * g % p
        b = b * c % p
        m = i
    return x


# --------------------------------------------------------------------------------------
# affine short Weierstrass, a = 0: src/bigint/affine-weierstrass.ts:29-175
# points are (x, y) tuples, None = zero/infinity
# --------------------------------------------------------------------------------------

AffinePoint = Optional[Tuple[int, int]]


def aff_double(P: AffinePoint, p: int) -> AffinePoint:
    # src/bigint/affine-weierstrass.ts:72-83 (+ the y = 0 order-2 case, which gives zero)
    if P is None:
        return None
    x, y = P
    if y % p == 0:
        return None
    m = 3 * x * x * inv_mod(2 * y, p) % p
    x2 = (m * m - 2 * x) % p
    y2 = (m * (x - x2) - y) % p
    return (x2, y2)


def aff_add(P1: AffinePoint, P2: AffinePoint, p: int) -> AffinePoint:
    # src/bigint/affine-weierstrass.ts:44-67
    if P1 is None:
        return P2
    if P2 is None:
        return P1
    x1, y1 = P1
    x2, y2 = P2
    if (x1 - x2) % p == 0:
        if (y1 - y2) % p == 0:
            return aff_double(P1, p)
        return None
    m = (y2 - y1) * inv_mod(x2 - x1, p) % p
    x3 = (m * m - x1 - x2) % p
    y3 = (m * (x1 - x3) - y1) % p
    return (x3, y3)


def aff_neg(P: AffinePoint, p: int) -> AffinePoint:
    if P is None:
        return None
    return (P[0], (-P[1]) % p)


def aff_scale(s: int, P: AffinePoint, p: int) -> AffinePoint:
    # src/bigint/affine-weierstrass.ts:110-118 (MSB-first double-and-add)
    Q = None
    for i in range(s.bit_length() - 1, -1, -1):
        Q = aff_double(Q, p)
        if (s >> i) & 1:
            Q = aff_add(Q, P, p)
    return Q


def aff_is_on_curve(P: AffinePoint, C: WeierstrassParams) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - (x * x * x + C.b)) % C.p == 0


# --------------------------------------------------------------------------------------
# homogeneous projective Weierstrass: src/bigint/projective-weierstrass.ts:18-232
# (X, Y, Z); Z = 0 is zero
# --------------------------------------------------------------------------------------

ProjPoint = Tuple[int, int, int]
PROJ_ZERO: ProjPoint = (0, 1, 0)


def proj_double(P: ProjPoint, p: int) -> ProjPoint:
    # dbl-1998-cmo-2, src/bigint/projective-weierstrass.ts:90-126
    X1, Y1, Z1 = P
    if Z1 % p == 0:
        return PROJ_ZERO
    w = 3 * X1 * X1 % p
    s = Y1 * Z1 % p
    ss = s * s % p
    sss = s * ss % p
    R = Y1 * s % p
    B = X1 * R % p
    h = (w * w - 8 * B) % p
    X3 = 2 * h * s % p
    Y3 = (w * (4 * B - h) - 8 * R * R) % p
    Z3 = 8 * sss % p
    return (X3, Y3, Z3)


def proj_add(P1: ProjPoint, P2: ProjPoint, p: int) -> ProjPoint:
    # add-1998-cmo-2 with edge cases, src/bigint/projective-weierstrass.ts:33-85
    X1, Y1, Z1 = P1
    X2, Y2, Z2 = P2
    if Z1 % p == 0:
        return P2
    if Z2 % p == 0:
        return P1
    Y1Z2 = Y1 * Z2 % p
    X1Z2 = X1 * Z2 % p
    Z1Z2 = Z1 * Z2 % p
    u = (Y2 * Z1 - Y1Z2) % p
    v = (X2 * Z1 - X1Z2) % p
    if v == 0:
        if u == 0:
            return proj_double(P1, p)
        return PROJ_ZERO
    uu = u * u % p
    vv = v * v % p
    vvv = v * vv % p
    R = vv * X1Z2 % p
    A = (uu * Z1Z2 - vvv - 2 * R) % p
    X3 = v * A % p
    Y3 = (u * (R - A) - vvv * Y1Z2) % p
    Z3 = vvv * Z1Z2 % p
    return (X3, Y3, Z3)


def proj_from_affine(P: AffinePoint) -> ProjPoint:
    return PROJ_ZERO if P is None else (P[0], P[1], 1)


def proj_to_affine(P: ProjPoint, p: int) -> AffinePoint:
    # src/curve-projective.ts:335-349 / src/bigint/projective-weierstrass.ts (toAffine)
    X, Y, Z = P
    if Z % p == 0:
        return None
    zi = inv_mod(Z, p)
    return (X * zi % p, Y * zi % p)


# --------------------------------------------------------------------------------------
# twisted Edwards a = -1, extended coordinates: src/bigint/twisted-edwards.ts:28-211
# --------------------------------------------------------------------------------------

TEPoint = Tuple[int, int, int, int]  # X, Y, Z, T
TE_ZERO: TEPoint = (0, 1, 1, 0)  # src/bigint/twisted-edwards.ts:34


def te_add(P1: TEPoint, P2: TEPoint, C: TwistedEdwardsParams) -> TEPoint:
    # add-2008-hwcd-3 with k = 2d, src/bigint/twisted-edwards.ts:52-85
    p = C.p
    k = 2 * C.d
    X1, Y1, Z1, T1 = P1
    X2, Y2, Z2, T2 = P2
    A = (Y1 - X1) * (Y2 - X2) % p
    B = (Y1 + X1) * (Y2 + X2) % p
    Cc = T1 * T2 % p * k % p
    D = 2 * Z1 * Z2 % p
    E = (B - A) % p
    F = (D - Cc) % p
    G = (D + Cc) % p
    H = (B + A) % p
    return (E * F % p, G * H % p, F * G % p, E * H % p)


def te_neg(P: TEPoint, C: TwistedEdwardsParams) -> TEPoint:
    X, Y, Z, T = P
    return ((-X) % C.p, Y, Z, (-T) % C.p)


def te_scale(s: int, P: TEPoint, C: TwistedEdwardsParams) -> TEPoint:
    Q = TE_ZERO
    for i in range(s.bit_length() - 1, -1, -1):
        Q = te_add(Q, Q, C)
        if (s >> i) & 1:
            Q = te_add(Q, P, C)
    return Q


def te_from_affine(xy: Tuple[int, int], C: TwistedEdwardsParams) -> TEPoint:
    x, y = xy
    return (x % C.p, y % C.p, 1, x * y % C.p)


def te_to_affine(P: TEPoint, C: TwistedEdwardsParams) -> Tuple[int, int]:
    X, Y, Z, _ = P
    zi = inv_mod(Z, C.p)
    return (X * zi % C.p, Y * zi % C.p)


def te_is_on_curve(P: TEPoint, C: TwistedEdwardsParams) -> bool:
    # src/bigint/twisted-edwards.ts:150-158
    X, Y, Z, T = P
    p = C.p
    if Z % p == 0 or (T * Z - X * Y) % p != 0:
        return False
    return (-X * X + Y * Y - Z * Z - C.d * T * T) % p == 0


# --------------------------------------------------------------------------------------
# GLV: lattice src/glv/glv.ts:21-50, constants + decomposition src/wasm/glv.ts:35-229
# --------------------------------------------------------------------------------------


def egcd_stop_early(lam: int, q: int) -> Tuple[Tuple[int, int], Tuple[int, int]]:
    """src/glv/glv.ts:21-50: returns V = [[v00, v01], [v10, v11]] with v0j + lam*v1j = 0 mod q."""
    assert lam <= q
    r0, r1 = q, lam
    t0, t1 = 0, 1
    while r1 * r1 > q:
        quo = r0 // r1
        r0, r1 = r1, r0 - quo * r1
        t0, t1 = t1, t0 - quo * t1
    quo = r0 // r1
    r2 = r0 - quo * r1
    t2 = t0 - quo * t1
    v00, v10 = r1, -t1
    if max(r0, abs(t0)) <= max(r2, abs(t2)):
        v01, v11 = r0, -t0
    else:
        v01, v11 = r2, -t2
    return ((v00, v01), (v10, v11))


def _trunc_div(a: int, b: int) -> int:
    """JS bigint division truncates toward zero (src/wasm/glv.ts:47-48 rely on it)."""
    qd = abs(a) // abs(b)
    return qd if (a >= 0) == (b >= 0) else -qd


@dataclass(frozen=True)
class GlvParams:
    q: int
    lam: int
    w: int
    n: int  # limbs of a full scalar
    n0: int  # limbs of a half scalar
    m: int  # bits dropped after the m_j multiplication
    k: int  # bits of s dropped before the multiplication
    v00: int
    v01: int
    v10: int
    v11: int
    m0: int
    m1: int
    max_bits: int


def glv_params(q: int, lam: int, w: int = 29) -> GlvParams:
    """Constants as derived by `glvGeneral`, src/wasm/glv.ts:35-63 and :216-228 (maxBits)."""
    from fractions import Fraction

    n = -(-(q.bit_length()) // w)  # createScalar: limbs for q at word size w
    n0 = -(-n // 2)
    m = n0 * w
    k = (n - n0) * w
    (v00, v01), (v10, v11) = egcd_stop_early(lam, q)
    det = v00 * v11 - v10 * v01
    m0 = _trunc_div((1 << (m + k)) * -v11, det)
    m1 = _trunc_div((1 << (m + k)) * v10, det)
    lim = 1 << m
    assert max(v00, v01, v10, v11) < lim and m0 < lim and m1 < lim
    # error bounds, src/wasm/glv.ts:216-226 (exact rationals instead of JS doubles)
    m0res = (1 << (m + k)) * -v11 - m0 * det
    m1res = (1 << (m + k)) * v10 - m1 * det
    m0err = abs(Fraction(m0res, det))
    m1err = abs(Fraction(m1res, det))
    x0err = Fraction(1, 2) + Fraction(m0, 1 << m) + m0err * Fraction(q, 1 << (m + k))
    x1err = Fraction(1, 2) + Fraction(m1, 1 << m) + m1err * Fraction(q, 1 << (m + k))
    max_s0 = abs(x0err * abs(v00)) + abs(x1err * abs(v01))
    max_s1 = abs(x0err * abs(v10)) + abs(x1err * abs(v11))

    def clog2(fr: Fraction) -> int:
        v = -(-fr.numerator // fr.denominator)  # ceil
        return ceil_log2(max(int(v), 1))

    max_bits = max(clog2(max_s0), clog2(max_s1))
    return GlvParams(q, lam, w, n, n0, m, k, v00, v01, v10, v11, m0, m1, max_bits)


def glv_decompose(s: int, G: GlvParams) -> Tuple[int, int, bool, bool]:
    """`decompose`, src/wasm/glv.ts:68-169: returns (|s0|, |s1|, s0<0, s1<0), s = s0 + s1*lam mod q.

    x_j = round(|m_j| * (s >> k) / 2^m) with round-half-up exactly as `multiplyMsb` (:187-214) does
    (bit m-1 of the product is tested), the sign of m_j applied afterwards (:101-102);
    s0 = s + v00*x0 + v01*x1, s1 = v10*x0 + v11*x1, both truncated to n limbs two's complement
    and sign-flipped when negative (:123-160).
    """
    s_hi = s >> G.k

    def mul_msb(x: int, mabs: int) -> int:
        prod = x * mabs
        return (prod >> G.m) + ((prod >> (G.m - 1)) & 1)

    x0 = mul_msb(s_hi, abs(G.m0)) * (1 if G.m0 >= 0 else -1)
    x1 = mul_msb(s_hi, abs(G.m1)) * (1 if G.m1 >= 0 else -1)
    s0 = s + G.v00 * x0 + G.v01 * x1
    s1 = G.v10 * x0 + G.v11 * x1
    lim = 1 << (G.n * G.w)
    # the wasm code traps (`unreachable`) if the carry out is not in {0, -1}: :130-131
    assert -lim < s0 < lim and -lim < s1 < lim
    return (abs(s0), abs(s1), s0 < 0, s1 < 0)


# --------------------------------------------------------------------------------------
# window slicing: src/msm-batched-affine.ts:175-203, src/msm-basic.ts:72-91,
# extractBitSlice src/wasm/field-helpers.ts:307-358, window table src/msm-common.ts:8-41
# --------------------------------------------------------------------------------------

_WINDOW_TABLE = {  # src/msm-common.ts:25-41
    "large": {14: 13, 15: 14, 16: 14, 17: 14, 18: 14, 19: 18, 20: 18},
    "small": {16: 12},
}


def almost_inverse_wordsliced(a: int, p: int, w: int, n: int, hi_bits: int = 63) -> Tuple[int, int, bool]:
    """The reference's experimental word-sliced almost-inverse, src/inverse/faster-inverse.ts:78-177 (wasm twin:
    src/inverse/faster-inverse-wasm.ts:133-343): Kaliski's binary gcd with w steps at a time decided on the low words
    and on `hi_bits`-bit approximations of the high ends, applied to the full values as a 2x2 matrix.
    Returns (s, k, sign_flip) with a * s = 2^k (mod p), 0 <= |s| < p after the final `makeOdd` step."""
    u, v, r, s, k = p, a, 0, 1, 0
    flip = False
    for _ in range(2 * n):
        f0, g0, f1, g1 = 1, 0, 0, 1
        ulo, vlo = u & ((1 << w) - 1), v & ((1 << w) - 1)
        shift = max(max(u.bit_length(), v.bit_length()) - hi_bits, 0)
        uhi, vhi = u >> shift, v >> shift
        for _ in range(w):
            if ulo & 1 == 0:
                uhi >>= 1; ulo >>= 1; f1 <<= 1; g1 <<= 1
            elif vlo & 1 == 0:
                vhi >>= 1; vlo >>= 1; f0 <<= 1; g0 <<= 1
            else:
                mhi = vhi - uhi
                if mhi <= 0:
                    uhi = -mhi >> 1; ulo = (ulo - vlo) >> 1
                    f0 += f1; g0 += g1; f1 <<= 1; g1 <<= 1
                else:
                    vhi = mhi >> 1; vlo = (vlo - ulo) >> 1
                    f1 += f0; g1 += g0; f0 <<= 1; g0 <<= 1
            k += 1
        unew, vnew = u * f0 - v * g0, v * g1 - u * f1
        assert unew & ((1 << w) - 1) == 0 and vnew & ((1 << w) - 1) == 0
        u, v = unew >> w, vnew >> w
        if u < 0:
            flip = True
            u, f0, g0 = -u, -f0, -g0
        if v < 0:
            flip = True
            v, f1, g1 = -v, -f1, -g1
        r, s = r * f0 + s * g0, r * f1 + s * g1
        if u == 0:
            break
        if v == 0:
            raise ValueError("v = 0: the input is not invertible")
    i = 0
    while i < w and s & 1 == 0:   # makeOdd: the last batch kept doubling s after u had reached 0
        s >>= 1
        k -= 1
        i += 1
    return s, k, flip


def window_size_reference(field_bits: int, n: int) -> int:
    """`windowSize`, src/msm-common.ts:8-13."""
    table = _WINDOW_TABLE["large" if field_bits > 260 else "small"]
    return table.get(n, max(n - 1, 1))


def signed_digits(s: int, c: int, K: int) -> List[Tuple[int, bool]]:
    """Signed c-bit recoding: list of (magnitude l in [0, L], negative?) per window,
    src/msm-batched-affine.ts:183-193.  Sum_k (-1)^neg l_k 2^(ck) == s."""
    L = 1 << (c - 1)
    mask = (1 << c) - 1
    out = []
    carry = 0
    for kk in range(K):
        l = ((s >> (kk * c)) & mask) + carry
        if l > L:
            l = 2 * L - l
            carry = 1
        else:
            carry = 0
        out.append((l, carry == 1))
    assert carry == 0, "top window overflow (K*c must be >= bits + 1)"
    return out


# --------------------------------------------------------------------------------------
# spec MSM (naive Pippenger): src/bigint/msm.ts:8-53
# --------------------------------------------------------------------------------------


def msm_spec_affine(scalars: Sequence[int], points: Sequence[AffinePoint], C: WeierstrassParams) -> AffinePoint:
    """Definition-level answer sum_i s_i * P_i via the unsigned-window bucket method of
    src/bigint/msm.ts:8-53 (c = max(log2 N - 1, 1), K = ceil(b / c))."""
    p = C.p
    N = len(scalars)
    assert N == len(points)
    if N == 0:
        return None
    b = C.q.bit_length()
    c = max(ceil_log2(N) - 1, 1)
    K = -(-b // c)
    L = 1 << c
    part = []
    for kk in range(K):
        buckets: List[AffinePoint] = [None] * (L - 1)
        for s, P in zip(scalars, points):
            l = (s >> (kk * c)) & (L - 1)
            if l:
                buckets[l - 1] = aff_add(buckets[l - 1], P, p)
        run = tri = None
        for l in range(L - 2, -1, -1):
            run = aff_add(run, buckets[l], p)
            tri = aff_add(tri, run, p)
        part.append(tri)
    res = part[K - 1]
    for kk in range(K - 2, -1, -1):
        for _ in range(c):
            res = aff_double(res, p)
        res = aff_add(res, part[kk], p)
    return res


def msm_naive_affine(scalars: Sequence[int], points: Sequence[AffinePoint], C: WeierstrassParams) -> AffinePoint:
    """Plain sum of double-and-add scalings (independent cross-check of the bucket methods)."""
    acc = None
    for s, P in zip(scalars, points):
        acc = aff_add(acc, aff_scale(s % C.q, P, C.p), C.p)
    return acc


# --------------------------------------------------------------------------------------
# the hot path: batched-affine Pippenger with GLV, src/msm-batched-affine.ts:69-340
# --------------------------------------------------------------------------------------


def batch_add_affine(G: List[AffinePoint], H: List[AffinePoint], p: int, safe: bool = True) -> List[AffinePoint]:
    """S_i = G_i + H_i with ONE inversion for the whole batch.

    Unsafe form: `batchAddUnsafeNew`, src/curve-affine.ts:463-522 (prefix products of dx, one
    inverse, backward sweep, `addAffinePacked` src/wasm/curve.ts:63-84).
    Safe form: `batchAddNew`, src/curve-affine.ts:376-458 (classification of zero / equal /
    opposite inputs, doubling through the same inversion with denominator 2y).
    """
    n = len(G)
    out: List[AffinePoint] = [None] * n
    den: List[int] = [1] * n
    kind = [0] * n  # 0 = generic add, 1 = double, 2 = result already known
    for i in range(n):
        g, h = G[i], H[i]
        if safe:
            if g is None:
                out[i], kind[i] = h, 2
                continue
            if h is None:
                out[i], kind[i] = g, 2
                continue
            if (g[0] - h[0]) % p == 0:
                if (g[1] - h[1]) % p == 0 and g[1] % p != 0:
                    kind[i], den[i] = 1, 2 * g[1] % p
                else:
                    out[i], kind[i] = None, 2
                continue
        den[i] = (h[0] - g[0]) % p
    # Montgomery's trick: src/curve-affine.ts:484-516 / src/wasm/inverse.ts:220-271
    prefix = [1] * (n + 1)
    for i in range(n):
        prefix[i + 1] = prefix[i] * den[i] % p
    inv = inv_mod(prefix[n], p) if n else 1
    for i in range(n - 1, -1, -1):
        d = inv * prefix[i] % p
        inv = inv * den[i] % p
        if kind[i] == 2:
            continue
        x1, y1 = G[i]
        x2, y2 = H[i]
        if kind[i] == 1:
            m = 3 * x1 * x1 * d % p
        else:
            m = (y2 - y1) * d % p
        x3 = (m * m - x1 - x2) % p
        y3 = (m * (x1 - x3) - y1) % p  # same point as the reference's (x2 - x3)*m - y2 form
        out[i] = (x3, y3)
    return out


def reduce_buckets_column_projective(buckets: List[ProjPoint], lstart: int, p: int) -> ProjPoint:
    """`reduceBucketsColumnProjective`, src/msm-batched-affine.ts:556-583:
    sum_l (lstart + l) * buckets[l] = triangle + (lstart - 1) * row."""
    tri = row = PROJ_ZERO
    for l in range(len(buckets) - 1, -1, -1):
        row = proj_add(row, buckets[l], p)
        tri = proj_add(tri, row, p)
    ls = lstart - 1
    while True:
        if ls & 1:
            tri = proj_add(tri, row, p)
        ls >>= 1
        if ls == 0:
            break
        row = proj_double(row, p)
    return tri


def msm_batched_affine(
    scalars: Sequence[int],
    points: Sequence[AffinePoint],
    C: WeierstrassParams = BLS12_377,
    c: Optional[int] = None,
    safe: bool = True,
    n_chunks: int = 1,
    glv: Optional[GlvParams] = None,
) -> AffinePoint:
    """The batched-affine GLV Pippenger of `createMsm().msm`, src/msm-batched-affine.ts:69-340,
    phase by phase (single 'thread'; `n_chunks` splits each window's bucket range like
    `computeBucketsSplit` :626-667 does across threads).  Returns the canonical affine result."""
    p = C.p
    N = len(scalars)
    assert N == len(points)
    if N == 0:
        return None
    glv = glv or glv_params(C.q, C.lam)
    b = glv.max_bits
    if c is None:
        c = window_size_reference(p.bit_length(), ceil_log2(N))
    K = -(-(b + 1) // c)
    L = 1 << (c - 1)

    # prep 1 (:350-421): GLV split, signs folded into the points A = sign(s0) G, B = sign(s1) phi(G)
    halves: List[int] = []
    half_points: List[AffinePoint] = []
    for s, P in zip(scalars, points):
        a0, a1, n0, n1 = glv_decompose(s, glv)
        endo = None if P is None else (C.beta * P[0] % p, P[1])  # src/wasm/curve.ts:90-103
        halves += [a0, a1]
        half_points += [aff_neg(P, p) if n0 else P, aff_neg(endo, p) if n1 else endo]

    # slice + count (:175-203), scatter (:456-502): bucket lists per (k, l)
    buckets: List[List[List[AffinePoint]]] = [[[] for _ in range(L + 1)] for _ in range(K)]
    for s, P in zip(halves, half_points):
        for kk, (l, neg) in enumerate(signed_digits(s, c, K)):
            if l:
                buckets[kk][l].append(aff_neg(P, p) if neg else P)

    # bucket accumulation (:243-282): rounds m = 1, 2, 4, ... of independent pair additions
    flat = [bk for kk in range(K) for bk in buckets[kk][1:]]
    max_size = max((len(bk) for bk in flat), default=0)
    m = 1
    while m < max_size:
        Gs, Hs, where = [], [], []
        for bk in flat:
            j = 0
            while j + m < len(bk):
                Gs.append(bk[j])
                Hs.append(bk[j + m])
                where.append((bk, j))
                j += 2 * m
        sums = batch_add_affine(Gs, Hs, p, safe=safe)
        for (bk, j), S in zip(where, sums):
            bk[j] = S
        m *= 2

    # bucket reduction (:289-299) in curve-projective, then partition + final sums (:312-333)
    part: List[ProjPoint] = []
    for kk in range(K):
        sums_k = [proj_from_affine(bk[0]) if bk else PROJ_ZERO for bk in buckets[kk][1:]]
        acc = PROJ_ZERO
        step = -(-L // n_chunks)
        for lstart in range(1, L + 1, step):
            chunk = sums_k[lstart - 1 : lstart - 1 + step]
            acc = proj_add(acc, reduce_buckets_column_projective(chunk, lstart, p), p)
        part.append(acc)
    res = part[K - 1]
    for kk in range(K - 2, -1, -1):
        for _ in range(c):
            res = proj_double(res, p)
        res = proj_add(res, part[kk], p)
    return proj_to_affine(res, p)


# --------------------------------------------------------------------------------------
# generic MSM (`msmBasic`): src/msm-basic.ts:45-164, used for twisted Edwards (src/parallel.ts:199)
# and for `msmProjective` (src/parallel.ts:69-87)
# --------------------------------------------------------------------------------------


def msm_basic_te(
    scalars: Sequence[int],
    points_xy: Sequence[Tuple[int, int]],
    C: TwistedEdwardsParams = ED_ON_BLS12_377,
    c: Optional[int] = None,
) -> Tuple[int, int]:
    """`msmBasic` on extended twisted-Edwards points (Z = 1, T = xy): signed windows without GLV,
    mixed add / sub into buckets (:103-123), `reduceBucketsChunk` (:180-211), Horner (:142-158).
    b = Scalar.sizeInBits = bit length of the subgroup order."""
    N = len(scalars)
    assert N == len(points_xy)
    if N == 0:
        return (0, 1)
    b = C.q.bit_length()
    if c is None:
        c = window_size_reference(C.p.bit_length(), ceil_log2(N))
    K = -(-(b + 1) // c)
    L = 1 << (c - 1)
    pts = [te_from_affine(xy, C) for xy in points_xy]
    part = []
    for kk in range(K):
        part.append([TE_ZERO] * L)
    for s, P in zip(scalars, pts):
        for kk, (l, neg) in enumerate(signed_digits(s, c, K)):
            if l:
                part[kk][l - 1] = te_add(part[kk][l - 1], te_neg(P, C) if neg else P, C)
    sums = []
    for kk in range(K):
        row = tri = TE_ZERO
        for l in range(L - 1, -1, -1):
            row = te_add(row, part[kk][l], C)
            tri = te_add(tri, row, C)
        sums.append(tri)
    res = sums[K - 1]
    for kk in range(K - 2, -1, -1):
        for _ in range(c):
            res = te_add(res, res, C)
        res = te_add(res, sums[kk], C)
    return te_to_affine(res, C)


def msm_basic_projective(
    scalars: Sequence[int],
    points: Sequence[AffinePoint],
    C: WeierstrassParams = BLS12_377,
    c: Optional[int] = None,
) -> AffinePoint:
    """`msmProjective` (src/parallel.ts:69-87): msmBasic over projective Weierstrass points with
    full-width scalars (Scalar.Simple, b = 253, no GLV) -- BASELINE config 1 (2^14, c = 13)."""
    p = C.p
    N = len(scalars)
    if N == 0:
        return None
    b = C.q.bit_length()
    if c is None:
        c = window_size_reference(p.bit_length(), ceil_log2(N))
    K = -(-(b + 1) // c)
    L = 1 << (c - 1)
    bk = [[PROJ_ZERO] * L for _ in range(K)]
    for s, P in zip(scalars, points):
        for kk, (l, neg) in enumerate(signed_digits(s, c, K)):
            if l:
                Q = aff_neg(P, p) if neg else P
                bk[kk][l - 1] = proj_add(bk[kk][l - 1], proj_from_affine(Q), p)
    sums = [reduce_buckets_column_projective(bk[kk], 1, p) for kk in range(K)]
    res = sums[K - 1]
    for kk in range(K - 2, -1, -1):
        for _ in range(c):
            res = proj_double(res, p)
        res = proj_add(res, sums[kk], p)
    return proj_to_affine(res, p)


# --------------------------------------------------------------------------------------
# wire formats: src/parallel.ts:97-133 (Weierstrass), :215-247 (twisted Edwards),
# packed bytes src/wasm/field-helpers.ts:211-301
# --------------------------------------------------------------------------------------


def points_to_bytes(points: Sequence[Tuple[int, int]], n_bytes: int) -> bytes:
    return b"".join(x.to_bytes(n_bytes, "little") + y.to_bytes(n_bytes, "little") for x, y in points)


def points_from_bytes(buf: bytes, n_bytes: int) -> List[Tuple[int, int]]:
    step = 2 * n_bytes
    assert len(buf) % step == 0
    return [
        (int.from_bytes(buf[i : i + n_bytes], "little"), int.from_bytes(buf[i + n_bytes : i + step], "little"))
        for i in range(0, len(buf), step)
    ]


def scalars_to_bytes(scalars: Sequence[int]) -> bytes:
    return b"".join(s.to_bytes(32, "little") for s in scalars)


def scalars_from_bytes(buf: bytes) -> List[int]:
    assert len(buf) % 32 == 0
    return [int.from_bytes(buf[i : i + 32], "little") for i in range(0, len(buf), 32)]


# --------------------------------------------------------------------------------------
# deterministic synthetic inputs (the reference is unseeded: src/util.ts:201-208)
# --------------------------------------------------------------------------------------


def prng_ints(seed: str, count: int, modulus: int) -> List[int]:
    """Counter-mode SHA-256 stream reduced mod `modulus` (bias < 2^-128: 512-bit draws)."""
    out = []
    for i in range(count):
        h = hashlib.sha256(f"{seed}/{i}/a".encode()).digest() + hashlib.sha256(f"{seed}/{i}/b".encode()).digest()
        out.append(int.from_bytes(h, "little") % modulus)
    return out


def random_points_bls377(seed: str, count: int, C: WeierstrassParams = BLS12_377) -> Tuple[List[Tuple[int, int]], List[int]]:
    """count subgroup points P_i = a_i * G with KNOWN a_i, built additively from a small random
    basis like `randomPointsFast` (src/curve-random.ts:14-92) so generation is O(count) additions."""
    G = (C.gx, C.gy)
    n_basis = 4
    tbl_bits = 6
    basis_k = prng_ints(seed + "/basis", n_basis, C.q)
    tables = []
    for bk in basis_k:
        B = aff_scale(bk, G, C.p)
        row, acc = [], None
        for _ in range(1 << tbl_bits):
            acc = aff_add(acc, B, C.p)
            row.append(acc)
        tables.append(row)
    idx = prng_ints(seed + "/idx", count, 1 << (n_basis * tbl_bits))
    pts, ks = [], []
    for v in idx:
        acc, a = None, 0
        for j in range(n_basis):
            t = (v >> (j * tbl_bits)) & ((1 << tbl_bits) - 1)
            acc = aff_add(acc, tables[j][t], C.p)
            a += (t + 1) * basis_k[j]
        if acc is None:  # vanishing probability; keep the list free of zero points
            acc, a = G, 1
        pts.append(acc)
        ks.append(a % C.q)
    return pts, ks


def random_points_ed377(seed: str, count: int, C: TwistedEdwardsParams = ED_ON_BLS12_377) -> Tuple[List[Tuple[int, int]], List[int]]:
    G = te_from_affine((C.gx, C.gy), C)
    n_basis, tbl_bits = 4, 6
    basis_k = prng_ints(seed + "/basis", n_basis, C.q)
    tables = []
    for bk in basis_k:
        B = te_scale(bk, G, C)
        row, acc = [], TE_ZERO
        for _ in range(1 << tbl_bits):
            acc = te_add(acc, B, C)
            row.append(acc)
        tables.append(row)
    idx = prng_ints(seed + "/idx", count, 1 << (n_basis * tbl_bits))
    pts, ks = [], []
    for v in idx:
        acc, a = TE_ZERO, 0
        for j in range(n_basis):
            t = (v >> (j * tbl_bits)) & ((1 << tbl_bits) - 1)
            acc = te_add(acc, tables[j][t], C)
            a += (t + 1) * basis_k[j]
        pts.append(te_to_affine(acc, C))
        ks.append(a % C.q)
    return pts, ks
