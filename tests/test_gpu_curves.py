"""BLS12-381 G1 (curve id 2) and Pallas (curve id 3) through the same HIP path: parity against the CPU
oracle.  `-m gpu`.

The reference runs its MSM test over pallas, bls12-377 and bls12-381 with one code path
(src/msm.test.ts:27-31); here the Weierstrass kernels are templates over the curve constants, and this
file repeats the BLS12-377 parity ladder for the 381-bit field (p != 1 mod 2^30: general Montgomery
factor, p^-1 != 1 in the divsteps inverse) and for Pallas (255-bit p carried in the same 13-limb path),
each with its GLV lattice (127-bit halves, K = 8 at c = 16).
"""
import json
import os

import pytest

from oracle import msm_oracle as O

pytestmark = pytest.mark.gpu

R = 1 << 390   # Montgomery radix and coordinate bytes of the case under test: set by the `cv` fixture (module-scoped, so the
CB = 48         # tests of one curve run together): 2^390 / 48 for BLS12-381, 2^270 / 32 for Pallas (limbs sized per field)
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def H(x):
    return int(x, 16)


def tb(v):
    return v.to_bytes(CB, "little")


def fb(b, i):
    return int.from_bytes(b[CB * i : CB * i + CB], "little")


def enc_pt(P):
    return b"\0" * (2 * CB) if P is None else tb(P[0]) + tb(P[1])


def golden_points(hexstr):
    """The golden files store 48-byte coordinates; a 32-byte curve takes the low 32 bytes of each."""
    raw = bytes.fromhex(hexstr)
    if CB == 48:
        return raw
    return b"".join(raw[48 * i : 48 * i + CB] for i in range(len(raw) // 48))


class Case:
    """One extra curve: oracle parameters, a live context and its golden file."""

    def __init__(self, name):
        from montgomery_amd import _lib
        from montgomery_amd.api import MsmContext

        self.name = name
        self.B = O.BLS12_381 if name == "bls381" else O.PALLAS
        self.ctx = MsmContext(_lib.CURVE_BLS12_381_G1 if name == "bls381" else _lib.CURVE_PALLAS)
        with open(os.path.join(GOLD, f"{name}.json")) as f:
            self.gold = json.load(f)


@pytest.fixture(scope="module", params=["bls381", "pallas"])
def cv(request):
    global R, CB
    case = Case(request.param)
    CB = case.ctx.coord_bytes
    R = 1 << (390 if CB == 48 else 270)
    assert CB == (48 if request.param == "bls381" else 32)
    yield case
    case.ctx.close()


def run_msm(ctx, scalars, points, c=None):
    ctx.set_points(b"".join(enc_pt(P) for P in points), check_curve=True)
    res, info = ctx.run(O.scalars_to_bytes(scalars), c=c)
    return res.as_tuple(), info


def test_fp_operators_and_inverse(cv):
    ctx, gold, B, P_MOD = cv.ctx, cv.gold, cv.B, cv.B.p
    from montgomery_amd import _lib

    p = P_MOD
    vals = [0, 1, 2, p - 1, p - 2, (p + 1) // 2, 1 << (p.bit_length() - 1), (1 << 30) - 1, 1 << 30, (1 << 250) + 5] + O.prng_ints(f"gpu/fp/{cv.name}", 500, p)
    n = len(vals)
    a = b"".join(tb(v) for v in vals)
    b = b"".join(tb(v) for v in reversed(vals))
    rinv = pow(R, -1, p)
    out = ctx.test_fp(_lib.OP_MUL, a, b)
    assert all(fb(out, i) == vals[i] * vals[n - 1 - i] * rinv % p for i in range(n))
    out = ctx.test_fp(_lib.OP_SQR, a)
    assert all(fb(out, i) == vals[i] * vals[i] * rinv % p for i in range(n))
    out = ctx.test_fp(_lib.OP_ADD, a, b)
    assert all(fb(out, i) == (vals[i] + vals[n - 1 - i]) % p for i in range(n))
    out = ctx.test_fp(_lib.OP_SUB, a, b)
    assert all(fb(out, i) == (vals[i] - vals[n - 1 - i]) % p for i in range(n))
    nz = [v for v in vals if v]
    an = b"".join(tb(v) for v in nz)
    mont = ctx.test_fp(_lib.OP_TO_MONT, an)
    assert all(fb(mont, i) == v * R % p for i, v in enumerate(nz))
    assert ctx.test_fp(_lib.OP_FROM_MONT, mont) == an
    back = ctx.test_fp(_lib.OP_FROM_MONT, ctx.test_fp(_lib.OP_INV, mont))
    assert all(fb(back, i) == pow(v, -1, p) for i, v in enumerate(nz))
    inv = ctx.test_fp(_lib.OP_INV, mont)
    assert ctx.test_fp(_lib.OP_INV_FERMAT, mont[: CB * 64]) == inv[: CB * 64]
    assert ctx.test_fp(_lib.OP_INV_KALISKI, mont[: CB * 64]) == inv[: CB * 64]
    assert ctx.test_fp(_lib.OP_INV_WORDSLICED, mont[: CB * 64]) == inv[: CB * 64]
    for per_lane in (1, 7, 100):
        out = ctx.test_fp(_lib.OP_FROM_MONT, ctx.test_batch_inverse(mont[: CB * 203], per_lane))
        assert all(fb(out, i) == pow(v, -1, p) for i, v in enumerate(nz[:203])), per_lane
    # golden field vectors
    cases = gold["fp"]
    ga = b"".join(tb(H(c["a"])) for c in cases)
    gb = b"".join(tb(H(c["b"])) for c in cases)
    am, bm = ctx.test_fp(_lib.OP_TO_MONT, ga), ctx.test_fp(_lib.OP_TO_MONT, gb)
    for op, key in ((_lib.OP_MUL, "mul"), (_lib.OP_ADD, "add"), (_lib.OP_SUB, "sub"), (_lib.OP_SQR, "sqr")):
        out = ctx.test_fp(_lib.OP_FROM_MONT, ctx.test_fp(op, am, bm))
        assert [fb(out, i) for i in range(len(cases))] == [H(c[key]) for c in cases], key


def test_glv_decompose(cv):
    ctx, gold, B, P_MOD = cv.ctx, cv.gold, cv.B, cv.B.p
    g = O.glv_params(B.q, B.lam)
    scalars = [H(c["s"]) for c in gold["glv"]] + O.prng_ints(f"gpu/glv/{cv.name}", 5000, B.q)
    got = ctx.test_glv(O.scalars_to_bytes(scalars))
    for s, r in zip(scalars, got):
        assert tuple(r) == O.glv_decompose(s, g), hex(s)


def test_batch_add(cv):
    ctx, gold, B, P_MOD = cv.ctx, cv.gold, cv.B, cv.B.p
    pts, _ = O.random_points_bls377(f"gpu/batch/{cv.name}", 400, B)
    G = (B.gx, B.gy)
    gs = pts[:200] + [pts[0], pts[1], None, pts[3], None, G]
    hs = pts[200:] + [pts[0], O.aff_neg(pts[1], P_MOD), pts[2], None, None, G]
    exp = [O.aff_add(a, b, P_MOD) for a, b in zip(gs, hs)]
    out = ctx.test_batch_add(b"".join(map(enc_pt, gs)), b"".join(map(enc_pt, hs)))
    for i, e in enumerate(exp):
        assert (fb(out, 2 * i), fb(out, 2 * i + 1)) == ((0, 0) if e is None else e), i


def test_msm_golden_vectors(cv):
    ctx, gold, B, P_MOD = cv.ctx, cv.gold, cv.B, cv.B.p
    for c in gold["msm"]:
        ctx.set_points(golden_points(c["points"]))
        exp = None if c["result"] is None else (H(c["result"][0]), H(c["result"][1]))
        for cc in (c["c"], None, 3, 11, 16):
            res, info = ctx.run(bytes.fromhex(c["scalars"]), c=cc)
            assert res.as_tuple() == exp, (c["name"], cc, info)


def test_msm_sizes_like_reference_msm_test(cv):
    ctx, gold, B, P_MOD = cv.ctx, cv.gold, cv.B, cv.B.p
    """N = 2^0, 2^2, ..., 2^12 (src/msm.test.ts:33-42): GPU == known-discrete-log answer; == oracle MSM up to 2^8."""
    pts, ks = O.random_points_bls377(f"gpu/sizes/{cv.name}", 4096, B)
    G = (B.gx, B.gy)
    for lg in range(0, 13, 2):
        n = 1 << lg
        sc = O.prng_ints(f"gpu/sizes/{cv.name}/{lg}", n, B.q)
        got, info = run_msm(ctx, sc, pts[:n])
        assert got == O.aff_scale(sum(a * b for a, b in zip(sc, ks[:n])) % B.q, G, P_MOD), (lg, info)
        if lg <= 8:
            assert got == O.msm_batched_affine(sc, pts[:n], B)


def test_msm_ragged_windows_and_edges(cv):
    ctx, gold, B, P_MOD = cv.ctx, cv.gold, cv.B, cv.B.p
    pts, ks = O.random_points_bls377(f"gpu/ragged/{cv.name}", 777, B)
    G = (B.gx, B.gy)
    q = B.q
    for n, c in ((1, 2), (2, 16), (3, 5), (17, 9), (100, 13), (255, 6), (777, 8), (777, 10), (513, 12)):
        sc = O.prng_ints(f"gpu/ragged/{cv.name}/{n}/{c}", n, q)
        got, info = run_msm(ctx, sc, pts[:n], c)
        assert got == O.aff_scale(sum(a * b for a, b in zip(sc, ks[:n])) % q, G, P_MOD), (n, c, info)
        assert info["c"] == c and info["K"] == -(-128 // c)
    ctx.set_points(b"")
    assert ctx.run(b"")[0].isZero
    assert run_msm(ctx, [0] * 10, pts[:10])[0] is None
    assert run_msm(ctx, [5, q - 5], [pts[0], pts[0]])[0] is None
    assert run_msm(ctx, [9, 9], [pts[0], O.aff_neg(pts[0], P_MOD)])[0] is None
    for s in (1, q - 1, q - 2, B.lam, B.lam + 1, (1 << 254) + 12345, (1 << 127) - 1, 1 << 127, 1 << 128):
        assert run_msm(ctx, [s], [pts[1]])[0] == O.aff_scale(s, pts[1], P_MOD), hex(s)
    assert run_msm(ctx, [q + 5], [pts[2]])[0] == O.aff_scale(5, pts[2], P_MOD)   # scalars >= q are reduced
    assert run_msm(ctx, [(1 << 256) - 1], [pts[2]])[0] == O.aff_scale(((1 << 256) - 1) % q, pts[2], P_MOD)
    sc = O.prng_ints(f"gpu/edge/{cv.name}/mix", 48, q)
    mix = list(pts[:48])
    mix[3] = None
    mix[5] = mix[4]; sc[5] = sc[4]
    mix[7] = O.aff_neg(mix[6], P_MOD); sc[7] = sc[6]
    for c in (4, 7, None):
        assert run_msm(ctx, sc, mix, c)[0] == O.msm_batched_affine(sc, mix, B, c=6), c


def test_msm_error_codes(cv):
    ctx, gold, B, P_MOD = cv.ctx, cv.gold, cv.B, cv.B.p
    from montgomery_amd import MsmError

    with pytest.raises(MsmError) as e:
        ctx.set_points(tb(P_MOD) + tb(1))               # coordinate >= p
    assert e.value.code == 3
    with pytest.raises(MsmError) as e:
        ctx.set_points(tb(5) + tb(7), check_curve=True)  # not on y^2 = x^3 + b
    assert e.value.code == 3


def test_window_shards(cv):
    ctx, gold, B, P_MOD = cv.ctx, cv.gold, cv.B, cv.B.p
    n = 300
    pts, _ = O.random_points_bls377(f"gpu/shard/{cv.name}", n, B)
    sc = O.prng_ints(f"gpu/shard/{cv.name}/s", n, B.q)
    full, info = run_msm(ctx, sc, pts, 8)
    K = info["K"]
    sb = O.scalars_to_bytes(sc)
    parts = b"".join(ctx.window_sums(sb, n, lo, min(K, lo + 3), c=8)[0] for lo in range(0, K, 3))
    assert ctx.combine(parts, K, 8).as_tuple() == full == O.msm_batched_affine(sc, pts, B)


@pytest.mark.parametrize("lg", [16, 20])
def test_msm_large_known_discrete_logs(cv, lg):
    ctx, gold, B, P_MOD = cv.ctx, cv.gold, cv.B, cv.B.p
    """sum s_i P_i = (sum s_i a_i) G with P_i = a_i G generated on the GPU."""
    n = 1 << lg
    a = O.scalars_from_bytes(ctx.generate_points(n, seed=300 + lg, want_scalars=True))
    G = (B.gx, B.gy)
    for i in (0, 1, n - 1):
        assert ctx.get_point(i) == O.aff_scale(a[i], G, P_MOD)
    dev, sb = ctx.generate_scalars(n, seed=400 + lg, to_host=True)
    s = O.scalars_from_bytes(sb)
    assert all(v < B.q for v in s) and max(s).bit_length() >= B.q.bit_length() - 1
    res, info = ctx.run_device(dev, n)
    assert res.as_tuple() == O.aff_scale(sum(x * y for x, y in zip(a, s)) % B.q, G, P_MOD), info
    res2, _ = ctx.run_device(dev, n, c=11)
    assert res2.as_tuple() == res.as_tuple()


@pytest.mark.parametrize("c", [20, 22])
def test_big_windows_chunk_ordered_round1(cv, c):
    """2^23 points with windows of 2^19 / 2^21 buckets: the three-pass split, round 1 walking its pairs chunk by chunk of
    the row table and leaving element records (one 64-byte record [x | y] per element on the 8-word field of Pallas, an x and
    a y record on the 12-word field of BLS12-381), round 2 reading them back -- against the known discrete logs."""
    from oracle import c_oracle

    ctx, B = cv.ctx, cv.B
    n = 1 << 23
    a = ctx.generate_points(n, seed=777 + c, want_scalars=True, raw=True)
    dev, s = ctx.generate_scalars(n, seed=888 + c, to_host=True, raw=True)
    res, info = ctx.run_device(dev, n, c=c)
    assert info["c"] == c
    k = c_oracle.dot_mod(a, s, n, B.q)
    assert res.as_tuple() == O.aff_scale(k, (B.gx, B.gy), B.p), info
    res16, _ = ctx.run_device(dev, n)
    assert res16.as_tuple() == res.as_tuple()


def test_reference_shaped_api(cv):
    ctx, gold, B, P_MOD = cv.ctx, cv.gold, cv.B, cv.B.p
    from montgomery_amd.api import BLS12_381_PARAMS, PALLAS_PARAMS, Weierstrass

    mod = Weierstrass.create(BLS12_381_PARAMS if cv.name == "bls381" else PALLAS_PARAMS)
    pts, _ = O.random_points_bls377(f"gpu/api/{cv.name}", 50, B)
    sc = O.prng_ints(f"gpu/api/{cv.name}/s", 50, B.q)
    par = mod.Parallel
    wb = B.n_bytes    # the reference's wire size: 48 (BLS12-381) or 32 (Pallas) bytes per coordinate
    pp, sp = par.getPointer(50 * 2 * wb), par.getScalarPointer(50 * 32)
    par.pointsFromBytes(pp, O.points_to_bytes(pts, wb), 50)
    par.scalarsFromBytes(sp, O.scalars_to_bytes(sc), 50)
    out = par.msmUnsafe(sp, pp, 50, True, {"c": 6})
    assert out["result"].as_tuple() == O.msm_batched_affine(sc, pts, B)
    assert out["log"][0][0]["K"] == 22
    proj = par.msmProjective(sp, pp, 50, {"c": 8})
    assert proj["result"].as_tuple() == out["result"].as_tuple()
    assert proj["info"]["K"] == -(-(B.q.bit_length() + 1) // 8)
    mod.context.close()
