"""Parity of the twisted-Edwards (msmBasic) HIP path against the oracle.  Needs an MI355X: `-m gpu`.
Reference tests mirrored: src/curve-twisted-edwards.test.ts (operators vs bigint), src/msm.test.ts:84-119
(MSM vs bigint msm, affine deep-equal), scripts/zprize23/submission-test.ts (fixed-point KAT)."""
import json
import os

import pytest

from oracle import msm_oracle as O

pytestmark = pytest.mark.gpu

E = O.ED_ON_BLS12_377
P_MOD = E.p
R = 1 << 270
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ed_ctx():
    from montgomery_amd import _lib
    from montgomery_amd.api import MsmContext

    ctx = MsmContext(_lib.CURVE_ED_ON_BLS12_377)
    yield ctx
    ctx.close()


def tb(v):
    return v.to_bytes(32, "little")


def fb(b, i):
    return int.from_bytes(b[32 * i : 32 * i + 32], "little")


def run(ctx, scalars, points, c=None):
    ctx.set_points(O.points_to_bytes(points, 32), check_curve=True)
    res, info = ctx.run(O.scalars_to_bytes(scalars), c=c)
    return (res.x, res.y), info


def test_fp253_operators(ed_ctx):
    from montgomery_amd import _lib

    vals = [0, 1, 2, P_MOD - 1, P_MOD - 2, (P_MOD + 1) // 2, (1 << 30) - 1, 1 << 252] + O.prng_ints("gpu/fp253", 400, P_MOD)
    n = len(vals)
    a = b"".join(tb(v) for v in vals)
    b = b"".join(tb(v) for v in reversed(vals))
    rinv = pow(R, -1, P_MOD)
    out = ed_ctx.test_fp(_lib.OP_MUL, a, b)
    assert all(fb(out, i) == vals[i] * vals[n - 1 - i] * rinv % P_MOD for i in range(n))
    out = ed_ctx.test_fp(_lib.OP_SQR, a)
    assert all(fb(out, i) == vals[i] * vals[i] * rinv % P_MOD for i in range(n))
    out = ed_ctx.test_fp(_lib.OP_ADD, a, b)
    assert all(fb(out, i) == (vals[i] + vals[n - 1 - i]) % P_MOD for i in range(n))
    out = ed_ctx.test_fp(_lib.OP_SUB, a, b)
    assert all(fb(out, i) == (vals[i] - vals[n - 1 - i]) % P_MOD for i in range(n))
    nz = [v for v in vals if v]
    mont = ed_ctx.test_fp(_lib.OP_TO_MONT, b"".join(tb(v) for v in nz))
    inv = ed_ctx.test_fp(_lib.OP_INV, mont)
    back = ed_ctx.test_fp(_lib.OP_FROM_MONT, inv)
    assert all(fb(back, i) == pow(v, -1, P_MOD) for i, v in enumerate(nz))
    for per_lane in (1, 7, 100):   # batchInverse, src/wasm/inverse.ts:220-271
        assert ed_ctx.test_batch_inverse(mont, per_lane) == inv, per_lane
    assert ed_ctx.test_fp(_lib.OP_INV_FERMAT, mont[: 32 * 64]) == inv[: 32 * 64]
    assert ed_ctx.test_fp(_lib.OP_INV_KALISKI, mont[: 32 * 64]) == inv[: 32 * 64]
    assert ed_ctx.test_fp(_lib.OP_INV_WORDSLICED, mont[: 32 * 64]) == inv[: 32 * 64]


def test_unified_addition_operator(ed_ctx):
    """`addOrSubtract` of src/curve-twisted-edwards.ts:84-165 (add-2008-hwcd-3, also used for doubling :215-217) as the
    gather round runs it, against the oracle: random pairs, P + P, P + (-P), P + identity, identity + identity."""
    pts, _ = O.random_points_ed377("gpu/ed/add", 120)
    neg = lambda P: ((E.p - P[0]) % E.p, P[1])
    ident = (0, 1)
    gs = pts[:60] + [pts[0], pts[1], pts[2], ident, ident]
    hs = pts[60:] + [pts[0], neg(pts[1]), ident, pts[3], ident]
    out = ed_ctx.test_batch_add(O.points_to_bytes(gs, 32), O.points_to_bytes(hs, 32))
    for i, (g, h) in enumerate(zip(gs, hs)):
        exp = O.te_to_affine(O.te_add(O.te_from_affine(g, E), O.te_from_affine(h, E), E), E)
        assert (fb(out, 2 * i), fb(out, 2 * i + 1)) == exp, i


def test_golden_vectors(ed_ctx):
    H = lambda x: int(x, 16)
    for c in json.load(open(os.path.join(GOLD, "msm_ed377.json")))["cases"]:
        ed_ctx.set_points(bytes.fromhex(c["points"]))
        for cc in (c["c"], None, 7, 12):
            res, info = ed_ctx.run(bytes.fromhex(c["scalars"]), c=cc)
            assert (res.x, res.y) == (H(c["result"][0]), H(c["result"][1])), (c["name"], cc, info)


def test_zprize_fixed_point(ed_ctx):
    x, y, _ = O.ZPRIZE_ED377_POINT
    assert run(ed_ctx, [2, E.q - 1], [(x, y), (x, y)])[0] == (x, y)


def test_sizes_vs_oracle(ed_ctx):
    """N = 2^0 .. 2^12 like src/msm.test.ts:84-119; the oracle's msmBasic restatement and the known discrete logs agree."""
    pts, ks = O.random_points_ed377("gpu/ed/sizes", 4096)
    G = O.te_from_affine((E.gx, E.gy), E)
    for lg in range(0, 13, 2):
        n = 1 << lg
        sc = O.prng_ints(f"gpu/ed/s{lg}", n, E.q)
        got, info = run(ed_ctx, sc, pts[:n])
        exp = O.te_to_affine(O.te_scale(sum(a * b for a, b in zip(sc, ks[:n])) % E.q, G, E), E)
        assert got == exp, (lg, info)
        if lg <= 8:
            assert O.msm_basic_te(sc, pts[:n], c=info["c"]) == exp


def test_ragged_and_edge_cases(ed_ctx):
    pts, _ = O.random_points_ed377("gpu/ed/edge", 300)
    for n, c in ((1, 4), (3, 6), (17, 9), (100, 14), (300, 7), (255, 16), (129, 5)):
        sc = O.prng_ints(f"gpu/ed/r{n}", n, E.q)
        got, info = run(ed_ctx, sc, pts[:n], c)
        assert got == O.msm_basic_te(sc, pts[:n], c=9), (n, c, info)
    # empty, zero scalars, cancellation -> identity (0, 1)
    ed_ctx.set_points(b"")
    res, _ = ed_ctx.run(b"")
    assert (res.x, res.y) == (0, 1)
    assert run(ed_ctx, [0] * 7, pts[:7])[0] == (0, 1)
    neg = ((-pts[0][0]) % P_MOD, pts[0][1])
    assert run(ed_ctx, [5, 5], [pts[0], neg])[0] == (0, 1)
    assert run(ed_ctx, [5, E.q - 5], [pts[0], pts[0]])[0] == (0, 1)
    # the identity itself as an input point, repeated points, scalar extremes, scalars >= q
    sc = O.prng_ints("gpu/ed/mix", 40, E.q)
    mix = list(pts[:40])
    mix[3] = (0, 1)
    mix[5] = mix[4]
    sc[6] = E.q - 1
    sc[7] = 1
    assert run(ed_ctx, sc, mix, 6)[0] == O.msm_basic_te(sc, mix, c=8)
    Pt = O.te_from_affine(pts[1], E)
    assert run(ed_ctx, [E.q + 9], [pts[1]])[0] == O.te_to_affine(O.te_scale(9, Pt, E), E)


def test_scalars_beyond_q_are_reduced(ed_ctx):
    """Any 256-bit scalar is accepted and reduced (2^256 < 56 q for this curve)."""
    base, ks = O.random_points_ed377("gpu/ed/bigscalar", 4)
    G = O.te_from_affine((E.gx, E.gy), E)
    sc = [(1 << 256) - 1, 41 * E.q + 5, 54 * E.q + 123, E.q]
    got, _ = run(ed_ctx, sc, base, 7)
    tot = sum(s * k for s, k in zip(sc, ks)) % E.q
    assert got == O.te_to_affine(O.te_scale(tot, G, E), E)


def test_error_codes(ed_ctx):
    from montgomery_amd import MsmError

    with pytest.raises(MsmError) as e:
        ed_ctx.set_points(tb(P_MOD) + tb(1))
    assert e.value.code == 3
    with pytest.raises(MsmError) as e:
        ed_ctx.set_points(tb(5) + tb(7), check_curve=True)
    assert e.value.code == 3


def test_large_2p20_known_discrete_logs(ed_ctx):
    """BASELINE configs[3]: 2^20 Ed-on-BLS12-377 MSM.  Points are 1024 known multiples of G tiled 1024 times, so
    sum s_i P_i = (sum s_i a_(i mod 1024)) G is checkable in O(N) on the host."""
    base, ks = O.random_points_ed377("gpu/ed/big", 1024)
    n = 1 << 20
    ed_ctx.set_points(O.points_to_bytes(base, 32) * 1024)
    import random

    rnd = random.Random(20)
    sc = [rnd.randrange(E.q) for _ in range(n)]
    res, info = ed_ctx.run(O.scalars_to_bytes(sc))
    G = O.te_from_affine((E.gx, E.gy), E)
    tot = sum(s * ks[i & 1023] for i, s in enumerate(sc)) % E.q
    assert (res.x, res.y) == O.te_to_affine(O.te_scale(tot, G, E), E), info
    res2, _ = ed_ctx.run(O.scalars_to_bytes(sc), c=12)
    assert (res2.x, res2.y) == (res.x, res.y)


@pytest.mark.parametrize("lg", [12, 20])
def test_generated_points_known_discrete_logs(ed_ctx, lg):
    """Device-generated inputs (randomPointsFast / randomScalars, src/curve-random.ts): P_i = a_i G with the a_i known to
    the host, 2^20 DISTINCT points (what bench.py --curve ed377 runs on): sum s_i P_i = (sum s_i a_i) G."""
    n = 1 << lg
    a = O.scalars_from_bytes(ed_ctx.generate_points(n, seed=500 + lg, want_scalars=True))
    G = O.te_from_affine((E.gx, E.gy), E)
    for i in (0, 1, n // 2, n - 1):
        assert ed_ctx.get_point(i) == O.te_to_affine(O.te_scale(a[i], G, E), E)
    assert len({ed_ctx.get_point(i) for i in range(0, n, max(1, n // 64))}) == min(n, 64)
    dev, sb = ed_ctx.generate_scalars(n, seed=600 + lg, to_host=True)
    s = O.scalars_from_bytes(sb)
    assert all(v < E.q for v in s) and max(s).bit_length() == E.q.bit_length()
    res, info = ed_ctx.run_device(dev, n)
    tot = sum(x * y for x, y in zip(a, s)) % E.q
    assert (res.x, res.y) == O.te_to_affine(O.te_scale(tot, G, E), E), info
    res2, _ = ed_ctx.run_device(dev, n, c=9)
    assert (res2.x, res2.y) == (res.x, res.y)


def test_window_shards_combine(ed_ctx):
    """Window shards of the Edwards MSM (msm_window_sums) recombine to the full result (msm_combine)."""
    pts, _ = O.random_points_ed377("gpu/ed/shard", 200)
    sc = O.prng_ints("gpu/ed/shard/s", 200, E.q)
    full, info = run(ed_ctx, sc, pts, 9)
    K = info["K"]
    sb = O.scalars_to_bytes(sc)
    parts = b"".join(ed_ctx.window_sums(sb, 200, lo, min(K, lo + 5), c=9)[0] for lo in range(0, K, 5))
    res = ed_ctx.combine(parts, K, 9)
    assert (res.x, res.y) == full == O.msm_basic_te(sc, pts, c=9)


def test_reference_shaped_api():
    from montgomery_amd.api import ED_ON_BLS12_377_PARAMS, TwistedEdwards, compute_msm_ed

    cv = TwistedEdwards.create(ED_ON_BLS12_377_PARAMS)
    x, y, t = O.ZPRIZE_ED377_POINT
    r = compute_msm_ed([{"x": x, "y": y, "z": 1, "t": t}] * 2, [2, E.q - 1], curve=cv)   # scripts/zprize23/submission-test.ts
    assert (r["x"], r["y"]) == (x, y)
    pts, _ = O.random_points_ed377("gpu/ed/api", 30)
    sc = O.prng_ints("gpu/ed/api/s", 30, E.q)
    par = cv.Parallel
    pp, sp = par.getPointer(30 * 64), par.getScalarPointer(30 * 32)
    par.pointsFromBytes(pp, O.points_to_bytes(pts, 32), 30)
    par.scalarsFromBytes(sp, O.scalars_to_bytes(sc), 30)
    out = par.msm(sp, pp, 30, True, {"c": 6})
    assert (out["result"].x, out["result"].y) == O.msm_basic_te(sc, pts, c=6)
    # the reference's way from `result` to affine bigints on this curve (scripts/zprize23/submission.ts:33-34)
    big = cv.Curve.toBigint(out["result"])
    assert big["Z"] == 1 and big["T"] == big["X"] * big["Y"] % E.p
    aff = cv.Bigint.toAffine({k: v * 5 % E.p for k, v in big.items()})   # any projective representative
    assert (aff["x"], aff["y"]) == O.msm_basic_te(sc, pts, c=6)
    cv.context.close()
