"""Pins the CPU oracle (oracle/msm_oracle.py and its C port) against the reference's known-answer
material and against the committed golden vectors.  No GPU needed."""
import json
import os

import pytest

from oracle import msm_oracle as O

C = O.BLS12_377
E = O.ED_ON_BLS12_377
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def H(x):
    return int(x, 16)


# ---- reference known-answer material -------------------------------------------------------


def test_field_identities():
    """Hard-coded identities of src/bigint/field.test.ts:81-104 on both base fields."""
    for p in (C.p, E.p):
        assert (3 - 8) % p == p - 5
        assert (p - 1 + 2) % p == 1
        assert (p - 1) * 2 % p == p - 2
        assert (p - 3) * (p - 3) % p == 9
        assert O.inv_mod(1, p) == 1
        assert O.inv_mod(2, p) == (p + 1) // 2
        assert O.inv_mod(p - 2, p) == (p - 1) // 2
        assert O.inv_mod(3, p) == ((2 * p + 1) // 3 if p % 3 == 1 else (p + 1) // 3)
        assert pow(p - 10, 2, p) == 100
        assert pow(2, p - 1, p) == 1 and pow(2, p - 1 + 3, p) == 8
        r = O.sqrt_mod(p - 1, p)
        assert r is not None and r * r % p == p - 1
    with pytest.raises(ZeroDivisionError):
        O.inv_mod(0, C.p)


def test_generators_and_endomorphism():
    """src/concrete/bls12-377.params.ts:25-34,49-63; ed-on-bls12-377.params.ts:18-22; src/bigint/curves.test.ts."""
    G = (C.gx, C.gy)
    assert O.aff_is_on_curve(G, C)
    assert O.aff_scale(C.q, G, C.p) is None
    assert O.aff_scale(C.q - 1, G, C.p) == O.aff_neg(G, C.p)
    assert pow(C.lam, 3, C.q) == 1 and pow(C.beta, 3, C.p) == 1
    assert C.lam == pow(11, (C.q - 1) // 3, C.q)
    assert O.aff_scale(C.lam, G, C.p) == (C.beta * C.gx % C.p, C.gy)
    GE = O.te_from_affine((E.gx, E.gy), E)
    assert O.te_is_on_curve(GE, E)
    assert O.te_to_affine(O.te_scale(E.q, GE, E), E) == (0, 1)
    # base field of the Edwards curve is the scalar field of BLS12-377
    assert E.p == C.q


def test_zprize_fixed_points():
    """scripts/zprize23/submission-test-bls377.ts:6-45 and submission-test.ts:5-21."""
    P = O.ZPRIZE_BLS377_POINT
    assert O.aff_is_on_curve(P, C) and O.aff_scale(C.q, P, C.p) is None
    assert O.msm_batched_affine([2, C.q - 1], [P, P]) == P
    sc = O.prng_ints("kat/same", 50, C.q)
    assert O.msm_batched_affine(sc, [P] * 50) == O.msm_batched_affine([sum(sc) % C.q], [P])
    x, y, t = O.ZPRIZE_ED377_POINT
    assert x * y % E.p == t and O.te_is_on_curve((x, y, 1, t), E)
    assert O.msm_basic_te([2, E.q - 1], [(x, y), (x, y)]) == (x, y)


def test_glv_constants_and_bounds():
    """Constants derived like glvGeneral (src/wasm/glv.ts:35-63, :216-228); SURVEY.md section 8 table."""
    g = O.glv_params(C.q, C.lam)
    assert (g.n, g.n0, g.m, g.k, g.max_bits) == (9, 5, 145, 116, 126)
    assert g.m0 == -438 and g.v00 == 1 and g.v11 == 1
    assert g.v01 == 0x452217CC900000010A11800000000001 and g.v10 == -0x452217CC900000010A11800000000000
    assert g.v00 * g.v11 - g.v10 * g.v01 == C.q
    for s in O.prng_ints("kat/glv", 3000, C.q) + [0, 1, C.q - 1]:
        a0, a1, n0, n1 = O.glv_decompose(s, g)
        s0, s1 = (-a0 if n0 else a0), (-a1 if n1 else a1)
        assert (s0 + s1 * C.lam - s) % C.q == 0     # src/glv/glv-test.ts:106-110
        assert a0 < (1 << g.max_bits) and a1 < (1 << g.max_bits)


def test_msm_algebraic_identities():
    """src/bigint/msm.test.ts:18-101 style identities between the three MSM restatements."""
    pts, ks = O.random_points_bls377("kat/pts", 40)
    sc = O.prng_ints("kat/sc", 40, C.q)
    spec = O.msm_spec_affine(sc, pts, C)
    assert spec == O.msm_naive_affine(sc, pts, C)
    for c, safe, chunks in ((None, True, 1), (4, False, 3), (7, True, 2)):
        assert O.msm_batched_affine(sc, pts, c=c, safe=safe, n_chunks=chunks) == spec
    assert O.msm_basic_projective(sc, pts, c=6) == spec
    assert O.msm_batched_affine(sc + [C.q - s for s in sc], pts + pts, c=5) is None
    assert O.msm_batched_affine([7] * 40, pts, c=4) == O.aff_scale(7, O.msm_batched_affine([1] * 40, pts, c=4), C.p)
    G = (C.gx, C.gy)
    assert spec == O.aff_scale(sum(a * b for a, b in zip(sc, ks)) % C.q, G, C.p)


def test_window_table_and_digits():
    """src/msm-common.ts:8-41 and the signed recoding of src/msm-batched-affine.ts:183-193."""
    assert [O.window_size_reference(377, n) for n in (14, 16, 20, 26)] == [13, 14, 18, 25]
    assert O.window_size_reference(253, 16) == 12 and O.window_size_reference(253, 20) == 19
    for c in (4, 13, 16):
        K = -(-127 // c)
        L = 1 << (c - 1)
        for s in O.prng_ints(f"kat/dig{c}", 200, 1 << 126) + [0, (1 << 126) - 1]:
            d = O.signed_digits(s, c, K)
            assert all(0 <= l <= L for l, _ in d)
            assert sum((-l if neg else l) << (c * k) for k, (l, neg) in enumerate(d)) == s


def test_twisted_edwards_msm_basic():
    pts, ks = O.random_points_ed377("kat/ed", 20)
    sc = O.prng_ints("kat/edsc", 20, E.q)
    GE = O.te_from_affine((E.gx, E.gy), E)
    exp = O.te_to_affine(O.te_scale(sum(a * b for a, b in zip(sc, ks)) % E.q, GE, E), E)
    assert O.msm_basic_te(sc, pts, c=5) == exp
    assert O.msm_basic_te(sc, pts) == exp


# ---- golden vectors --------------------------------------------------------------------------


def test_golden_fp_and_glv():
    d = load("fp377.json")
    p = H(d["modulus"])
    assert p == C.p
    for c in d["cases"]:
        a, b = H(c["a"]), H(c["b"])
        assert a * b % p == H(c["mul"]) and (a + b) % p == H(c["add"]) and (a - b) % p == H(c["sub"]) and a * a % p == H(c["sqr"])
        if c["inv"]:
            assert H(c["inv"]) * a % p == 1
    d = load("glv377.json")
    g = O.glv_params(C.q, C.lam)
    for c in d["cases"]:
        assert O.glv_decompose(H(c["s"]), g) == (H(c["s0"]), H(c["s1"]), c["neg0"], c["neg1"])


def test_golden_point_add():
    for c in load("point_add377.json")["cases"]:
        dec = lambda P: None if P is None else (H(P[0]), H(P[1]))
        assert O.aff_add(dec(c["g"]), dec(c["h"]), C.p) == dec(c["sum"])
        # the batched form agrees with the one-by-one form
        assert O.batch_add_affine([dec(c["g"])], [dec(c["h"])], C.p)[0] == dec(c["sum"])


def test_golden_msm_python_oracle():
    for c in load("msm377.json")["cases"]:
        if c["n"] > 128:
            continue
        sc = O.scalars_from_bytes(bytes.fromhex(c["scalars"]))
        pts = [None if P == (0, 0) else P for P in O.points_from_bytes(bytes.fromhex(c["points"]), 48)]
        exp = None if c["result"] is None else (H(c["result"][0]), H(c["result"][1]))
        assert O.msm_batched_affine(sc, pts, c=c["c"]) == exp, c["name"]
    for c in load("msm_ed377.json")["cases"]:
        sc = O.scalars_from_bytes(bytes.fromhex(c["scalars"]))
        pts = O.points_from_bytes(bytes.fromhex(c["points"]), 32)
        assert O.msm_basic_te(sc, pts, c=c["c"]) == (H(c["result"][0]), H(c["result"][1])), c["name"]


# ---- C port of the oracle ------------------------------------------------------------------------


def test_c_oracle_field_and_glv(c_oracle):
    g = O.glv_params(C.q, C.lam)
    for s in O.prng_ints("kat/cglv", 2000, C.q) + [0, 1, C.q - 1, C.lam]:
        assert c_oracle.glv_decompose(s) == O.glv_decompose(s, g)
    vals = O.prng_ints("kat/cfp", 50, C.p) + [0, 1, C.p - 1]
    for i, a in enumerate(vals):
        b = vals[-1 - i]
        assert c_oracle.fp_op(0, a, b) == a * b % C.p
        assert c_oracle.fp_op(2, a, b) == (a + b) % C.p
        assert c_oracle.fp_op(3, a, b) == (a - b) % C.p
        if a:
            assert c_oracle.fp_op(1, a) == pow(a, -1, C.p)


def test_c_oracle_golden_msm(c_oracle):
    for c in load("msm377.json")["cases"]:
        exp = None if c["result"] is None else (H(c["result"][0]), H(c["result"][1]))
        for cc in (c["c"] or 0, 0):
            got, _ = c_oracle.msm_bls377(bytes.fromhex(c["points"]), bytes.fromhex(c["scalars"]), cc)
            assert got == exp, (c["name"], cc)


def test_c_oracle_n4096_known_logs(c_oracle):
    d = load("msm377_4096.json")
    pts, _ = O.random_points_bls377(d["seed_points"], d["n"])
    sc = O.prng_ints(d["seed_scalars"], d["n"], C.q)
    got, _ = c_oracle.msm_bls377(O.points_to_bytes(pts, 48), O.scalars_to_bytes(sc), 0)
    assert got == (H(d["result"][0]), H(d["result"][1]))


def test_config1_msm_basic_2p14_cpu_plumbing(c_oracle):
    """BASELINE configs[0]: 2^14 BLS12-377 G1 MSM via the msm-basic algorithm (`msmProjective`,
    src/parallel.ts:69-87: full 253-bit scalars, no GLV, c = 13, K = 20) on the CPU, single thread --
    the oracle's restatement against the known discrete logs and against the batched-affine C port."""
    base, ks = O.random_points_bls377("cfg1", 256)
    n = 1 << 14
    sc = O.prng_ints("cfg1/s", n, C.q)
    pts = [base[i & 255] for i in range(n)]
    assert O.window_size_reference(377, 14) == 13
    got = O.msm_basic_projective(sc, pts)
    G = (C.gx, C.gy)
    assert got == O.aff_scale(sum(s * ks[i & 255] for i, s in enumerate(sc)) % C.q, G, C.p)
    ref, _ = c_oracle.msm_bls377(O.points_to_bytes(pts, 48), O.scalars_to_bytes(sc), 0)
    assert ref == got


# ---- BLS12-381 G1 and Pallas (src/concrete/bls12-381.params.ts, pasta.params.ts; src/msm.test.ts:29-31) ----

EXTRA_CURVES = [("bls381", "bls381.json"), ("pallas", "pallas.json")]


def extra_curve(name):
    return O.BLS12_381 if name == "bls381" else O.PALLAS


@pytest.mark.parametrize("name,gold", EXTRA_CURVES)
def test_extra_curve_params_glv_and_msm_identities(name, gold):
    B = extra_curve(name)
    G = (B.gx, B.gy)
    assert O.aff_is_on_curve(G, B) and O.aff_scale(B.q, G, B.p) is None
    # (beta x, y) = lambda (x, y): bls12-381.params.ts:11-31 (the lambda2 / beta2 pair), pasta.params.ts:24-36
    assert pow(B.lam, 3, B.q) == 1 and pow(B.beta, 3, B.p) == 1
    assert O.aff_scale(B.lam, G, B.p) == (B.beta * B.gx % B.p, B.gy)
    g = O.glv_params(B.q, B.lam)
    assert (g.n, g.n0, g.m, g.k, g.max_bits) == (9, 5, 145, 116, 127)
    assert g.v00 * g.v11 - g.v10 * g.v01 == B.q
    for s in O.prng_ints(f"kat/glv{name}", 2000, B.q) + [0, 1, B.q - 1]:
        a0, a1, n0, n1 = O.glv_decompose(s, g)
        s0, s1 = (-a0 if n0 else a0), (-a1 if n1 else a1)
        assert (s0 + s1 * B.lam - s) % B.q == 0 and max(a0, a1) < (1 << g.max_bits)
    pts, ks = O.random_points_bls377(f"kat/{name}", 40, B)
    sc = O.prng_ints(f"kat/{name}/s", 40, B.q)
    spec = O.msm_naive_affine(sc, pts, B)
    assert spec == O.aff_scale(sum(a * b for a, b in zip(sc, ks)) % B.q, G, B.p)
    for c, safe, chunks in ((None, True, 1), (4, False, 3), (7, True, 2)):
        assert O.msm_batched_affine(sc, pts, B, c=c, safe=safe, n_chunks=chunks) == spec
    assert O.msm_basic_projective(sc, pts, B, c=6) == spec


@pytest.mark.parametrize("name,gold", EXTRA_CURVES)
def test_golden_extra_curves(name, gold):
    B = extra_curve(name)
    d = load(gold)
    p = H(d["modulus"])
    assert p == B.p and H(d["q"]) == B.q and d["max_bits"] == 127
    for c in d["fp"]:
        a, b = H(c["a"]), H(c["b"])
        assert a * b % p == H(c["mul"]) and (a + b) % p == H(c["add"]) and (a - b) % p == H(c["sub"]) and a * a % p == H(c["sqr"])
        if c["inv"]:
            assert H(c["inv"]) * a % p == 1
    g = O.glv_params(B.q, B.lam)
    assert [hex(g.v00), hex(g.v01), str(g.v10), hex(g.v11)] == d["v"] and str(g.m0) == d["m0"] and str(g.m1) == d["m1"]
    for c in d["glv"]:
        assert O.glv_decompose(H(c["s"]), g) == (H(c["s0"]), H(c["s1"]), c["neg0"], c["neg1"])
    for c in d["msm"]:
        if c["n"] > 128:
            continue
        sc = O.scalars_from_bytes(bytes.fromhex(c["scalars"]))
        pts = [None if P == (0, 0) else P for P in O.points_from_bytes(bytes.fromhex(c["points"]), 48)]
        exp = None if c["result"] is None else (H(c["result"][0]), H(c["result"][1]))
        assert O.msm_batched_affine(sc, pts, B, c=c["c"]) == exp, c["name"]


def test_wordsliced_almost_inverse_restatement():
    """src/inverse/faster-inverse.ts:55-60 asserts, on random field elements: k + 1 >= b and k <= 2 n w, s < p,
    x s = 2^k (mod p).  Same assertions on the restatement, for the reference's own configuration (Pallas, w = 32) and
    for the limb sizes the GPU variant uses (w = 30: 13 limbs for the 377-bit prime, 9 for the 253-bit one)."""
    for p, w in ((O.PALLAS.p, 32), (O.BLS12_377.p, 30), (O.ED_ON_BLS12_377.p, 30), (O.BLS12_381.p, 30)):
        b = p.bit_length()
        n = -(-b // w)
        for x in O.prng_ints(f"kat/ws/{w}/{b}", 200, p - 1):
            x += 1
            s, k, _ = O.almost_inverse_wordsliced(x, p, w, n)
            assert k + 1 >= b and k <= 2 * n * w
            assert abs(s) < p
            assert (x * s - (1 << k)) % p == 0
