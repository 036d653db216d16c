#!/usr/bin/env python3
"""Generates the committed golden vectors in tests/golden/*.json from the CPU oracle.

The reference (TypeScript + wasmati) cannot run in this image, and it stores no golden outputs of its
own (every test there is differential or algebraic), so the vectors below are
produced by oracle/msm_oracle.py -- which tests/test_oracle_kat.py pins against the reference's
known-answer material -- with SEEDED inputs.  Re-run from the repo root:
    python tests/golden/make_golden.py [file.json ...]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import msm_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
C = O.BLS12_377
E = O.ED_ON_BLS12_377


def hx(v):
    return hex(v)


def field_vectors():
    p = C.p
    special = [0, 1, 2, p - 1, p - 2, (p + 1) // 2, (1 << 376), (1 << 377) - 1 - ((1 << 377) - 1 >= p) * ((1 << 377) - p)]
    vals = special + O.prng_ints("golden/fp377", 24, p)
    out = []
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        out.append({
            "a": hx(a), "b": hx(b),
            "mul": hx(a * b % p), "add": hx((a + b) % p), "sub": hx((a - b) % p), "sqr": hx(a * a % p),
            "inv": hx(pow(a, -1, p)) if a else None,
        })
    return {"modulus": hx(p), "cases": out}


def glv_vectors():
    g = O.glv_params(C.q, C.lam)
    scalars = [0, 1, 2, C.q - 1, C.q - 2, C.lam, C.lam - 1, C.lam + 1, C.q // 2, (1 << 252), (1 << 116) - 1, 1 << 116] + O.prng_ints("golden/glv", 52, C.q)
    cases = []
    for s in scalars:
        a0, a1, n0, n1 = O.glv_decompose(s, g)
        cases.append({"s": hx(s), "s0": hx(a0), "s1": hx(a1), "neg0": n0, "neg1": n1})
    return {
        "q": hx(C.q), "lambda": hx(C.lam), "max_bits": g.max_bits, "m": g.m, "k": g.k,
        "v": [hx(g.v00), hx(g.v01), str(g.v10), hx(g.v11)], "m0": str(g.m0), "m1": str(g.m1),
        "cases": cases,
    }


def point_add_vectors():
    pts, _ = O.random_points_bls377("golden/pts", 12)
    p = C.p
    cases = []

    def enc(P):
        return None if P is None else [hx(P[0]), hx(P[1])]

    pairs = [(pts[i], pts[i + 1]) for i in range(0, 8, 2)]
    pairs += [(pts[0], pts[0]), (pts[1], O.aff_neg(pts[1], p)), (None, pts[2]), (pts[3], None), (None, None)]
    pairs += [((C.gx, C.gy), (C.gx, C.gy)), ((p - 1, 0), (p - 1, 0))]  # generator doubling; 2-torsion point (-1, 0)
    for g, h in pairs:
        cases.append({"g": enc(g), "h": enc(h), "sum": enc(O.aff_add(g, h, p))})
    return {"cases": cases}


def msm_vectors():
    cases = []
    pts, ks = O.random_points_bls377("golden/msm", 4096)
    G = (C.gx, C.gy)
    P = O.ZPRIZE_BLS377_POINT

    def add(name, scalars, points, c=None):
        res = O.msm_batched_affine(scalars, points, c=c)
        cases.append({
            "name": name, "c": c, "n": len(scalars),
            "scalars": O.scalars_to_bytes(scalars).hex(),
            "points": O.points_to_bytes([(0, 0) if Q is None else Q for Q in points], 48).hex(),
            "result": None if res is None else [hx(res[0]), hx(res[1])],
        })
        return res

    # reference KAT: 2P + (q-1)P = P (scripts/zprize23/submission-test-bls377.ts:17-27)
    r = add("zprize_2P_minus_P", [2, C.q - 1], [P, P])
    assert r == P
    add("single_generator", [1], [G])
    add("single_random", O.prng_ints("golden/s1", 1, C.q), pts[:1])
    add("n3_c4", O.prng_ints("golden/s3", 3, C.q), pts[:3], 4)
    add("n7_default", O.prng_ints("golden/s7", 7, C.q), pts[:7])
    add("n64_c7", O.prng_ints("golden/s64", 64, C.q), pts[:64], 7)
    add("n100_c5_ragged", O.prng_ints("golden/s100", 100, C.q), pts[:100], 5)
    # edge cases: zero scalars, q-1, repeated points, P and -P, all equal, identity inputs
    sc = O.prng_ints("golden/edge", 40, C.q)
    sc[0] = 0; sc[1] = C.q - 1; sc[2] = 1; sc[3] = 0
    epts = list(pts[:40])
    epts[5] = epts[4]                       # repeated point -> P + P
    sc[5] = sc[4]
    epts[7] = O.aff_neg(epts[6], C.p)       # P and -P with equal scalars -> identity in a bucket
    sc[7] = sc[6]
    epts[9] = None                          # identity input
    add("edge_mix_c6", sc, epts, 6)
    same = O.prng_ints("golden/same", 33, C.q)
    r = add("all_same_point", same, [P] * 33, 5)
    assert r == O.aff_scale(sum(same) % C.q, P, C.p)   # submission-test-bls377.ts:29-45
    add("all_zero_scalars", [0] * 9, pts[:9], 4)
    add("cancel_to_identity", [5, C.q - 5], [pts[0], pts[0]], 4)
    # a mid-size case checked through the known discrete logs
    s1k = O.prng_ints("golden/s1k", 1024, C.q)
    r = add("n1024_c9", s1k, pts[:1024], 9)
    assert r == O.aff_scale(sum(a * b for a, b in zip(s1k, ks[:1024])) % C.q, G, C.p)
    return {"cases": cases}


def msm_large_vector():
    """N = 2^12 (the largest size the reference's own msm.test.ts uses): inputs are regenerated from
    the seed by the test, only the result is stored."""
    pts, ks = O.random_points_bls377("golden/msm4096", 4096)
    sc = O.prng_ints("golden/s4096", 4096, C.q)
    G = (C.gx, C.gy)
    res = O.aff_scale(sum(a * b for a, b in zip(sc, ks)) % C.q, G, C.p)
    return {"seed_points": "golden/msm4096", "seed_scalars": "golden/s4096", "n": 4096, "result": [hx(res[0]), hx(res[1])]}


def ed_vectors():
    x, y, t = O.ZPRIZE_ED377_POINT
    pts, ks = O.random_points_ed377("golden/ed", 64)
    cases = []

    def add(name, scalars, points, c=None):
        res = O.msm_basic_te(scalars, points, c=c)
        cases.append({"name": name, "c": c, "n": len(scalars), "scalars": O.scalars_to_bytes(scalars).hex(),
                      "points": O.points_to_bytes(points, 32).hex(), "result": [hx(res[0]), hx(res[1])]})
        return res

    r = add("zprize_2P_minus_P", [2, E.q - 1], [(x, y), (x, y)], 4)
    assert r == (x, y)   # scripts/zprize23/submission-test.ts:12-20
    add("n64_c6", O.prng_ints("golden/eds", 64, E.q), pts, 6)
    add("n10_c4", O.prng_ints("golden/eds10", 10, E.q), pts[:10], 4)
    return {"cases": cases}


def bls381_vectors():
    return curve_vectors(O.BLS12_381, "381")


def pallas_vectors():
    return curve_vectors(O.PALLAS, "pallas")


def curve_vectors(B, tag):
    """BLS12-381 G1 / Pallas (src/concrete/bls12-381.params.ts, pasta.params.ts; the reference covers both in
    src/msm.test.ts:29-31): field, GLV and MSM vectors in one file."""
    p, q = B.p, B.q
    g = O.glv_params(q, B.lam)
    vals = [0, 1, 2, p - 1, p - 2, (p + 1) // 2, 1 << (p.bit_length() - 1), (1 << 30) - 1] + O.prng_ints(f"golden/fp{tag}", 24, p)
    fp = []
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        fp.append({"a": hx(a), "b": hx(b), "mul": hx(a * b % p), "add": hx((a + b) % p), "sub": hx((a - b) % p),
                   "sqr": hx(a * a % p), "inv": hx(pow(a, -1, p)) if a else None})
    scalars = [0, 1, 2, q - 1, q - 2, B.lam, B.lam - 1, B.lam + 1, q // 2, 1 << 254, (1 << 127) - 1, 1 << 127] + O.prng_ints(f"golden/glv{tag}", 52, q)
    glv = []
    for s in scalars:
        a0, a1, n0, n1 = O.glv_decompose(s, g)
        glv.append({"s": hx(s), "s0": hx(a0), "s1": hx(a1), "neg0": n0, "neg1": n1})
    pts, ks = O.random_points_bls377(f"golden/msm{tag}", 1024, B)
    G = (B.gx, B.gy)
    cases = []

    def add(name, sc, points, c=None):
        res = O.msm_batched_affine(sc, points, B, c=c)
        cases.append({"name": name, "c": c, "n": len(sc), "scalars": O.scalars_to_bytes(sc).hex(),
                      "points": O.points_to_bytes([(0, 0) if Q is None else Q for Q in points], 48).hex(),
                      "result": None if res is None else [hx(res[0]), hx(res[1])]})
        return res

    assert add("2G_minus_G", [2, q - 1], [G, G]) == G
    add("single_generator", [1], [G])
    add("n3_c4", O.prng_ints(f"golden/{tag}/s3", 3, q), pts[:3], 4)
    add("n64_c7", O.prng_ints(f"golden/{tag}/s64", 64, q), pts[:64], 7)
    add("n100_c5_ragged", O.prng_ints(f"golden/{tag}/s100", 100, q), pts[:100], 5)
    sc = O.prng_ints(f"golden/{tag}/edge", 40, q)
    sc[0] = 0; sc[1] = q - 1; sc[2] = 1; sc[3] = 0
    epts = list(pts[:40])
    epts[5] = epts[4]; sc[5] = sc[4]
    epts[7] = O.aff_neg(epts[6], p); sc[7] = sc[6]
    epts[9] = None
    add("edge_mix_c6", sc, epts, 6)
    add("all_zero_scalars", [0] * 9, pts[:9], 4)
    add("cancel_to_identity", [5, q - 5], [pts[0], pts[0]], 4)
    s1k = O.prng_ints(f"golden/{tag}/s1k", 1024, q)
    r = add("n1024_c9", s1k, pts, 9)
    assert r == O.aff_scale(sum(a * b for a, b in zip(s1k, ks)) % q, G, p)
    return {
        "modulus": hx(p), "q": hx(q), "lambda": hx(B.lam), "max_bits": g.max_bits, "m": g.m, "k": g.k,
        "v": [hx(g.v00), hx(g.v01), str(g.v10), hx(g.v11)], "m0": str(g.m0), "m1": str(g.m1),
        "fp": fp, "glv": glv, "msm": cases,
    }


def main():
    only = set(sys.argv[1:])
    files = {
        "fp377.json": field_vectors,
        "glv377.json": glv_vectors,
        "point_add377.json": point_add_vectors,
        "msm377.json": msm_vectors,
        "msm377_4096.json": msm_large_vector,
        "msm_ed377.json": ed_vectors,
        "bls381.json": bls381_vectors,
        "pallas.json": pallas_vectors,
    }
    for name, fn in files.items():
        if only and name not in only:
            continue
        data = fn()
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(data, f, indent=1)
        print("wrote", name)


if __name__ == "__main__":
    main()
