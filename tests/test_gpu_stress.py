"""Shapes the small parity cases do not reach, on every Weierstrass curve: ragged large N, forced window sizes on
either side of the one-level / two-level sort boundary, few distinct scalars (a handful of enormous buckets per
window: deep trees, long tail rounds), short scalars (empty upper windows), two live contexts.  `-m gpu`.
Checked through the discrete logs of the generated points: sum s_i P_i = (sum s_i a_i) G."""
import pytest

from oracle import msm_oracle as O

pytestmark = pytest.mark.gpu


def curve(name):
    from montgomery_amd import _lib

    return {"bls12-377": (_lib.CURVE_BLS12_377_G1, O.BLS12_377), "bls12-381": (_lib.CURVE_BLS12_381_G1, O.BLS12_381),
            "pallas": (_lib.CURVE_PALLAS, O.PALLAS)}[name]


def expected(B, a, s):
    return O.aff_scale(sum(x * y for x, y in zip(a, s)) % B.q, (B.gx, B.gy), B.p)


@pytest.mark.parametrize("name", ["bls12-377", "bls12-381", "pallas"])
@pytest.mark.parametrize("n", [(1 << 17) - 1, 3 * (1 << 18) + 7])
def test_ragged_skewed_short(name, n):
    from montgomery_amd.api import MsmContext

    cid, B = curve(name)
    ctx = MsmContext(cid)
    try:
        a = O.scalars_from_bytes(ctx.generate_points(n, seed=n & 0xFFFF, want_scalars=True))
        dev, sb = ctx.generate_scalars(n, seed=99, to_host=True)
        exp = expected(B, a, O.scalars_from_bytes(sb))
        for c in (None, 12, 19):
            res, info = ctx.run_device(dev, n, c=c)
            assert res.as_tuple() == exp, (name, n, info)
        vals = O.prng_ints(f"stress/{name}", 5, B.q)
        sk = [vals[i % 5] for i in range(n)]
        res, info = ctx.run(O.scalars_to_bytes(sk))
        assert res.as_tuple() == expected(B, a, sk), info
        assert info["max_bucket"] >= n // 5
        sm = [(i * 2654435761) & 0xFFFFF for i in range(n)]
        res, info = ctx.run(O.scalars_to_bytes(sm))
        assert res.as_tuple() == expected(B, a, sm), info
    finally:
        ctx.close()


def test_two_live_contexts_interleaved():
    from montgomery_amd.api import MsmContext

    (c1id, B1), (c2id, B2) = curve("bls12-377"), curve("bls12-381")
    c1, c2 = MsmContext(c1id), MsmContext(c2id)
    try:
        n = 1 << 16
        a1 = O.scalars_from_bytes(c1.generate_points(n, seed=1, want_scalars=True))
        a2 = O.scalars_from_bytes(c2.generate_points(n, seed=2, want_scalars=True))
        d1, s1 = c1.generate_scalars(n, seed=3, to_host=True)
        d2, s2 = c2.generate_scalars(n, seed=4, to_host=True)
        for _ in range(3):
            r1, _i = c1.run_device(d1, n)
            r2, _i = c2.run_device(d2, n)
        assert r1.as_tuple() == expected(B1, a1, O.scalars_from_bytes(s1))
        assert r2.as_tuple() == expected(B2, a2, O.scalars_from_bytes(s2))
    finally:
        c1.close()
        c2.close()


def test_two_contexts_from_two_threads():
    """"Calls on one context are serialised, distinct contexts are independent" (include/msm_hip.h): two host threads, a context
    each, MSMs of different curves and sizes at the same time (ctypes drops the GIL inside the library), host-buffer and
    device-buffer scalars mixed.  Every result must be the one the context gives alone."""
    import threading

    from montgomery_amd.api import MsmContext

    jobs = [(curve("bls12-377"), (1 << 19) + 5, 11), (curve("bls12-381"), (1 << 18) + 9, 12)]
    ctxs, refs, outs, errs = [], [], [[], []], []
    for (cid, B), n, seed in jobs:
        ctx = MsmContext(cid)
        ctx.generate_points(n, seed=seed)
        dev, host = ctx.generate_scalars(n, seed=seed + 100, to_host=True)
        refs.append(ctx.run_device(dev, n)[0].as_tuple())
        ctxs.append((ctx, dev, host, n))

    def work(i):
        try:
            ctx, dev, host, n = ctxs[i]
            for rep in range(6):
                r, _ = ctx.run(host) if rep % 2 else ctx.run_device(dev, n)
                outs[i].append(r.as_tuple())
        except Exception as e:   # noqa: BLE001 -- reported by the assertion below
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    try:
        assert not errs, errs
        for i in range(2):
            assert outs[i] == [refs[i]] * 6
    finally:
        for ctx, *_ in ctxs:
            ctx.close()


def test_degenerate_inputs_through_the_big_window_paths():
    """2^23 copies of one point, then P and -P alternating, at c = 22: every pair of the tree is a doubling or a cancellation,
    and it goes through the three-pass sort, the chunk-ordered round 1 and the element records.  Expected: (sum of the signed
    scalars) P."""
    import numpy as np

    from montgomery_amd.api import MsmContext

    C = O.BLS12_377
    P = O.ZPRIZE_BLS377_POINT
    n = 1 << 23
    rng = np.random.default_rng(23)
    sc = rng.integers(0, 256, size=n * 32, dtype=np.uint8)
    sc[31::32] &= 0x0F                                           # < 2^252 < q
    cols = sc.reshape(n, 32).astype(np.int64)

    def scalar_sum(signs=None):
        m = cols if signs is None else cols * signs[:, None]
        return sum(int(m[:, j].sum()) << (8 * j) for j in range(32))

    ctx = MsmContext()
    try:
        for pts, signs in (([P, P], None), ([P, O.aff_neg(P, C.p)], np.tile(np.array([1, -1], dtype=np.int64), n // 2))):
            ctx.set_points(O.points_to_bytes(pts, 48) * (n // 2))
            k = scalar_sum(signs) % C.q
            res, info = ctx.run(sc.tobytes(), c=22)
            assert info["c"] == 22 and res.as_tuple() == (O.aff_scale(k, P, C.p) if k else None), info
    finally:
        ctx.close()


def test_2p27_points_sort_in_capped_slices():
    """From 2^27 points the first pass of the bin split multiplies its slices instead of growing them (an entry is named by a
    17-bit offset inside its slice, sort_kernels.h BS_SPAN_LOG): the six-window plan the library picks must give the element
    the 16-bit plan (radix split, no such slices) gives, and a forced 18-bit plan as well."""
    from montgomery_amd.api import MsmContext

    n = 1 << 27
    ctx = MsmContext()
    try:
        ctx.generate_points(n, seed=127)
        dev, _ = ctx.generate_scalars(n, seed=227)
        res, info = ctx.run_device(dev, n)
        assert (info["c"], info["K"]) == (21, 6) and not info["tables"], info
        ref, _ = ctx.run_device(dev, n, c=16)
        assert res.as_tuple() == ref.as_tuple()
        r18, i18 = ctx.run_device(dev, n, c=18)
        assert i18["K"] == 7 and r18.as_tuple() == ref.as_tuple()
    finally:
        ctx.close()
