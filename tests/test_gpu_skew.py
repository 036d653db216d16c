"""Skewed scalar distributions (montgomery_amd/workloads.py): the sort's heavy bins are cut into parts (sort_kernels.h, "Parts of
heavy bins"), deep buckets go through the tail rounds.  The result is a group element: every distribution must give
(sum s_i a_i) G for generated points P_i = a_i G, whatever path the sizes select.  The reference runs any bucket-size
distribution through the same rounds (src/msm-batched-affine.ts:204,243-263).  Needs an MI355X: `-m gpu`."""
import numpy as np
import pytest

from montgomery_amd import workloads
from oracle import msm_oracle as O

pytestmark = pytest.mark.gpu

C = O.BLS12_377
G = (C.gx, C.gy)


def expected(c_oracle, a, s_np, n):
    return O.aff_scale(c_oracle.dot_mod(a, s_np.tobytes(), n, C.q), G, C.p)


@pytest.mark.parametrize("lg,c,kind", [
    (18, None, "one"),       # window tables, merged window: seven windows' entries in seven buckets
    (18, 21, "one"),         # bin split + padded slots, every window one bin of 2^19 records
    (20, None, "prover"),    # tables
    (20, 21, "prover"),
    (22, None, "prover"),    # seven folded 18-bit windows, slots
    (22, None, "one"),
    (21, 16, "one"),         # radix split (one block per coarse bin)
])
def test_skewed_scalars_small(gpu_ctx, c_oracle, lg, c, kind):
    n = 1 << lg
    a = gpu_ctx.generate_points(n, seed=900 + lg, want_scalars=True, raw=True)
    s = workloads.scalars(kind, n, seed=lg)
    exp = expected(c_oracle, a, s, n)
    res, info = gpu_ctx.run(s.tobytes(), c=c)
    assert res.as_tuple() == exp, (lg, c, kind, info)
    if c is None:
        plain, ip = gpu_ctx.run(s.tobytes(), no_tables=True)
        assert plain.as_tuple() == exp, (lg, kind, ip)


@pytest.mark.parametrize("lg,kind", [(23, "one"), (24, "one"), (24, "prover")])
def test_skewed_scalars_through_the_tile_ordered_round(gpu_ctx, c_oracle, lg, kind):
    """2^23: seven 18-bit windows (padded slots); 2^24: six 21-bit windows, the pairs of round 1 emitted by the parts of the
    one heavy bin of every window (k_bin_pairs), 24 tail rounds behind them."""
    n = 1 << lg
    a = gpu_ctx.generate_points(n, seed=950 + lg, want_scalars=True, raw=True)
    s = workloads.scalars(kind, n, seed=lg)
    exp = expected(c_oracle, a, s, n)
    dev = gpu_ctx.device_alloc(32 * n)
    try:
        gpu_ctx.device_upload(dev, s.tobytes())
        res, info = gpu_ctx.run_device(dev, n, no_tables=True)
        assert res.as_tuple() == exp, (lg, kind, info)
        if kind == "one":
            assert info["max_bucket"] >= n, info      # every entry of a half scalar's window in one bucket
        if lg == 24:
            res22, i22 = gpu_ctx.run_device(dev, n, c=22)
            assert res22.as_tuple() == exp, (lg, kind, i22)
    finally:
        gpu_ctx.device_free(dev)


def test_one_scalar_repeated_over_degenerate_points(gpu_ctx):
    """The two point patterns of tools/adversarial_big.py at 2^23 with ONE scalar repeated: 2^23 copies of one point (every pair
    of the tree is a doubling), and P, -P alternating (every pair cancels): expected n s P and the identity."""
    n = 1 << 23
    P = O.ZPRIZE_BLS377_POINT
    s = workloads.scalars("one", n, seed=23)
    s0 = int.from_bytes(bytes(s[0]), "little")
    sb = s.tobytes()
    for pts, exp in (([P, P], O.aff_scale(s0 * n % C.q, P, C.p)), ([P, O.aff_neg(P, C.p)], None)):
        gpu_ctx.set_points(O.points_to_bytes(pts, 48) * (n // 2))
        for c in (None, 22):
            res, info = gpu_ctx.run(sb, c=c, no_tables=True)
            assert res.as_tuple() == exp, (c, info)
    gpu_ctx.set_points(O.points_to_bytes([P], 48))


@pytest.mark.parametrize("kind", ["one", "prover"])
def test_skewed_scalars_on_the_edwards_path(c_oracle, kind):
    """Ed-on-BLS12-377 at 2^22: the 18-bit windows of the plain path take the bin split (padded slots), whose heavy bins are cut
    into parts like the Weierstrass ones; and the same input on window tables (one merged window).  Scalars above the 251-bit
    group order are reduced by the library (msm_opts.strict = 0), the expected value is taken mod q as well."""
    from montgomery_amd import _lib
    from montgomery_amd.api import MsmContext

    E = O.ED_ON_BLS12_377
    n = 1 << 22
    ctx = MsmContext(_lib.CURVE_ED_ON_BLS12_377)
    try:
        a = ctx.generate_points(n, seed=977, want_scalars=True, raw=True)
        s = workloads.scalars(kind, n, seed=22)
        k = c_oracle.dot_mod(a, s.tobytes(), n, E.q)
        exp = O.te_to_affine(O.te_scale(k, O.te_from_affine((E.gx, E.gy), E), E), E)
        plain, ip = ctx.run(s.tobytes(), no_tables=True)
        assert ip["c"] == 18 and (plain.x, plain.y) == exp, ip
        tab, it = ctx.run(s.tobytes())
        assert it["tables"] and (tab.x, tab.y) == exp, it
    finally:
        ctx.close()
