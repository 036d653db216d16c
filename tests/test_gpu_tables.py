"""Window tables (msm_precompute, include/msm_hip.h): K resident tables 2^(c k) P of a point set, all windows of an MSM in ONE set
of buckets.  The result is a group element, so every run on tables must equal the plain path, the oracle and the golden vectors
bit for bit.  Needs an MI355X: `-m gpu`."""
import json
import os

import pytest

from oracle import msm_oracle as O

pytestmark = pytest.mark.gpu

C = O.BLS12_377
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_golden_4096_points_on_tables(gpu_ctx):
    """tests/golden/msm377_4096.json: the first default-plan call over the whole set builds the tables and runs on them."""
    gold = json.load(open(os.path.join(GOLD, "msm377_4096.json")))
    n = gold["n"]
    pts, _ = O.random_points_bls377(gold["seed_points"], n)
    sc = O.prng_ints(gold["seed_scalars"], n, C.q)
    gpu_ctx.set_points(O.points_to_bytes(pts, 48), check_curve=True)
    assert gpu_ctx.tables_info() == (0, 0, 0)
    sb = O.scalars_to_bytes(sc)
    res, info = gpu_ctx.run(sb)
    assert info["tables"] and gpu_ctx.tables_info()[:2] == (info["c"], info["K"])
    assert res.as_tuple() == (int(gold["result"][0], 16), int(gold["result"][1], 16))
    plain, pinfo = gpu_ctx.run(sb, no_tables=True)
    assert not pinfo["tables"] and plain.as_tuple() == res.as_tuple()
    # a prefix of the points, or another window size, takes the plain path over table 0 -- which is still the plain row table
    part, i2 = gpu_ctx.run(sb[: 32 * 1000])
    assert not i2["tables"] and part.as_tuple() == O.msm_batched_affine(sc[:1000], pts[:1000], c=i2["c"])
    other, i3 = gpu_ctx.run(sb, c=13)
    assert not i3["tables"] and other.as_tuple() == res.as_tuple()
    # new points drop the tables
    gpu_ctx.set_points(O.points_to_bytes(pts[:64], 48))
    assert gpu_ctx.tables_info() == (0, 0, 0)


@pytest.mark.parametrize("lg", [13, 16, 20])
def test_tables_against_known_discrete_logs(gpu_ctx, c_oracle, lg):
    n = 1 << lg
    a = gpu_ctx.generate_points(n, seed=5000 + lg, want_scalars=True, raw=True)
    dev, s = gpu_ctx.generate_scalars(n, seed=6000 + lg, to_host=True, raw=True)
    exp = O.aff_scale(c_oracle.dot_mod(a, s, n, C.q), (C.gx, C.gy), C.p)
    res, info = gpu_ctx.run_device(dev, n)
    assert info["tables"] and (info["c"], info["K"]) == gpu_ctx.plan(n), info
    assert res.as_tuple() == exp
    plain, pinfo = gpu_ctx.run_device(dev, n, no_tables=True)
    assert not pinfo["tables"] and plain.as_tuple() == exp
    ser, _ = gpu_ctx.run_device(dev, n, serial=True)
    assert ser.as_tuple() == exp


def test_explicit_window_sizes_and_degenerate_scalars(gpu_ctx):
    """msm_precompute for an explicit c (folded 18- and 21-bit plans, the 16-bit one, a plan with a short top window); constant
    scalars put all entries of a window group into one or two buckets of the merged window."""
    n = 1 << 13
    gpu_ctx.generate_points(n, seed=77)
    dev, sb = gpu_ctx.generate_scalars(n, seed=78, to_host=True)
    want, _ = gpu_ctx.run_device(dev, n, no_tables=True)
    for c in (16, 18, 21, 13, 19):
        cc, K, nbytes = gpu_ctx.precompute(n, c=c)
        assert cc == c and nbytes == K * n * 256
        got, info = gpu_ctx.run_device(dev, n, c=c)
        assert info["tables"] and info["K"] == K and got.as_tuple() == want.as_tuple(), (c, info)
        for val in (C.q - 1, 1, (1 << 252) - 1, 0):
            const = O.scalars_to_bytes([val] * n)
            a, ia = gpu_ctx.run(const, c=c)
            b, ib = gpu_ctx.run(const, c=c, no_tables=True)
            assert ia["tables"] and not ib["tables"] and a.as_tuple() == b.as_tuple(), (c, hex(val))
    # the default call keeps the plain path while tables of another plan are resident ... unless they are the default's
    d, info = gpu_ctx.run_device(dev, n)
    assert d.as_tuple() == want.as_tuple()


def test_limit_reserve_and_point_sets(gpu_ctx):
    n = 1 << 12
    gpu_ctx.generate_points(n, seed=31)
    dev, _ = gpu_ctx.generate_scalars(n, seed=32)
    gpu_ctx.set_tables_limit(0)
    try:
        r0, i0 = gpu_ctx.run_device(dev, n)
        assert not i0["tables"] and gpu_ctx.tables_info() == (0, 0, 0)
    finally:
        gpu_ctx.set_tables_limit(28 << 30)
    gpu_ctx.reserve(n)                      # msm_reserve: workspace AND tables exist before the first real call
    c, K, nbytes = gpu_ctx.tables_info()
    assert K >= 2 and nbytes == K * n * 256
    r1, i1 = gpu_ctx.run_device(dev, n)
    assert i1["tables"] and r1.as_tuple() == r0.as_tuple()
    # tables travel with their point set
    other = gpu_ctx.pointset_create()
    gpu_ctx.generate_points(n, seed=33)
    assert gpu_ctx.tables_info() == (0, 0, 0)
    r2, i2 = gpu_ctx.run_device(dev, n)
    assert i2["tables"] and r2.as_tuple() != r1.as_tuple()
    gpu_ctx.pointset_select(0)
    assert gpu_ctx.tables_info() == (c, K, nbytes)
    assert gpu_ctx.run_device(dev, n)[0].as_tuple() == r1.as_tuple()
    gpu_ctx.pointset_select(other)
    assert gpu_ctx.run_device(dev, n)[0].as_tuple() == r2.as_tuple()
    gpu_ctx.pointset_destroy(other)


def test_edwards_and_other_curves_on_tables():
    """All four curves build tables by default; the Edwards path finishes the merged window's sums bit-sliced."""
    from montgomery_amd import _lib
    from montgomery_amd.api import MsmContext

    for curve, te in ((_lib.CURVE_ED_ON_BLS12_377, True), (_lib.CURVE_BLS12_381_G1, False), (_lib.CURVE_PALLAS, False)):
        ctx = MsmContext(curve)
        key = (lambda r: (r.x, r.y)) if te else (lambda r: r.as_tuple())
        for n in ((1 << 14), (1 << 18) + 77):
            ctx.generate_points(n, seed=11)
            dev, _ = ctx.generate_scalars(n, seed=12)
            plain, ip = ctx.run_device(dev, n, no_tables=True)
            got, it = ctx.run_device(dev, n)
            assert it["tables"] and not ip["tables"] and key(got) == key(plain), (curve, n)
        ctx.close()


def test_a_call_split_by_points_leaves_the_tables(gpu_ctx):
    """A run on window tables addresses row k * n_points + i from the entry index alone; a call that has to run over ranges of
    the points (a tight msm_set_workspace_limit here, host scalars arriving range by range below) would read other points'
    rows.  Such a call takes the plain path under the same window and says so (msm_result.tables = 0)."""
    n = 1 << 18
    gpu_ctx.generate_points(n, seed=91)
    dev, _ = gpu_ctx.generate_scalars(n, seed=92)
    want, iw = gpu_ctx.run_device(dev, n, no_tables=True)
    on_tab, it = gpu_ctx.run_device(dev, n)
    assert it["tables"] and on_tab.as_tuple() == want.as_tuple()
    gpu_ctx.set_workspace_limit(60 << 20)
    try:
        got, ig = gpu_ctx.run_device(dev, n)
        assert not ig["tables"] and (ig["c"], ig["K"]) == (it["c"], it["K"]), ig
        assert got.as_tuple() == want.as_tuple()
    finally:
        gpu_ctx.set_workspace_limit(0)
    again, ia = gpu_ctx.run_device(dev, n)
    assert ia["tables"] and again.as_tuple() == want.as_tuple()


def test_host_scalars_with_the_tables_window_at_2p24(gpu_ctx):
    """msm_run over HOST scalars of 2^24 points with the explicit window the tables were built for: the scalars cross PCIe range
    by range of the points, so the call must leave the tables (it read the wrong rows and returned MSM_OK)."""
    n = 1 << 24
    gpu_ctx.generate_points(n, seed=93)
    dev, sb = gpu_ctx.generate_scalars(n, seed=94, to_host=True)
    want, _ = gpu_ctx.run_device(dev, n, no_tables=True)
    c, K = gpu_ctx.plan(n)
    on_tab, it = gpu_ctx.run_device(dev, n)
    assert it["tables"] and (it["c"], it["K"]) == (c, K) and on_tab.as_tuple() == want.as_tuple()
    got, ig = gpu_ctx.run(sb, c=c)
    assert not ig["tables"] and ig["c"] == c and got.as_tuple() == want.as_tuple(), ig
    gpu_ctx.set_points(O.points_to_bytes([(C.gx, C.gy)], 48))   # give the 4 GB of rows and 24 GB of tables back


def test_window_tables_over_a_range_of_the_points(gpu_ctx):
    """Round 6: the share of one rank of a points-split run is a RANGE [point_lo, point_lo + n) of the resident points.  Its window
    tables live in a buffer of their own; msm_run / msm_window_sums(merged_sums) over exactly that range run on them.  Built by
    msm_precompute(point_lo), or by the library when a call comes back for the same range a second time in a row."""
    from montgomery_amd import _lib
    from montgomery_amd.distributed import combine_groups_host

    n = 1 << 16
    gpu_ctx.generate_points(n, seed=61)
    dev, _ = gpu_ctx.generate_scalars(n, seed=62)
    full, _ = gpu_ctx.run_device(dev, n, no_tables=True)
    world = 4
    share = n // world
    c, K = gpu_ctx.plan(share, merged=True, point_lo=share)
    assert (c, K) == gpu_ctx.plan(share, merged=True, point_lo=0)

    def shard(g, merged=True):
        return gpu_ctx.window_sums(dev + 32 * g * share, share, 0, K, c=c, on_device=True, point_lo=g * share, merged=merged)

    # walking over the shards builds nothing (no range comes back twice in a row); every shard runs the plain path
    parts = []
    for g in range(world):
        p, info = shard(g)
        assert not info["tables"] and gpu_ctx.tables_info() == (0, 0, 0)
        parts.append(p)
    assert combine_groups_host(b"".join(parts), world, K, c, _lib.CURVE_BLS12_377_G1) == full.as_tuple()
    # the same range again: the tables of the range are built and used; the sums come back merged (identities behind slot 0)
    p1, i1 = shard(world - 1)
    assert i1["tables"] and gpu_ctx.tables_info() == (c, K, K * share * 256) and gpu_ctx.tables_range() == ((world - 1) * share, share)
    ident = lambda part, j: part[144 * j + 96 : 144 * j + 144] == bytes(48)       # Z = 0
    assert not ident(p1, 0) and all(ident(p1, j) for j in range(1, K))
    parts[world - 1] = p1
    assert combine_groups_host(b"".join(parts), world, K, c, _lib.CURVE_BLS12_377_G1) == full.as_tuple()
    # ... and msm_run over that range runs on them too, a prefix of the range or the whole set does not
    r_tab, it = gpu_ctx.run_device(dev + 32 * (world - 1) * share, share, point_lo=(world - 1) * share)
    r_pl, ip = gpu_ctx.run_device(dev + 32 * (world - 1) * share, share, point_lo=(world - 1) * share, no_tables=True)
    assert it["tables"] and not ip["tables"] and r_tab.as_tuple() == r_pl.as_tuple()
    r_sub, isub = gpu_ctx.run_device(dev + 32 * (world - 1) * share, share // 2, point_lo=(world - 1) * share, c=c)
    assert not isub["tables"]
    # without merged_sums msm_window_sums keeps one P_k per slot and the plain path
    p_plain, i_plain = shard(world - 1, merged=False)
    assert not i_plain["tables"] and not any(ident(p_plain, j) for j in range(K))
    parts[world - 1] = p_plain
    assert combine_groups_host(b"".join(parts), world, K, c, _lib.CURVE_BLS12_377_G1) == full.as_tuple()
    # msm_precompute for another range replaces them; every shard on the tables of its own range
    parts = []
    for g in range(world):
        assert gpu_ctx.precompute(share, c=c, point_lo=g * share) == (c, K, K * share * 256)
        assert gpu_ctx.tables_range() == (g * share, share)
        p, info = shard(g)
        assert info["tables"]
        parts.append(p)
    assert combine_groups_host(b"".join(parts), world, K, c, _lib.CURVE_BLS12_377_G1) == full.as_tuple()
    # a window shard on tables: windows [2, 5) through tables 0 .. 2, their sum in slot 2 of the full set of slots
    lo, hi = 2, 5
    pw, iw = gpu_ctx.window_sums(dev + 32 * (world - 1) * share, share, lo, hi, c=c, on_device=True, point_lo=(world - 1) * share, merged=True)
    ref = b"".join(gpu_ctx.window_sums(dev + 32 * (world - 1) * share, share, k, k + 1, c=c, on_device=True, point_lo=(world - 1) * share)[0]
                   for k in range(K))
    mixed = ref[: 144 * lo] + pw + ref[144 * hi:]
    from montgomery_amd.distributed import combine_host
    assert iw["tables"] and combine_host(mixed, K, c, _lib.CURVE_BLS12_377_G1) == combine_host(ref, K, c, _lib.CURVE_BLS12_377_G1)
    # the whole set: its tables go into the row buffer and replace the range's; a range then leaves them alone
    whole, iwh = gpu_ctx.run_device(dev, n)
    assert iwh["tables"] and gpu_ctx.tables_range() == (0, n) and whole.as_tuple() == full.as_tuple()
    for _ in range(2):
        p, info = shard(1)
        assert not info["tables"] and gpu_ctx.tables_range() == (0, n)
    # new points drop everything
    gpu_ctx.generate_points(4096, seed=63)
    assert gpu_ctx.tables_info() == (0, 0, 0) and gpu_ctx.tables_range() == (0, 0)


def test_edwards_range_tables():
    from montgomery_amd import _lib
    from montgomery_amd.api import MsmContext
    from montgomery_amd.distributed import combine_groups_host

    ctx = MsmContext(_lib.CURVE_ED_ON_BLS12_377)
    try:
        n = 1 << 15
        ctx.generate_points(n, seed=71)
        dev, _ = ctx.generate_scalars(n, seed=72)
        full, _ = ctx.run_device(dev, n, no_tables=True)
        share = n // 2
        c, K = ctx.plan(share, merged=True)
        parts = []
        for g in range(2):
            ctx.precompute(share, c=c, point_lo=g * share)
            p, info = ctx.window_sums(dev + 32 * g * share, share, 0, K, c=c, on_device=True, point_lo=g * share, merged=True)
            assert info["tables"], info
            parts.append(p)
        assert combine_groups_host(b"".join(parts), 2, K, c, _lib.CURVE_ED_ON_BLS12_377) == (full.x, full.y)
    finally:
        ctx.close()
