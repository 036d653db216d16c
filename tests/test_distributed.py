"""N > 1 path on CPU: world_size 2, gloo.  The window sums of each rank come from the oracle (there is no
GPU here); sharding, the all-gather layout and the host-side Horner combination are the product's."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_window_sums(scalars, points, c, K, C=None, buckets=None):
    """P_k for every window k as 144-byte (X, Y, Z) records, from the oracle's spec arithmetic.
    buckets = (lo, hi): only the digits whose bucket index l - 1 lies in [lo, hi) (a bucket-range shard)."""
    from oracle import msm_oracle as O

    C = C or O.BLS12_377
    g = O.glv_params(C.q, C.lam)
    sums = [None] * K
    for s, P in zip(scalars, points):
        a0, a1, n0, n1 = O.glv_decompose(s, g)
        for a, neg, Q in ((a0, n0, P), (a1, n1, (C.beta * P[0] % C.p, P[1]))):
            for k, (l, dneg) in enumerate(O.signed_digits(a, c, K)):
                if l and (buckets is None or buckets[0] <= l - 1 < buckets[1]):
                    T = O.aff_scale(l, Q, C.p)
                    sums[k] = O.aff_add(sums[k], O.aff_neg(T, C.p) if neg ^ dneg else T, C.p)
    out = []
    for S in sums:
        X, Y, Z = (0, 1, 0) if S is None else (S[0] * 7 % C.p, S[1] * 7 % C.p, 7)   # any projective representative
        out.append(X.to_bytes(48, "little") + Y.to_bytes(48, "little") + Z.to_bytes(48, "little"))
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from montgomery_amd.distributed import bucket_shard_of, sharded_msm, sharded_msm_buckets, sharded_msm_points
    from oracle import msm_oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        C = O.BLS12_377
        results = []
        # bucket split: every rank sums ALL windows over ALL points, restricted to its range of the buckets
        for name, n, c in (("ba", 40, 5), ("bb", 3, 9)):   # 2^4 = 16 buckets over 2 or 3 ranks: uneven ranges
            pts, _ = O.random_points_bls377("dist/" + name, n)
            sc = O.prng_ints("dist/s/" + name, n, C.q)
            K = -(-127 // c)
            out = sharded_msm_buckets(lambda r, w: b"".join(_oracle_window_sums(sc, pts, c, K, buckets=bucket_shard_of(r, w, 1 << (c - 1)))), K, c)
            if rank == 0:
                results.append((out[1], O.msm_batched_affine(sc, pts, c=c)))
            else:
                assert out is None
        # points split: every rank sums ALL windows over its share of the points; rank 0 adds the groups per window
        for name, n, c in (("pa", 23, 9), ("pb", 1, 13)):   # n = 1: rank 0 has no points at all
            pts, _ = O.random_points_bls377("dist/" + name, n)
            sc = O.prng_ints("dist/s/" + name, n, C.q)
            K = -(-127 // c)
            timing = {}
            out = sharded_msm_points(lambda first, count: b"".join(_oracle_window_sums(sc[first:first + count], pts[first:first + count], c, K)),
                                     n, K, c, timing=timing)
            assert "all_gather_ms" in timing
            if rank == 0:
                results.append((out[1], O.msm_batched_affine(sc, pts, c=c)))
            else:
                assert out is None
        # points split over the TABLES path: a rank's sums come back merged (msm_opts.merged_sums) -- its first slot carries
        # sum_k 2^(c k) P_k, the others the identity (Z = 0) -- and combine exactly like one P_k per slot
        for name, n, c in (("ta", 19, 7), ("tb", 2, 11)):
            pts, _ = O.random_points_bls377("dist/" + name, n)
            sc = O.prng_ints("dist/s/" + name, n, C.q)
            K = -(-127 // c)

            def merged_sums(first, count):
                tot = O.msm_naive_affine(sc[first:first + count], pts[first:first + count], C) if count else None
                X, Y, Z = (0, 1, 0) if tot is None else (tot[0] * 5 % C.p, tot[1] * 5 % C.p, 5)
                return X.to_bytes(48, "little") + Y.to_bytes(48, "little") + Z.to_bytes(48, "little") + bytes(144 * (K - 1))

            out = sharded_msm_points(merged_sums, n, K, c)
            if rank == 0:
                results.append((out[1], O.msm_batched_affine(sc, pts, c=c)))
            else:
                assert out is None
        for name, n, c in (("a", 24, 9), ("b", 5, 16), ("cancel", 2, 7)):
            pts, _ = O.random_points_bls377("dist/" + name, n)
            sc = O.prng_ints("dist/s/" + name, n, C.q)
            if name == "cancel":
                pts = [pts[0], pts[0]]
                sc = [11, C.q - 11]
            K = -(-127 // c)
            allw = _oracle_window_sums(sc, pts, c, K)
            out = sharded_msm(lambda lo, hi: b"".join(allw[lo:hi]), K, c)
            if rank == 0:
                results.append((out[1], O.msm_batched_affine(sc, pts, c=c)))
            else:
                assert out is None
        if rank == 0:
            q.put(results)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_window_shards_partition():
    from montgomery_amd.distributed import window_shards

    for K in (1, 5, 8, 10, 32):
        for world in (1, 2, 3, 4, 8, 16):
            sh = window_shards(K, world)
            assert len(sh) == world and sh[0][0] == 0 and sh[-1][1] == K
            assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
            sizes = [hi - lo for lo, hi in sh]
            assert max(sizes) - min(sizes) <= 1


def test_combine_host_matches_oracle():
    from montgomery_amd.distributed import combine_host
    from oracle import msm_oracle as O

    C = O.BLS12_377
    pts, _ = O.random_points_bls377("dist/combine", 12)
    sc = O.prng_ints("dist/combine/s", 12, C.q)
    for c in (4, 13):
        K = -(-127 // c)
        allw = _oracle_window_sums(sc, pts, c, K)
        assert combine_host(b"".join(allw), K, c) == O.msm_batched_affine(sc, pts, c=c)


def test_combine_host_other_curves():
    """`msm_combine_curve`: rank 0 combines for BLS12-381 / Pallas without a context (host arithmetic only)."""
    from montgomery_amd import _lib
    from montgomery_amd.distributed import combine_host
    from oracle import msm_oracle as O

    for cid, B in ((_lib.CURVE_BLS12_381_G1, O.BLS12_381), (_lib.CURVE_PALLAS, O.PALLAS)):
        pts, _ = O.random_points_bls377(f"dist/combine/{B.label}", 10, B)
        sc = O.prng_ints(f"dist/combine/{B.label}/s", 10, B.q)
        for c in (5, 16):
            K = -(-128 // c)
            allw = _oracle_window_sums(sc, pts, c, K, B)
            assert combine_host(b"".join(allw), K, c, curve=cid) == O.msm_batched_affine(sc, pts, B, c=c)
    from montgomery_amd import MsmError

    with pytest.raises(MsmError):
        combine_host(b"\0" * 144, 1, 4, curve=9)


def test_combine_host_edwards():
    """Twisted Edwards shards travel as (X : Y : Z); the combine rebuilds T = X Y / Z projectively and runs the
    unified-addition Horner (src/msm-basic.ts:142-158)."""
    from montgomery_amd import _lib
    from montgomery_amd.distributed import combine_host
    from oracle import msm_oracle as O

    E = O.ED_ON_BLS12_377
    pts, _ = O.random_points_ed377("dist/ed", 9)
    sc = O.prng_ints("dist/ed/s", 9, E.q)
    for c in (4, 14):
        K = -(-(E.q.bit_length() + 1) // c)
        sums = [O.te_from_affine((0, 1), E)] * K
        for s, P in zip(sc, pts):
            for k, (l, neg) in enumerate(O.signed_digits(s, c, K)):
                if l:
                    T = O.te_scale(l, O.te_from_affine(P, E), E)
                    sums[k] = O.te_add(sums[k], O.te_neg(T, E) if neg else T, E)
        parts = b""
        for k, S in enumerate(sums):
            x, y = O.te_to_affine(S, E)
            lam = 7 + k   # any projective representative
            parts += b"".join((v * lam % E.p).to_bytes(48, "little") for v in (x, y, 1))
        assert combine_host(parts, K, c, curve=_lib.CURVE_ED_ON_BLS12_377) == O.msm_basic_te(sc, pts, c=c)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])   # 3: windows and points that do not divide evenly, a rank without points
def test_sharded_msm_gloo(world):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(results) == 9   # 2 bucket splits, 2 points splits, 2 points splits on merged sums (the tables path), 3 window splits
    for got, exp in results:
        assert got == exp


def test_point_shards_partition_and_group_combine():
    """The share of every rank as msm_ctx_create_multi cuts it, and msm_combine_groups = sum over the groups per window,
    then the Horner step (checked against the oracle with the points split three ways, one share empty)."""
    from montgomery_amd.distributed import combine_groups_host, point_shards
    from oracle import msm_oracle as O

    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            sh = point_shards(n, world)
            assert sh[0][0] == 0 and sum(c for _, c in sh) == n
            assert all(a[0] + a[1] == b[0] for a, b in zip(sh, sh[1:]))
            assert max(c for _, c in sh) - min(c for _, c in sh) <= 1
    C = O.BLS12_377
    pts, _ = O.random_points_bls377("dist/groups", 11)
    sc = O.prng_ints("dist/groups/s", 11, C.q)
    c = 6
    K = -(-127 // c)
    parts = b""
    for first, count in ((0, 5), (5, 0), (5, 6)):
        parts += b"".join(_oracle_window_sums(sc[first:first + count], pts[first:first + count], c, K))
    assert combine_groups_host(parts, 3, K, c) == O.msm_batched_affine(sc, pts, c=c)


def test_bucket_shards_partition():
    from montgomery_amd.distributed import bucket_shard_of

    for L in (16, 1 << 15, 1 << 21):
        for world in (1, 2, 3, 8):
            sh = [bucket_shard_of(r, world, L) for r in range(world)]
            assert sh[0][0] == 0 and sh[-1][1] == L and all(a[1] == b[0] for a, b in zip(sh, sh[1:]))


def test_choose_window_for_shards():
    """The window of a sharded MSM: by points the pick for a rank's share of the points; by windows a K the ranks divide
    (the single-GPU pick at 2^26, c = 22 / K = 6, would leave two of eight ranks idle)."""
    from montgomery_amd.distributed import choose_window

    def plan(m, c):   # msm_plan of the BLS12-377 context: b + 1 = 127 bits, the carry bit folded for c = 18 and 21
        c = c or (21 if m >= 1 << 24 else 18 if m >= 1 << 22 else 16)
        return c, -(-127 // c) - (1 if c in (18, 21) else 0)

    n = 1 << 26
    assert choose_window(plan, n, 1, "windows") == (21, 6)
    assert choose_window(plan, n, 2, "windows") == (21, 6)              # the single-GPU plan divides: three windows per rank
    assert choose_window(plan, n, 6, "windows") == (21, 6)
    for world in (4, 8):
        assert choose_window(plan, n, world, "windows") == (16, 8)      # K = 8 divides: every rank the same number of windows
    assert choose_window(plan, n, 2, "points") == (21, 6)               # a share of 2^25 points: the library's pick for it
    assert choose_window(plan, n, 4, "points") == (21, 6)
    assert choose_window(plan, n, 8, "points") == (18, 7)               # 2^23 points per rank
    assert choose_window(plan, n, 3, "windows") == (21, 6)              # K = 6 divides by 3
    assert choose_window(plan, 1 << 20, 8, "windows") == (16, 8)
    for world in (2, 3, 8):
        assert choose_window(plan, n, world, "buckets") == (21, 6)        # a bucket-range shard keeps the single-GPU plan

    # a plan that knows window tables (MsmContext.plan): shards of the windows and of the buckets ask for the plain plan, a
    # shard of the points for the plan of a call that may run on the tables of its range
    asked = []

    def plan2(m, c, no_tables=False, merged=False):
        asked.append((m, no_tables, merged))
        if c is None and not no_tables:
            c = 18 if m >= 1 << 16 else 16       # tables: 18-bit windows from 2^16 points
        return plan(m, c)

    assert choose_window(plan2, 1 << 20, 8, "points") == (18, 7) and ((1 << 17), False, True) in asked
    assert choose_window(plan2, 1 << 20, 8, "windows") == (16, 8)
    assert choose_window(plan2, 1 << 20, 8, "buckets") == (16, 8)
