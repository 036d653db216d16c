"""One MSM sharded across the GPUs of one node (one process per GPU, torch.distributed), SURVEY.md section 8(e).

By window (`sharded_msm`): windows are independent until the final Horner step (reference
src/msm-batched-affine.ts:312-333), so rank r computes the partition sums P_k of its contiguous window range with
`msm_window_sums`, the ranks exchange K x 144 bytes with ONE all-gather (RCCL on GPUs, gloo in the CPU tests) and rank 0
finishes with `msm_combine`.  Every rank needs all scalars.
By points (`sharded_msm_points`): rank r runs ALL K windows on its share [n r / G, n (r + 1) / G) of the points -- it needs
only that share of the scalars and decomposes only those -- the same single all-gather carries K x 144 bytes per rank
and rank 0 adds the G sums of every window before the Horner step (`msm_combine_groups`).
By buckets (`sharded_msm_buckets`, round 5): rank r runs all K windows over ALL points but keeps only the digits that fall
into its eighth of every window's buckets (`msm_opts.bucket_shard`) -- it slices all scalars and sorts and adds 1 / G of the
entries, with the single-GPU plan (K = 6 at 2^26: the window split needs K = 8 so that eight ranks divide it).  The reference
splits every window's buckets across its threads the same way (src/msm-common.ts:72-172).  Same exchange and the same
`msm_combine_groups` as the points split: partial sums keep their buckets' true weights.
An element-wise reduce would be wrong in all forms: limb-wise addition is not the group law.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Tuple

from . import _lib
from ._lib import MsmError, MsmResult

PARTIAL_BYTES = 144  # X || Y || Z, 48-byte little-endian canonical integers


def window_shards(K: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced window ranges [lo, hi), one per rank; ranks beyond K get an empty range."""
    base, rem = divmod(K, world)
    out, lo = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((lo, lo + n))
        lo += n
    return out


def combine_host(partials: bytes, K: int, c: int, curve: int = _lib.CURVE_BLS12_377_G1) -> Optional[Tuple[int, int]]:
    """S = sum_k 2^(ck) P_k -> canonical affine (x, y) or None; host arithmetic in libmsm_hip.so, no GPU needed."""
    lib = _lib.load()
    if len(partials) != PARTIAL_BYTES * K:
        raise MsmError(_lib.MSM_ERR_ARG, f"expected {PARTIAL_BYTES * K} bytes of window sums, got {len(partials)}")
    buf = (C.c_uint8 * len(partials)).from_buffer_copy(partials)
    res = MsmResult()
    rc = lib.msm_combine_curve(curve, buf, K, c, C.byref(res))
    if rc != _lib.MSM_OK:
        raise MsmError(rc, "msm_combine failed")
    if res.is_infinity:
        return None
    return int.from_bytes(bytes(res.x), "little"), int.from_bytes(bytes(res.y), "little")


def point_shards(n: int, world: int) -> List[Tuple[int, int]]:
    """(first point, count) of every rank's share: rank r owns [n r / G, n (r + 1) / G), as msm_ctx_create_multi cuts it."""
    return [(n * r // world, n * (r + 1) // world - n * r // world) for r in range(world)]


def choose_window(plan: Callable[[int, Optional[int]], Tuple[int, int]], n: int, world: int, split: str) -> Tuple[int, int]:
    """(c, K) for one MSM of n points sharded over `world` ranks; plan(n, c[, no_tables]) -> (c, K) is the library's `msm_plan`
    (`MsmContext.plan`: asked with no_tables=True, because shards run the plain path whatever tables the context holds).
    All ranks must use the same window: their window sums meet slot by slot.
    by points: every rank runs a whole MSM over n / world points -- the window the library picks for THAT size (on the window
      tables of the rank's range of the points where they fit: `window_sums(..., merged=True)`);
    by windows: a rank runs K / world windows over all points.  The big windows a single GPU takes from 2^24 points (K = 6)
      do not divide among 4 or 8 ranks: the single-GPU plan where the ranks divide its K (2 ranks at 2^26: three 21-bit windows
      each -- since the round-5 sort 72.4 ms against 78.9 for four 16-bit ones, tools/window_shard_time.py), else c = 16
      (K = 8) where they divide that, else the pick for a rank's share of the points."""
    def plain(m, c):
        # shards always run the plain path (msm_window_sums): a `plan` that knows window tables (MsmContext.plan) is asked for
        # the plain plan, one that takes (n, c) only -- the CPU tests' -- is the plain plan already
        try:
            return plan(m, c, no_tables=True)
        except TypeError:
            return plan(m, c)

    def shard(m):
        # a points shard may run on the window tables of its range (window_sums(..., merged=True)): the plan of such a call
        try:
            return plan(m, None, merged=True)
        except TypeError:
            return plain(m, None)

    if world <= 1 or split == "buckets":   # a bucket-range shard keeps the single-GPU plan: every rank runs all of its windows
        return plain(n, None)
    if split != "points":
        # the single-GPU plan where the ranks divide its windows (2^26 on 2 ranks: three 21-bit windows each, 72.4 ms against
        # 78.9 for four 16-bit ones -- profiles/r06_experiments.txt item 9), else K = 8
        c1, K1 = plain(n, None)
        if K1 % world == 0:
            return c1, K1
        c16, K16 = plain(n, 16)
        if K16 % world == 0:
            return c16, K16
    c, _ = shard(max(n // world, 1)) if split == "points" else plain(max(n // world, 1), None)
    return plain(n, c)   # (K for the call the ranks actually make)


def choose_split(n: int, world: int, K: int) -> str:
    """'points' or 'windows' for one MSM of n points over `world` GPUs.  Measured per-rank work on one MI355X (2^26, K = 8,
    tools/shard_time.py, profiles/r03_shard_proxy.txt): points shards 79.9 / 42.8 / 22.3 ms against window shards
    81.1 / 44.5 / 23.8 ms at 2 / 4 / 8 ranks -- a points shard decomposes and sorts only its own scalars, and every rank
    has the same work (the top window is lighter than the others) -- and it needs n / G instead of n scalars per GPU.
    Window shards remain for K divisible by the rank count on small inputs, where a rank's share of the points would drop
    below what fills a GPU."""
    if world <= 1:
        return "windows"
    if n // world >= (1 << 20) or K % world:
        return "points"
    return "windows"


def combine_groups_host(partials: bytes, G: int, K: int, c: int, curve: int = _lib.CURVE_BLS12_377_G1) -> Optional[Tuple[int, int]]:
    """P_k = sum over the G groups, S = sum_k 2^(ck) P_k -> canonical affine (x, y) or None (host arithmetic, no GPU)."""
    lib = _lib.load()
    if len(partials) != PARTIAL_BYTES * K * G:
        raise MsmError(_lib.MSM_ERR_ARG, f"expected {PARTIAL_BYTES * K * G} bytes of window sums, got {len(partials)}")
    buf = (C.c_uint8 * len(partials)).from_buffer_copy(partials)
    res = MsmResult()
    rc = lib.msm_combine_groups(curve, buf, G, K, c, C.byref(res))
    if rc != _lib.MSM_OK:
        raise MsmError(rc, "msm_combine_groups failed")
    if res.is_infinity:
        return None
    return int.from_bytes(bytes(res.x), "little"), int.from_bytes(bytes(res.y), "little")


class ShardExchange:
    """The one collective of a sharded MSM: every rank contributes `row` bytes, every rank receives world x row bytes.
    Buffers live as long as the object: a pinned host row, its device twin and the gathered tensor, so that one step is
    copy-in (asynchronous), all-gather, copy-out with ONE synchronisation at the end (the `.cpu()` of rank 0) instead of a
    fresh tensor and a blocking copy on either side.  On a CPU process group (gloo) the host row is the collective's operand."""

    def __init__(self, row: int, device="cpu", group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.row = row
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.host = torch.zeros(row, dtype=torch.uint8, pin_memory=self.on_gpu)
        self.dev = torch.zeros(row, dtype=torch.uint8, device=self.device) if self.on_gpu else self.host
        self.gathered = torch.zeros(self.world * row, dtype=torch.uint8, device=self.device)

    def all_gather(self, mine: bytes, timing: Optional[dict] = None) -> Optional[bytes]:
        """`mine` (row bytes) from every rank -> world x row bytes on rank 0 (None elsewhere).  timing["all_gather_ms"] is the
        collective alone: device events around it on a GPU group (read after the step's one synchronisation -- a host
        clock around the call would only time the enqueue), the host clock on a CPU group, where the call blocks."""
        import time

        torch = self.torch
        if len(mine) != self.row:
            raise MsmError(_lib.MSM_ERR_ARG, f"expected {self.row} bytes for the all-gather, got {len(mine)}")
        self.host.copy_(torch.frombuffer(bytearray(mine), dtype=torch.uint8))
        ev = None
        if self.on_gpu:
            self.dev.copy_(self.host, non_blocking=True)
            if timing is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
        t0 = time.perf_counter()
        self.dist.all_gather_into_tensor(self.gathered, self.dev, group=self.group)
        if ev:
            ev[1].record()
        elif timing is not None:
            timing["all_gather_ms"] = (time.perf_counter() - t0) * 1e3
        out = self.gathered.cpu().numpy().tobytes() if self.rank == 0 else None   # rank 0: the step's one blocking copy
        if self.on_gpu and self.rank != 0:
            torch.cuda.current_stream(self.device).synchronize()   # the pinned row is rewritten by the next step: its copy must be done
        if ev:
            ev[1].synchronize()
            timing["all_gather_ms"] = ev[0].elapsed_time(ev[1])
        return out


def sharded_msm_points(point_sums: Callable[[int, int], bytes], n: int, K: int, c: int, device="cpu", group=None,
                       curve: int = _lib.CURVE_BLS12_377_G1, timing: Optional[dict] = None, exchange: Optional[ShardExchange] = None):
    """One points-split MSM on the current process group.

    point_sums(first, count) -> K * 144 bytes: the K window sums over this rank's share of the points
    (product: `MsmContext.window_sums(..., point_lo=first, merged=True)` -- the sums may come back merged, the first slot
    carrying sum_k 2^(c k) P_k and the others the identity, as a run on the range's window tables leaves them; the CPU tests
    inject a checker of either form).
    `exchange`: a ShardExchange of K * 144 bytes per rank kept by the caller across steps (default: one per call).
    Returns (True, affine-or-None) on rank 0 and None elsewhere."""
    ex = exchange or ShardExchange(PARTIAL_BYTES * K, device, group)
    first, count = point_shards(n, ex.world)[ex.rank]
    # a rank without points contributes K identities: Z = 0 (Weierstrass), (0 : 1 : 1) (twisted Edwards)
    if curve == _lib.CURVE_ED_ON_BLS12_377:
        one = (1).to_bytes(48, "little")
        ident = (bytes(48) + one + one) * K
    else:
        ident = bytes(PARTIAL_BYTES * K)
    part = point_sums(first, count) if count else ident
    if len(part) != PARTIAL_BYTES * K:
        raise MsmError(_lib.MSM_ERR_ARG, "point_sums returned the wrong number of bytes")
    g = ex.all_gather(part, timing)
    if g is None:
        return None
    return True, combine_groups_host(g, ex.world, K, c, curve)


def bucket_shard_of(rank: int, world: int, span: int, L: Optional[int] = None) -> Tuple[int, int]:
    """[lo, hi) of the bucket indices (l - 1) rank `rank` keeps in a window whose digits cover `span` buckets of its L
    (make_plan, msm_opts.bucket_shard): the g-th of G equal parts of the span; the last part runs to the end of the window."""
    L = span if L is None else L
    return span * rank // world, (L if rank + 1 == world else span * (rank + 1) // world)


def sharded_msm_buckets(bucket_sums: Callable[[int, int], bytes], K: int, c: int, device="cpu", group=None,
                        curve: int = _lib.CURVE_BLS12_377_G1, timing: Optional[dict] = None, exchange: Optional[ShardExchange] = None):
    """One bucket-split MSM on the current process group.

    bucket_sums(rank, world) -> K * 144 bytes: the K window sums over ALL points restricted to this rank's range of the
    buckets (product: `MsmContext.window_sums(..., bucket_shard=(rank, world))`; the CPU tests inject a checker).
    Returns (True, affine-or-None) on rank 0 and None elsewhere."""
    ex = exchange or ShardExchange(PARTIAL_BYTES * K, device, group)
    part = bucket_sums(ex.rank, ex.world)
    if len(part) != PARTIAL_BYTES * K:
        raise MsmError(_lib.MSM_ERR_ARG, "bucket_sums returned the wrong number of bytes")
    g = ex.all_gather(part, timing)
    if g is None:
        return None
    return True, combine_groups_host(g, ex.world, K, c, curve)


def sharded_msm(window_sums: Callable[[int, int], bytes], K: int, c: int, device="cpu", group=None,
                curve: int = _lib.CURVE_BLS12_377_G1, timing: Optional[dict] = None,
                exchange: Optional[ShardExchange] = None) -> Optional[Tuple[bool, Optional[Tuple[int, int]]]]:
    """Runs one window-sharded MSM on the current process group.

    window_sums(k_lo, k_hi) -> (k_hi - k_lo) * 144 bytes: this rank's partition sums
    (product: `MsmContext.window_sums`, i.e. the HIP path; the CPU tests inject a checker).
    `exchange`: a ShardExchange of K * 144 bytes per rank kept by the caller across steps (default: one per call).
    Returns (True, affine-or-None) on rank 0 and None elsewhere."""
    ex = exchange or ShardExchange(PARTIAL_BYTES * K, device, group)
    shards = window_shards(K, ex.world)
    lo, hi = shards[ex.rank]
    mine = bytearray(PARTIAL_BYTES * K)   # the row of every rank has all K slots; a rank fills those of its windows
    if hi > lo:
        part = window_sums(lo, hi)
        if len(part) != PARTIAL_BYTES * (hi - lo):
            raise MsmError(_lib.MSM_ERR_ARG, "window_sums returned the wrong number of bytes")
        mine[PARTIAL_BYTES * lo : PARTIAL_BYTES * hi] = part
    g = ex.all_gather(bytes(mine), timing)
    if g is None:
        return None
    row = PARTIAL_BYTES * K
    allp = b"".join(g[r * row + PARTIAL_BYTES * a : r * row + PARTIAL_BYTES * b] for r, (a, b) in enumerate(shards))
    return True, combine_host(allp, K, c, curve)
