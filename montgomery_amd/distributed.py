"""One MSM sharded across the GPUs of one node (one process per GPU, torch.distributed), SURVEY.md section 8(e).

By window (`sharded_msm`): windows are independent until the final Horner step (reference
src/msm-batched-affine.ts:312-333), so rank r computes the partition sums P_k of its contiguous window range with
`msm_window_sums`, the ranks exchange K x 144 bytes with ONE all-gather (RCCL on GPUs, gloo in the CPU tests) and rank 0
finishes with `msm_combine`.  Every rank needs all scalars.
By points (`sharded_msm_points`): rank r runs ALL K windows on its share [n r / G, n (r + 1) / G) of the points -- it needs
only that share of the scalars and decomposes only those -- the same single all-gather carries K x 144 bytes per rank
and rank 0 adds the G sums of every window before the Horner step (`msm_combine_groups`).
An element-wise reduce would be wrong in both forms: limb-wise addition is not the group law.
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


def sharded_msm_points(point_sums: Callable[[int, int], bytes], n: int, K: int, c: int, device="cpu", group=None,
                       curve: int = _lib.CURVE_BLS12_377_G1, timing: Optional[dict] = None):
    """One points-split MSM on the current process group.

    point_sums(first, count) -> K * 144 bytes: the K window sums over this rank's share of the points
    (product: `MsmContext.window_sums(..., point_lo=first)`; the CPU tests inject a checker).
    Returns (True, affine-or-None) on rank 0 and None elsewhere."""
    import time

    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    first, count = point_shards(n, world)[rank]
    # a rank without points contributes K identities: Z = 0 (Weierstrass), (0 : 1 : 1) (twisted Edwards)
    if curve == _lib.CURVE_ED_ON_BLS12_377:
        one = (1).to_bytes(48, "little")
        ident = (bytes(48) + one + one) * K
    else:
        ident = bytes(PARTIAL_BYTES * K)
    part = point_sums(first, count) if count else ident
    if len(part) != PARTIAL_BYTES * K:
        raise MsmError(_lib.MSM_ERR_ARG, "point_sums returned the wrong number of bytes")
    mine = torch.frombuffer(bytearray(part), dtype=torch.uint8).to(device)
    gathered = torch.zeros(world * PARTIAL_BYTES * K, dtype=torch.uint8, device=device)
    t0 = time.perf_counter()
    dist.all_gather_into_tensor(gathered, mine, group=group)
    if timing is not None:
        timing["all_gather_ms"] = (time.perf_counter() - t0) * 1e3
    if rank != 0:
        return None
    return True, combine_groups_host(gathered.cpu().numpy().tobytes(), world, K, c, curve)


def sharded_msm(window_sums: Callable[[int, int], bytes], K: int, c: int, device="cpu", group=None,
                curve: int = _lib.CURVE_BLS12_377_G1, timing: Optional[dict] = None) -> Optional[Tuple[bool, Optional[Tuple[int, int]]]]:
    """Runs one window-sharded MSM on the current process group.

    window_sums(k_lo, k_hi) -> (k_hi - k_lo) * 144 bytes: this rank's partition sums
    (product: `MsmContext.window_sums`, i.e. the HIP path; the CPU tests inject a checker).
    Returns (True, affine-or-None) on rank 0 and None elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    shards = window_shards(K, world)
    lo, hi = shards[rank]
    mine = torch.zeros(PARTIAL_BYTES * K, dtype=torch.uint8, device=device)
    if hi > lo:
        part = window_sums(lo, hi)
        if len(part) != PARTIAL_BYTES * (hi - lo):
            raise MsmError(_lib.MSM_ERR_ARG, "window_sums returned the wrong number of bytes")
        mine[PARTIAL_BYTES * lo : PARTIAL_BYTES * hi] = torch.frombuffer(bytearray(part), dtype=torch.uint8).to(device)
    gathered = torch.zeros(world * PARTIAL_BYTES * K, dtype=torch.uint8, device=device)
    import time

    t0 = time.perf_counter()
    dist.all_gather_into_tensor(gathered, mine, group=group)
    if timing is not None:
        timing["all_gather_ms"] = (time.perf_counter() - t0) * 1e3
    if rank != 0:
        return None
    g = gathered.cpu().numpy().tobytes()
    row = PARTIAL_BYTES * K
    allp = b"".join(g[r * row + PARTIAL_BYTES * a : r * row + PARTIAL_BYTES * b] for r, (a, b) in enumerate(shards))
    return True, combine_host(allp, K, c, curve)
