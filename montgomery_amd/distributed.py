"""Window-sharded MSM across the GPUs of one node (one process per GPU, torch.distributed).

Windows are independent until the final Horner step (reference src/msm-batched-affine.ts:312-333), so
rank r computes the partition sums P_k of its contiguous window range with `msm_window_sums`, the ranks
exchange K x 144 bytes with ONE all-gather (RCCL on GPUs, gloo in the CPU tests) and rank 0 finishes with
`msm_combine`.  An element-wise reduce would be wrong: limb-wise addition is not the group law.
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


def sharded_msm(window_sums: Callable[[int, int], bytes], K: int, c: int, device="cpu", group=None,
                curve: int = _lib.CURVE_BLS12_377_G1) -> Optional[Tuple[bool, Optional[Tuple[int, int]]]]:
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
    dist.all_gather_into_tensor(gathered, mine, group=group)
    if rank != 0:
        return None
    g = gathered.cpu().numpy().tobytes()
    row = PARTIAL_BYTES * K
    allp = b"".join(g[r * row + PARTIAL_BYTES * a : r * row + PARTIAL_BYTES * b] for r, (a, b) in enumerate(shards))
    return True, combine_host(allp, K, c, curve)
