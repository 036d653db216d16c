"""ctypes loader for the C port of the oracle (oracle/msm_oracle.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libmsm_oracle.so")
_lib = None


def build() -> None:
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        lib = C.CDLL(_PATH)
        lib.oracle_msm_bls377.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.oracle_msm_bls377.restype = C.c_int
        lib.oracle_glv_decompose.argtypes = [C.c_void_p, C.c_void_p]
        lib.oracle_glv_decompose.restype = None
        lib.oracle_fp_op.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_fp_op.restype = None
        lib.oracle_window_size.argtypes = [C.c_int]
        lib.oracle_window_size.restype = C.c_int
        lib.oracle_cpu_quota.argtypes = []
        lib.oracle_cpu_quota.restype = C.c_int
        lib.oracle_dot_u256.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        lib.oracle_dot_u256.restype = None
        _lib = lib
    return _lib


def msm_bls377(points: bytes, scalars: bytes, c: int = 0) -> Tuple[Optional[Tuple[int, int]], int]:
    """Returns (affine result or None for the identity, threads used)."""
    lib = load()
    n = len(scalars) // 32
    assert len(points) == 96 * n
    out = (C.c_uint8 * 96)()
    inf, thr = C.c_int(0), C.c_int(0)
    # ctypes arrays are handed over as they are (the timing leg maps 6.4 GB of points at 2^26: no copies inside the timed call)
    pb = points if isinstance(points, C.Array) else (C.c_uint8 * max(len(points), 1)).from_buffer_copy(points or b"\0")
    sb = scalars if isinstance(scalars, C.Array) else (C.c_uint8 * max(len(scalars), 1)).from_buffer_copy(scalars or b"\0")
    rc = lib.oracle_msm_bls377(pb, sb, n, c, out, C.byref(inf), C.byref(thr))
    if rc != 0:
        raise ValueError(f"oracle_msm_bls377 failed: {rc}")
    if inf.value:
        return None, thr.value
    b = bytes(out)
    return (int.from_bytes(b[:48], "little"), int.from_bytes(b[48:], "little")), thr.value


def glv_decompose(s: int) -> Tuple[int, int, bool, bool]:
    lib = load()
    out = (C.c_uint8 * 40)()
    sb = (C.c_uint8 * 32).from_buffer_copy(s.to_bytes(32, "little"))
    lib.oracle_glv_decompose(sb, out)
    b = bytes(out)
    return (int.from_bytes(b[:16], "little"), int.from_bytes(b[16:32], "little"), bool(b[32]), bool(b[36]))


def fp_op(op: int, a: int, b: int = 0) -> int:
    lib = load()
    out = (C.c_uint8 * 48)()
    ab = (C.c_uint8 * 48).from_buffer_copy(a.to_bytes(48, "little"))
    bb = (C.c_uint8 * 48).from_buffer_copy(b.to_bytes(48, "little"))
    lib.oracle_fp_op(op, ab, bb, out)
    return int.from_bytes(bytes(out), "little")


def dot_mod(a, s, n: int, q: int) -> int:
    """sum_i a_i * s_i mod q over n pairs of 32-byte little-endian integers; `a`, `s`: bytes or ctypes arrays (not copied)."""
    lib = load()

    def ptr(x):
        return x if isinstance(x, C.Array) else (C.c_uint8 * max(len(x), 1)).from_buffer_copy(x or b"\0")

    out = (C.c_uint64 * 10)()
    lib.oracle_dot_u256(ptr(a), ptr(s), n, out)
    return sum(int(w) << (64 * i) for i, w in enumerate(out)) % q
