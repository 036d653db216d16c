"""Host-side mirror of the reference's curve-module API for the MSM path.

Reference surface being mirrored (names, argument meaning, error behaviour):
  * ``Weierstraß.create(params)``                      src/parallel.ts:40-177
  * ``Curve.Parallel.{msm, msmUnsafe, pointsFromBytes, scalarsFromBytes, getPointer,
    getScalarPointer, randomPointsFast, randomScalars}``   src/parallel.ts:135-145
  * ``msm(scalarPtr, pointPtr, N, verbose, {c, useSafeAdditions}) -> {result, log}``
                                                        src/msm-batched-affine.ts:69-78, :339
  * ``compute_msm(points, scalars) -> {x, y}``          scripts/zprize23/submission-bls377.ts:20-65

In the reference "pointers" are byte offsets into the shared wasm memory; here they are small handle
objects for buffers owned by the HIP library (points live in HBM in the library's row format, scalars
either on the host or in HBM).  All arithmetic happens in libmsm_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple, Union

from . import _lib
from ._lib import MsmError, MsmOpts, MsmResult

BytesLike = Union[bytes, bytearray, memoryview]


# ---------------------------------------------------------------------------------------------
# curve parameters (src/concrete/bls12-377.params.ts:11-45)
# ---------------------------------------------------------------------------------------------


@dataclass(frozen=True)
class WeierstrassParams:
    label: str
    modulus: int
    order: int
    cofactor: int
    a: int
    b: int
    generator: Tuple[int, int]
    endomorphism: Tuple[int, int]  # (lambda, beta)


BLS12_377_PARAMS = WeierstrassParams(
    label="bls12-377",
    modulus=0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001,
    order=0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001,
    cofactor=0x170B5D44300000000000000000000000,
    a=0,
    b=1,
    generator=(
        0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF,
        0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6,
    ),
    endomorphism=(
        0x12AB655E9A2CA55660B44D1E5C37B00114885F32400000000000000000000000,
        0x1AE3A4617C510EABC8756BA8F8C524EB8882A75CC9BC8E359064EE822FB5BFFD1E945779FFFFFFFFFFFFFFFFFFFFFFF,
    ),
)


BLS12_381_PARAMS = WeierstrassParams(  # src/concrete/bls12-381.params.ts:6-55
    label="bls12-381",
    modulus=0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
    order=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
    cofactor=0x396C8C005555E1568C00AAAB0000AAAB,
    a=0,
    b=4,
    generator=(
        0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
        0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    ),
    endomorphism=(
        0xD201000000010000 ** 2 - 1,
        0x1A0111EA397FE699EC02408663D4DE85AA0D857D89759AD4897D29650FB85F9B409427EB4F49FFFD8BFD00000000AAAC,
    ),
)

_PALLAS_P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
_PALLAS_Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
PALLAS_PARAMS = WeierstrassParams(  # src/concrete/pasta.params.ts:10-53
    label="pallas",
    modulus=_PALLAS_P,
    order=_PALLAS_Q,
    cofactor=1,
    a=0,
    b=5,
    generator=(1, 0x1B74B5A30A12937C53DFA9F06378EE548F655BD4333D477119CF7A23CAED2ABB),
    endomorphism=(pow(5, (_PALLAS_Q - 1) // 3, _PALLAS_Q), pow(pow(5, (_PALLAS_P - 1) // 3, _PALLAS_P), 2, _PALLAS_P)),
)

# curves with device constants (montgomery_amd/csrc/constants_gen.h), by label
_WEIERSTRASS_CURVE_IDS = {"bls12-377": _lib.CURVE_BLS12_377_G1, "bls12-381": _lib.CURVE_BLS12_381_G1,
                          "pallas": _lib.CURVE_PALLAS}


@dataclass(frozen=True)
class TwistedEdwardsParams:
    """src/concrete/ed-on-bls12-377.params.ts:5-31"""

    label: str
    modulus: int
    order: int
    cofactor: int
    d: int
    generator: Tuple[int, int]


ED_ON_BLS12_377_PARAMS = TwistedEdwardsParams(
    label="ed-on-bls12-377",
    modulus=0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001,
    order=0x4AAD957A68B2955982D1347970DEC005293A3AFC43C8AFEB95AEE9AC33FD9FF,
    cofactor=4,
    d=3021,
    generator=(
        0x9F1B5A5BAF6ACF06FED91C9AE9EBFA06068DD2835790980894E2328F3EBCA05,
        0x9A20DF36571AC3CD906B256080BA8454453C177AAF3131BB50A67BF1A806781,
    ),
)


# ---------------------------------------------------------------------------------------------
# low-level context
# ---------------------------------------------------------------------------------------------


@dataclass
class AffineResult:
    """Canonical affine result, as `Affine.toBigint` returns it (src/curve-affine.ts:220-233)."""

    x: int
    y: int
    isZero: bool

    def as_tuple(self) -> Optional[Tuple[int, int]]:
        return None if self.isZero else (self.x, self.y)


def _result_to_dict(res: MsmResult) -> Dict:
    return {
        "c": res.c,
        "K": res.K,
        "rounds": res.rounds,
        "n_pairs": int(res.n_pairs),
        "n_pairs_algo": int(res.n_pairs_algo),
        "max_bucket": int(res.max_bucket),
        "tables": bool(res.tables),
        "phase_ms": {name: float(res.phase_ms[i]) for i, name in enumerate(_lib.PHASE_NAMES)},
    }


class MsmContext:
    """One curve bound to one GPU: thin object wrapper over the msm_* C functions."""

    def __init__(self, curve: int = _lib.CURVE_BLS12_377_G1, device: int = 0, devices: Optional[Sequence[int]] = None):
        """device: one GPU.  devices: a list of GPUs of this node -- the context then shards every MSM by scalar window
        across them from host threads inside the library (msm_ctx_create_multi)."""
        self._lib = _lib.load()
        h = C.c_void_p()
        if devices is not None and len(devices) > 0:
            arr = (C.c_int32 * len(devices))(*devices)
            rc = self._lib.msm_ctx_create_multi(C.byref(h), curve, arr, len(devices))
            device = devices[0]
        else:
            rc = self._lib.msm_ctx_create(C.byref(h), curve, device)
        if rc != _lib.MSM_OK:
            raise MsmError(rc, "msm_ctx_create failed (no usable GPU?) -- there is no CPU fallback")
        self._h = h
        self.curve = curve
        self.device = device
        self.n_points = 0
        self._cur_set = 0
        self._set_sizes: Dict[int, int] = {0: 0}
        self.coord_bytes = 32 if curve in (_lib.CURVE_ED_ON_BLS12_377, _lib.CURVE_PALLAS) else 48   # per field, as the reference sizes them
        self._gen_buf, self._gen_cap = 0, 0   # device buffer of generate_scalars(into=0)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.msm_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc != _lib.MSM_OK:
            raise MsmError(rc, self._lib.msm_last_error(self._h).decode())

    # -- handles: point sets and device buffers ----------------------------------------------
    def pointset_create(self) -> int:
        """A new, empty resident point set; it becomes the current one (msm_pointset_create)."""
        i = C.c_int32()
        self._check(self._lib.msm_pointset_create(self._h, C.byref(i)))
        self.n_points = 0
        self._cur_set = i.value
        self._set_sizes[i.value] = 0
        return i.value

    def pointset_select(self, set_id: int) -> None:
        self._check(self._lib.msm_pointset_select(self._h, set_id))
        self._cur_set = set_id
        self.n_points = self._set_sizes.get(set_id, 0)

    def pointset_destroy(self, set_id: int) -> None:
        self._check(self._lib.msm_pointset_destroy(self._h, set_id))
        self._set_sizes.pop(set_id, None)
        if self._cur_set == set_id:
            self._cur_set = 0
            self.n_points = self._set_sizes.get(0, 0)

    def device_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self._lib.msm_device_alloc(self._h, nbytes, C.byref(p)))
        return int(p.value)

    def device_free(self, dev_ptr: int) -> None:
        self._check(self._lib.msm_device_free(self._h, C.c_void_p(dev_ptr)))

    def device_upload(self, dev_ptr: int, data: BytesLike) -> None:
        buf = data if isinstance(data, C.Array) else (C.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
        self._check(self._lib.msm_device_upload(self._h, C.c_void_p(dev_ptr), buf, len(data)))

    def set_workspace_limit(self, nbytes: int) -> None:
        """Device memory the working buffers of one call may take (0 = automatic); see include/msm_hip.h."""
        self._check(self._lib.msm_set_workspace_limit(self._h, int(nbytes)))

    @property
    def n_devices(self) -> int:
        return int(self._lib.msm_ctx_device_count(self._h))

    # -- points ---------------------------------------------------------------------------
    def set_points(self, points: BytesLike, check_curve: bool = False) -> int:
        step = 2 * self.coord_bytes
        if len(points) % step:
            raise MsmError(_lib.MSM_ERR_ARG, f"point buffer length {len(points)} is not a multiple of {step}")
        n = len(points) // step
        buf = (C.c_uint8 * max(len(points), 1)).from_buffer_copy(bytes(points) or b"\0")
        self._check(self._lib.msm_set_points(self._h, buf, n, 0, int(check_curve)))
        self.n_points = n
        self._set_sizes[self._cur_set] = n
        return n

    def set_points_device(self, dev_ptr: int, n: int, check_curve: bool = False) -> int:
        self._check(self._lib.msm_set_points(self._h, C.c_void_p(dev_ptr), n, 1, int(check_curve)))
        self.n_points = n
        self._set_sizes[self._cur_set] = n
        return n

    def generate_points(self, n: int, seed: int = 1, want_scalars: bool = False, raw: bool = False):
        """n resident points P_i = a_i G generated on the GPU.  want_scalars: also return the a_i (n x 32 bytes LE) --
        as `bytes`, or with raw=True as the ctypes array itself (no second 2 GB copy at 2^26)."""
        out = (C.c_uint8 * (32 * n))() if want_scalars and n else None
        self._check(self._lib.msm_generate_points(self._h, n, seed, out))
        self.n_points = n
        self._set_sizes[self._cur_set] = n
        if out is not None and raw:
            return out
        return bytes(out) if out is not None else (b"" if want_scalars else None)

    def generate_scalars(self, n: int, seed: int = 1, to_host: bool = False, into: int = 0, raw: bool = False):
        """n random scalars < q on the device.  `into`: caller-owned device pointer (n * 32 bytes); 0 = a device buffer this
        object allocates (msm_device_alloc) and keeps until the next such call or close().
        to_host: also return a host copy (bytes; the ctypes array itself with raw=True)."""
        if not into:
            if self._gen_buf and self._gen_cap < 32 * n:
                self.device_free(self._gen_buf)
                self._gen_buf = 0
            if not self._gen_buf:
                self._gen_buf, self._gen_cap = self.device_alloc(max(32 * n, 32)), max(32 * n, 32)
            into = self._gen_buf
        out = (C.c_uint8 * (32 * n))() if to_host and n else None
        self._check(self._lib.msm_generate_scalars(self._h, n, seed, C.c_void_p(into), out))
        if out is not None and raw:
            return int(into), out
        return int(into), (bytes(out) if out is not None else None)

    def get_points(self, first: int, count: int) -> bytes:
        step = 2 * self.coord_bytes
        out = (C.c_uint8 * max(step * count, 1))()
        self._check(self._lib.msm_get_points(self._h, first, count, out))
        return bytes(out)[: step * count]

    def get_point(self, i: int) -> Optional[Tuple[int, int]]:
        b = self.get_points(i, 1)
        nb = self.coord_bytes
        x, y = int.from_bytes(b[:nb], "little"), int.from_bytes(b[nb:], "little")
        return None if (x == 0 and y == 0) else (x, y)

    # -- msm ------------------------------------------------------------------------------
    def plan(self, n: int, c: Optional[int] = None, no_tables: bool = False, merged: bool = False, point_lo: int = 0) -> Tuple[int, int]:
        """(c, K) of msm_run over n points; over the whole resident point set that is the plan on window tables where they
        exist or would be built -- no_tables: the plain plan (what msm_window_sums without `merged` and bucket shards run);
        merged: the plan of window_sums(..., merged=True) over the points [point_lo, point_lo + n), which may run on the window
        tables of that range."""
        opts = MsmOpts(c=c or 0, no_tables=int(no_tables), merged_sums=int(merged), point_lo=point_lo)
        cc, kk = C.c_int32(), C.c_int32()
        self._check(self._lib.msm_plan(self._h, n, C.byref(opts), C.byref(cc), C.byref(kk)))
        return cc.value, kk.value

    def run(self, scalars: BytesLike, c: Optional[int] = None, unsafe: bool = False, no_glv: bool = False,
            by_window: bool = False, no_tables: bool = False) -> Tuple[AffineResult, Dict]:
        if len(scalars) % 32:
            raise MsmError(_lib.MSM_ERR_ARG, f"scalar buffer length {len(scalars)} is not a multiple of 32")
        n = len(scalars) // 32
        # a ctypes array is handed over as it is (no 2 GB copies at 2^26), anything else is copied once
        buf = scalars if isinstance(scalars, C.Array) else (C.c_uint8 * max(len(scalars), 1)).from_buffer_copy(bytes(scalars) or b"\0")
        return self._run(buf, n, 0, c, unsafe, no_glv=no_glv, by_window=by_window, no_tables=no_tables)

    def run_device(self, dev_ptr: int, n: int, c: Optional[int] = None, unsafe: bool = False, serial: bool = False,
                   no_glv: bool = False, by_window: bool = False, point_lo: int = 0, no_tables: bool = False) -> Tuple[AffineResult, Dict]:
        """by_window: a device-list context shards by scalar window instead of by points.  point_lo: the MSM covers the
        resident points [point_lo, point_lo + n) (scalar i belongs to point point_lo + i).  no_tables: the plain path even
        where window tables exist or would be built (msm_opts.no_tables)."""
        return self._run(C.c_void_p(dev_ptr), n, 1, c, unsafe, serial, no_glv, by_window, point_lo, no_tables)

    def run_placed(self, dev_ptrs: Sequence[int], n: int, c: Optional[int] = None) -> Tuple[AffineResult, Dict]:
        """Device-list context, scalars already placed: dev_ptrs[d] on devices[d] holds the scalars of that device's share
        [n d / G, n (d + 1) / G) of the points (msm_run_placed)."""
        arr = (C.c_void_p * len(dev_ptrs))(*[C.c_void_p(int(p)) for p in dev_ptrs])
        opts, res = MsmOpts(c=c or 0), MsmResult()
        self._check(self._lib.msm_run_placed(self._h, arr, n, C.byref(opts), C.byref(res)))
        nb = self.coord_bytes
        out = AffineResult(int.from_bytes(bytes(res.x)[:nb], "little"), int.from_bytes(bytes(res.y)[:nb], "little"),
                           bool(res.is_infinity))
        return out, _result_to_dict(res)

    def _run(self, ptr, n: int, on_device: int, c: Optional[int], unsafe: bool, serial: bool = False,
             no_glv: bool = False, by_window: bool = False, point_lo: int = 0, no_tables: bool = False) -> Tuple[AffineResult, Dict]:
        opts = MsmOpts(c=c or 0, unsafe=int(unsafe), serial=int(serial), no_glv=int(no_glv), by_window=int(by_window),
                       point_lo=point_lo, no_tables=int(no_tables))
        res = MsmResult()
        self._check(self._lib.msm_run(self._h, ptr, n, on_device, C.byref(opts), C.byref(res)))
        nb = self.coord_bytes
        out = AffineResult(
            x=int.from_bytes(bytes(res.x)[:nb], "little"),
            y=int.from_bytes(bytes(res.y)[:nb], "little"),
            isZero=bool(res.is_infinity),
        )
        return out, _result_to_dict(res)

    # -- window tables (msm_precompute, include/msm_hip.h) ------------------------------------
    def precompute(self, n: Optional[int] = None, c: Optional[int] = None, no_glv: bool = False, point_lo: int = 0) -> Tuple[int, int, int]:
        """Builds the window tables of the current point set -- of its points [point_lo, point_lo + n): the share of one rank of a
        points-split run -- for the plan msm_run(n, c) would use (no-op if present or if they do not fit the limit).
        Returns tables_info()."""
        opts = MsmOpts(c=c or 0, no_glv=int(no_glv), point_lo=point_lo)
        self._check(self._lib.msm_precompute(self._h, self.n_points if n is None else n, C.byref(opts)))
        return self.tables_info()

    def tables_info(self) -> Tuple[int, int, int]:
        """(window bits, number of tables, bytes) of the current point set's window tables; (0, 0, 0): none."""
        c, k, b = C.c_int32(), C.c_int32(), C.c_uint64()
        self._check(self._lib.msm_tables_info(self._h, C.byref(c), C.byref(k), C.byref(b)))
        return c.value, k.value, b.value

    def tables_range(self) -> Tuple[int, int]:
        """(first point, number of points) the current point set's window tables cover; (0, 0): none."""
        lo, n = C.c_uint64(), C.c_uint64()
        self._check(self._lib.msm_tables_range(self._h, C.byref(lo), C.byref(n)))
        return lo.value, n.value

    def set_tables_limit(self, nbytes: int) -> None:
        self._check(self._lib.msm_set_tables_limit(self._h, nbytes))

    def reserve(self, n: int, c: Optional[int] = None) -> None:
        """Everything a later run_device(n, c) would allocate or build, now (msm_reserve)."""
        opts = MsmOpts(c=c or 0)
        self._check(self._lib.msm_reserve(self._h, n, C.byref(opts)))

    def window_sums(self, scalars: Union[BytesLike, int], n: int, k_lo: int, k_hi: int, c: Optional[int] = None,
                    on_device: bool = False, point_lo: int = 0, by_window: bool = False,
                    bucket_shard: Tuple[int, int] = (0, 0), merged: bool = False) -> Tuple[bytes, Dict]:
        """Partition sums P_k, k in [k_lo, k_hi), over the resident points [point_lo, point_lo + n) (scalar i belongs to
        point point_lo + i): (k_hi - k_lo) x 144 bytes (X, Y, Z).  merged (msm_opts.merged_sums): the caller only combines the
        sums, so they may come back merged -- the first slot carries sum_k 2^(c (k - k_lo)) P_k, the others the identity -- and
        the call may run on window tables (of the whole set, or of exactly this range of the points)."""
        if k_hi <= k_lo or k_lo < 0:   # (0, 0) would mean "all windows" to the C side and overrun the 144-byte buffer below
            raise MsmError(_lib.MSM_ERR_ARG, f"empty or negative window range [{k_lo}, {k_hi})")
        # bucket_shard = (g, G): only the buckets [L g / G, L (g + 1) / G) of every window (msm_opts.bucket_shard)
        opts = MsmOpts(c=c or 0, k_lo=k_lo, k_hi=k_hi, point_lo=point_lo, by_window=int(by_window),
                       bucket_shard=bucket_shard[0], bucket_shards=bucket_shard[1], merged_sums=int(merged))
        res = MsmResult()
        out = (C.c_uint8 * (144 * max(k_hi - k_lo, 1)))()
        if on_device:
            ptr = C.c_void_p(int(scalars))
        elif isinstance(scalars, C.Array):
            ptr = scalars   # handed over as it is (no copy of a 2 GB buffer)
        else:
            ptr = (C.c_uint8 * max(32 * n, 1)).from_buffer_copy(bytes(scalars) or b"\0")
        self._check(self._lib.msm_window_sums(self._h, ptr, n, int(on_device), C.byref(opts), out, C.byref(res)))
        return bytes(out)[: 144 * (k_hi - k_lo)], _result_to_dict(res)

    def combine(self, partials: BytesLike, K: int, c: int) -> AffineResult:
        buf = (C.c_uint8 * len(partials)).from_buffer_copy(bytes(partials))
        res = MsmResult()
        self._check(self._lib.msm_combine(self._h, buf, K, c, C.byref(res)))
        nb = self.coord_bytes
        return AffineResult(int.from_bytes(bytes(res.x)[:nb], "little"), int.from_bytes(bytes(res.y)[:nb], "little"), bool(res.is_infinity))

    # -- fine-grained operators (GPU test kernels) -------------------------------------------
    def test_fp(self, op: int, a: BytesLike, b: Optional[BytesLike] = None) -> bytes:
        nb = self.coord_bytes
        n = len(a) // nb
        b = a if b is None else b
        ba = (C.c_uint8 * len(a)).from_buffer_copy(bytes(a))
        bb = (C.c_uint8 * len(b)).from_buffer_copy(bytes(b))
        out = (C.c_uint8 * len(a))()
        self._check(self._lib.msm_test_fp(self._h, op, ba, bb, out, n))
        return bytes(out)

    def test_glv(self, scalars: BytesLike) -> List[Tuple[int, int, bool, bool]]:
        n = len(scalars) // 32
        bs = (C.c_uint8 * len(scalars)).from_buffer_copy(bytes(scalars))
        out = (C.c_uint8 * (40 * n))()
        self._check(self._lib.msm_test_glv(self._h, bs, out, n))
        raw = bytes(out)
        res = []
        for i in range(n):
            r = raw[40 * i : 40 * i + 40]
            res.append((int.from_bytes(r[:16], "little"), int.from_bytes(r[16:32], "little"),
                        bool(int.from_bytes(r[32:36], "little")), bool(int.from_bytes(r[36:40], "little"))))
        return res

    def test_batch_inverse(self, xs: BytesLike, per_lane: int = 7) -> bytes:
        n = len(xs) // self.coord_bytes
        bx = (C.c_uint8 * len(xs)).from_buffer_copy(bytes(xs))
        out = (C.c_uint8 * len(xs))()
        self._check(self._lib.msm_test_batch_inverse(self._h, bx, out, n, per_lane))
        return bytes(out)

    def test_fp_raw(self, op: int, a_limbs: Sequence[Sequence[int]], b_limbs: Sequence[Sequence[int]]) -> List[List[int]]:
        """fe_mul / fe_sqr on raw 30-bit-limb operands (lists of NL ints per element); returns the raw result limbs."""
        nl = 9 if self.curve in (_lib.CURVE_ED_ON_BLS12_377, _lib.CURVE_PALLAS) else 13   # limbs are sized per field
        n = len(a_limbs)
        A = (C.c_uint32 * (nl * n))(*[w for e in a_limbs for w in e])
        B = (C.c_uint32 * (nl * n))(*[w for e in b_limbs for w in e])
        out = (C.c_uint32 * (nl * n))()
        self._check(self._lib.msm_test_fp_raw(self._h, op, A, B, out, n))
        return [[int(out[i * nl + j]) for j in range(nl)] for i in range(n)]

    def test_curve_op(self, op: int, p: BytesLike, q: BytesLike) -> bytes:
        """Projective (X || Y || Z, one coordinate width each) or extended Edwards (X || Y || Z || T, 32 B each) operator; see
        msm_hip.h."""
        bp = (C.c_uint8 * len(p)).from_buffer_copy(bytes(p))
        bq = (C.c_uint8 * len(q)).from_buffer_copy(bytes(q))
        out = (C.c_uint8 * len(p))()
        nb = 128 if self.curve == _lib.CURVE_ED_ON_BLS12_377 else 3 * self.coord_bytes
        self._check(self._lib.msm_test_curve_op(self._h, op, bp, bq, out, len(p) // nb))
        return bytes(out)

    def test_batch_add_mode(self, g: BytesLike, h: BytesLike, mode: int, steps: int) -> bytes:
        n = len(g) // (2 * self.coord_bytes)
        bg = (C.c_uint8 * len(g)).from_buffer_copy(bytes(g))
        bh = (C.c_uint8 * len(h)).from_buffer_copy(bytes(h))
        out = (C.c_uint8 * len(g))()
        self._check(self._lib.msm_test_batch_add_mode(self._h, bg, bh, out, n, mode, steps))
        return bytes(out)

    def test_bucket_reduce(self, buckets: BytesLike, K: int, L: int, mode: int = 0, c0: int = 2) -> Tuple[bytes, float]:
        """P_k = sum_l l B_(k,l) for K windows of L buckets (x || y, 48-byte LE each, (0, 0) = empty): K x 144 bytes (X, Y, Z)
        and the device time in ms.  mode 0: the projective reduction of the MSM; mode 1: the reference's all-affine
        reduction (reduceBucketsAffine) out of in-place batched additions, chunks of 2^c0 buckets."""
        if len(buckets) != 2 * self.coord_bytes * K * L:
            raise MsmError(_lib.MSM_ERR_ARG, f"expected {2 * self.coord_bytes * K * L} bytes of buckets, got {len(buckets)}")
        buf = (C.c_uint8 * len(buckets)).from_buffer_copy(bytes(buckets))
        out = (C.c_uint8 * (144 * K))()
        ms = C.c_float(0)
        self._check(self._lib.msm_test_bucket_reduce(self._h, buf, K, L, mode, c0, out, C.byref(ms)))
        return bytes(out), float(ms.value)

    def test_batch_add(self, g: BytesLike, h: BytesLike) -> bytes:
        n = len(g) // (2 * self.coord_bytes)
        bg = (C.c_uint8 * len(g)).from_buffer_copy(bytes(g))
        bh = (C.c_uint8 * len(h)).from_buffer_copy(bytes(h))
        out = (C.c_uint8 * len(g))()
        self._check(self._lib.msm_test_batch_add(self._h, bg, bh, out, n))
        return bytes(out)


# ---------------------------------------------------------------------------------------------
# reference-shaped facade
# ---------------------------------------------------------------------------------------------


class PointPtr:
    """Handle standing in for the reference's `pointPtr` (byte offset of an affine point array): one resident point set
    of the context, its own allocation like every pointer of the reference; freed when the handle goes away."""

    def __init__(self, ctx: Optional["MsmContext"] = None, size: int = 0, n: int = 0, set_id: int = 0):
        self._ctx, self.size, self.n, self.set_id = ctx, size, n, set_id

    def close(self) -> None:
        ctx, self._ctx = self._ctx, None
        if ctx is not None and self.set_id > 0 and getattr(ctx, "_h", None):
            ctx.pointset_destroy(self.set_id)

    def __iter__(self):
        """`[pointPtr] = Parallel.randomPointsFast(N)`: the reference returns one pointer per point and its callers keep the first
        (scripts/msm-weierstrass.ts:18); a handle here stands for the whole array and unpacks to itself."""
        yield self

    def __enter__(self) -> "PointPtr":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ScalarPtr:
    """Handle standing in for `scalarPtr`: host bytes (`scalarsFromBytes`) or a device buffer of its own
    (`randomScalars`); the device buffer is freed when the handle goes away."""

    def __init__(self, ctx: Optional["MsmContext"] = None, size: int = 0, data: bytes = b"", dev_ptr: int = 0, n: int = 0):
        self._ctx, self.size, self.data, self.dev_ptr, self.n = ctx, size, data, dev_ptr, n

    def close(self) -> None:
        ctx, self._ctx = self._ctx, None
        if ctx is not None and self.dev_ptr and getattr(ctx, "_h", None):
            ctx.device_free(self.dev_ptr)
        self.dev_ptr = 0

    def __iter__(self):
        """`[scalarPtr] = Parallel.randomScalars(N)` (scripts/msm-weierstrass.ts:21,29): unpacks to itself."""
        yield self

    def __enter__(self) -> "ScalarPtr":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Parallel:
    """`Curve.Parallel` (src/parallel.ts:135-145 Weierstrass, :251-259 twisted Edwards)."""

    def __init__(self, ctx: MsmContext, params):
        self._ctx = ctx
        self._params = params
        # wire bytes per coordinate = the reference's packed field size (src/wasm/field-helpers.ts:211-301) = the C ABI's
        self._wire_bytes = (params.modulus.bit_length() + 7) // 8 if hasattr(params, "modulus") else ctx.coord_bytes

    def getPointer(self, size: int) -> PointPtr:
        return PointPtr(self._ctx, size=size, set_id=self._ctx.pointset_create())

    def getScalarPointer(self, size: int) -> ScalarPtr:
        return ScalarPtr(self._ctx, size=size)

    def pointsFromBytes(self, pointPtr: PointPtr, pointInput: BytesLike, n: int) -> None:
        """src/parallel.ts:97-116 (96 B/point) / :215-229 (64 B/point): x || y little-endian -> resident device points."""
        wb, cb = self._wire_bytes, self._ctx.coord_bytes
        buf = bytes(pointInput)[: 2 * wb * n]
        if wb != cb:   # zero-pad every coordinate to the ABI width
            import numpy as np

            padded = np.zeros((2 * n, cb), dtype=np.uint8)
            padded[:, :wb] = np.frombuffer(buf, dtype=np.uint8).reshape(2 * n, wb)
            buf = padded.tobytes()
        self._ctx.pointset_select(pointPtr.set_id)
        self._ctx.set_points(buf)
        pointPtr.n = n

    def scalarsFromBytes(self, scalarPtr: ScalarPtr, scalarInput: BytesLike, n: int) -> None:
        """src/parallel.ts:119-133: n scalars of 32 bytes little-endian."""
        scalarPtr.data = bytes(scalarInput)[: 32 * n]
        scalarPtr.dev_ptr = 0
        scalarPtr.n = n

    def randomPointsFast(self, n: int, seed: int = 1) -> PointPtr:
        """src/curve-random.ts:14-92 (generated on the GPU; the seed is explicit, the reference is unseeded)."""
        ptr = self.getPointer(2 * self._wire_bytes * n)
        self._ctx.generate_points(n, seed)
        ptr.n = n
        return ptr

    def randomScalars(self, n: int, seed: int = 1) -> ScalarPtr:
        """src/curve-random.ts:151-194."""
        dev = self._ctx.device_alloc(32 * max(n, 1))   # the handle's own buffer: two handles never alias
        self._ctx.generate_scalars(n, seed, into=dev)
        return ScalarPtr(self._ctx, size=32 * n, dev_ptr=dev, n=n)

    def msm(self, scalarPtr: ScalarPtr, pointPtr: PointPtr, N: int, verboseTiming: bool = False,
            options: Optional[Dict] = None) -> Dict:
        """`msm` (src/msm-batched-affine.ts:69-340): returns {"result": AffineResult, "log": [...]}.
        options: {"c": window bits, "useSafeAdditions": bool}."""
        options = options or {}
        self._ctx.pointset_select(pointPtr.set_id)
        if N > pointPtr.n or N > self._ctx.n_points:
            raise MsmError(_lib.MSM_ERR_NO_POINTS, f"{N} scalars but {min(pointPtr.n, self._ctx.n_points)} points behind this pointer")
        if N > scalarPtr.n:
            raise MsmError(_lib.MSM_ERR_ARG, f"{N} scalars requested but the scalar pointer holds {scalarPtr.n}")
        unsafe = not options.get("useSafeAdditions", True)
        no_glv = bool(options.get("noGlv", False))
        if scalarPtr.dev_ptr:
            res, info = self._ctx.run_device(scalarPtr.dev_ptr, N, options.get("c"), unsafe, no_glv=no_glv)
        else:
            res, info = self._ctx.run(scalarPtr.data[: 32 * N], options.get("c"), unsafe, no_glv=no_glv)
        log: List = []
        if verboseTiming:
            # the reference's shape (createLog, src/msm-common.ts:176-214, filled at src/msm-batched-affine.ts:79-338): the
            # parameters first, one "label... x.xms" line per phase under the reference's labels, "msm total" last
            t = info["phase_ms"]
            log.append([{"n": (N - 1).bit_length() if N > 1 else 0, "K": info["K"], "c": info["c"]}])
            for label, key in (("scalars to device", "upload"), ("slice scalars & count buckets", "digits"), ("sort points", "sort"),
                               ("bucket accumulation (first round)", "accumulate_round1"), ("bucket accumulation", "accumulate"),
                               ("bucket reduction (local)", "reduce"), ("final sum", "final"), ("msm total", "total")):
                log.append([f"{label}... {t[key]:.1f}ms"])
        return {"result": res, "log": log, "info": info}

    def msmProjective(self, scalarPtr: ScalarPtr, pointPtr: PointPtr, N: int, options: Optional[Dict] = None) -> Dict:
        """`msmProjective` (src/parallel.ts:69-87: msmBasic over projective points): signed windows of the whole scalar,
        no endomorphism split, K = ceil((b + 1) / c) with b = bit length of q (src/msm-basic.ts:56-59).  The bucket sums
        still come from the batched-affine tree; the value is the same group element either way."""
        options = dict(options or {})
        options["noGlv"] = True
        return self.msm(scalarPtr, pointPtr, N, False, options)

    def msmUnsafe(self, scalarPtr: ScalarPtr, pointPtr: PointPtr, N: int, verboseTiming: bool = False,
                  options: Optional[Dict] = None) -> Dict:
        """`msmUnsafe` (src/msm-batched-affine.ts:587-598). The GPU kernels always handle the edge cases."""
        options = dict(options or {})
        options["useSafeAdditions"] = False
        return self.msm(scalarPtr, pointPtr, N, verboseTiming, options)


class _ValuePtr:
    """Stand-in for a wasm pointer of the reference (`Field.getPointer(size)`): holds the value written through it."""

    def __init__(self, size: int = 0):
        self.size, self.value = size, None


class _FieldShim:
    """`Curve.Field.getPointer / getPointers` as the reference's callers use them around an MSM."""

    @staticmethod
    def getPointer(size: int = 0) -> _ValuePtr:
        return _ValuePtr(size)

    @staticmethod
    def getPointers(n: int, size: int = 0) -> List[_ValuePtr]:
        return [_ValuePtr(size) for _ in range(n)]


class _AffineShim:
    """`Curve.Affine.toBigint(ptr)` (src/curve-affine.ts:220-233) -> {"x", "y", "isZero"}; the value behind the pointer is
    already canonical affine here."""

    def __init__(self, coord_bytes: int):
        self.size = 2 * coord_bytes + 4

    @staticmethod
    def toBigint(ptr) -> Dict:
        r = ptr.value if isinstance(ptr, _ValuePtr) else ptr
        return {"x": r.x, "y": r.y, "isZero": bool(r.isZero)}


class _ProjectiveShim:
    """`Curve.Projective.toAffine(scratch, affinePtr, result)` (src/curve-projective.ts:335-349): the reference's callers pass
    the `result` of `Parallel.msm` through it (scripts/msm-weierstrass.ts:89-91).  The library has normalised the sum already,
    so this only stores it behind the pointer."""

    def __init__(self, coord_bytes: int):
        self.size = 3 * coord_bytes + 4

    @staticmethod
    def toAffine(_scratch, affinePtr: _ValuePtr, result: AffineResult) -> None:
        affinePtr.value = result


class _TeCurveShim:
    """`Curve.Curve.toBigint(result)` of the twisted-Edwards module -> extended point {"X", "Y", "Z", "T"}
    (scripts/msm-twisted-edwards.ts:87, scripts/zprize23/submission.ts:33)."""

    def __init__(self, p: int):
        self.p = p

    def toBigint(self, result) -> Dict:
        r = result.value if isinstance(result, _ValuePtr) else result
        return {"X": r.x, "Y": r.y, "Z": 1, "T": r.x * r.y % self.p}


class _TeBigintShim:
    """`Curve.Bigint.toAffine(P)` (src/bigint/twisted-edwards.ts): {"X", "Y", "Z", ...} -> {"x", "y"}."""

    def __init__(self, p: int):
        self.p = p

    def toAffine(self, P: Dict) -> Dict:
        zi = pow(P["Z"], -1, self.p)
        return {"x": P["X"] * zi % self.p, "y": P["Y"] * zi % self.p}


class Weierstrass:
    """Curve module as `Weierstraß.create(params)` returns it (src/parallel.ts:147-160), MSM path only."""

    def __init__(self, params: WeierstrassParams, device: int = 0, devices: Optional[Sequence[int]] = None):
        """`devices`: a device list instead of one device -- every device holds the whole point set and `Parallel.msm` runs
        all windows on each device's share of the points (by points, the default; `by_window=True` on the context's run calls
        shards by scalar window instead; msm_ctx_create_multi); the counterpart of the reference's thread count,
        src/parallel.ts:40-66."""
        if params.label not in _WEIERSTRASS_CURVE_IDS:
            raise MsmError(_lib.MSM_ERR_ARG, f"curve {params.label!r} has no device constants "
                                             f"(have {sorted(_WEIERSTRASS_CURVE_IDS)})")
        self.params = params
        self.context = MsmContext(_WEIERSTRASS_CURVE_IDS[params.label], device, devices=devices)
        self.Parallel = _Parallel(self.context, params)
        self.Field = _FieldShim()
        self.Affine = _AffineShim(self.context.coord_bytes)
        self.Projective = _ProjectiveShim(self.context.coord_bytes)

    @classmethod
    def create(cls, params: WeierstrassParams, device: int = 0, devices: Optional[Sequence[int]] = None) -> "Weierstrass":
        return cls(params, device, devices)


def create_weierstrass(params: WeierstrassParams = BLS12_377_PARAMS, device: int = 0,
                       devices: Optional[Sequence[int]] = None) -> Weierstrass:
    return Weierstrass.create(params, device, devices)


class TwistedEdwards:
    """Curve module as `TwistedEdwards.create(params)` returns it (src/parallel.ts:179-289), MSM path only:
    `Parallel.msm` is `msmBasic` (src/msm-basic.ts:45-164) on extended points."""

    def __init__(self, params: TwistedEdwardsParams, device: int = 0, devices: Optional[Sequence[int]] = None):
        if params.label != "ed-on-bls12-377":
            raise MsmError(_lib.MSM_ERR_ARG, f"curve {params.label!r} has no device constants (only ed-on-bls12-377)")
        self.params = params
        self.context = MsmContext(_lib.CURVE_ED_ON_BLS12_377, device, devices=devices)
        self.Parallel = _Parallel(self.context, params)
        self.Field = _FieldShim()
        self.Curve = _TeCurveShim(params.modulus)
        self.Bigint = _TeBigintShim(params.modulus)

    @classmethod
    def create(cls, params: TwistedEdwardsParams, device: int = 0, devices: Optional[Sequence[int]] = None) -> "TwistedEdwards":
        return cls(params, device, devices)


class _LazyCurve:
    """`BLS12377` of src/concrete/bls12-377.ts: created on first use (needs a GPU)."""

    def __init__(self, params: WeierstrassParams):
        self._params = params
        self._curve: Optional[Weierstrass] = None

    def _get(self) -> Weierstrass:
        if self._curve is None:
            self._curve = Weierstrass.create(self._params)
        return self._curve

    def __getattr__(self, name):
        return getattr(self._get(), name)


Weierstraß = Weierstrass   # the reference's spelling, src/parallel.ts:40


def startThreads(n: Optional[int] = None) -> None:
    """`startThreads(n)` of src/parallel.ts:291-309, which the reference's callers run before any MSM
    (scripts/msm-weierstrass.ts:14, src/msm.test.ts:23).  The worker pool it starts is replaced by the GPU grid: nothing to do."""


def stopThreads() -> None:
    """`stopThreads()` of src/parallel.ts:317-320: nothing to stop (contexts are closed through their curve objects)."""


BLS12377 = _LazyCurve(BLS12_377_PARAMS)
BLS12381 = _LazyCurve(BLS12_381_PARAMS)  # src/concrete/bls12-381.ts


def compute_msm_ed(inputPoints, inputScalars, curve: Optional[TwistedEdwards] = None) -> Dict[str, int]:
    """ZPrize entry point for the twisted Edwards curve, scripts/zprize23/submission.ts:19-60.

    inputPoints: bytes (n x 64, x || y little-endian) or a list of {"x", "y", ...} dicts (z = 1 assumed);
    inputScalars: bytes (n x 32) or a list of ints.  Returns {"x": int, "y": int} (the identity is (0, 1))."""
    cv = curve or TwistedEdwards.create(ED_ON_BLS12_377_PARAMS)
    if isinstance(inputScalars, (bytes, bytearray, memoryview)):
        sbytes = bytes(inputScalars)
    else:
        sbytes = b"".join(int(s).to_bytes(32, "little") for s in inputScalars)
    n = len(sbytes) // 32
    if isinstance(inputPoints, (bytes, bytearray, memoryview)):
        pbytes = bytes(inputPoints)
    else:
        pbytes = b"".join(int(P["x"]).to_bytes(32, "little") + int(P["y"]).to_bytes(32, "little") for P in inputPoints)
    par = cv.Parallel
    with par.getPointer(len(pbytes)) as pp, par.getScalarPointer(len(sbytes)) as sp:   # freed on every path, errors included
        par.pointsFromBytes(pp, pbytes, n)
        par.scalarsFromBytes(sp, sbytes, n)
        res = par.msm(sp, pp, n)["result"]
    return {"x": res.x, "y": res.y}


def compute_msm(inputPoints, inputScalars, curve: Optional[Weierstrass] = None) -> Dict[str, int]:
    """ZPrize entry point, scripts/zprize23/submission-bls377.ts:20-65.

    inputPoints: bytes (n x 96, x || y little-endian) or a list of {"x", "y", "isZero"} dicts;
    inputScalars: bytes (n x 32 little-endian) or a list of ints.  Returns {"x": int, "y": int}.
    """
    cv = curve or BLS12377._get()
    if isinstance(inputScalars, (bytes, bytearray, memoryview)):
        sbytes = bytes(inputScalars)
    else:
        sbytes = b"".join(int(s).to_bytes(32, "little") for s in inputScalars)
    n = len(sbytes) // 32
    if isinstance(inputPoints, (bytes, bytearray, memoryview)):
        pbytes = bytes(inputPoints)
    else:
        chunks = []
        for P in inputPoints:
            if P.get("isZero"):
                chunks.append(b"\0" * 96)
            else:
                chunks.append(int(P["x"]).to_bytes(48, "little") + int(P["y"]).to_bytes(48, "little"))
        pbytes = b"".join(chunks)
    par = cv.Parallel
    with par.getPointer(len(pbytes)) as pp, par.getScalarPointer(len(sbytes)) as sp:   # freed on every path, errors included
        par.pointsFromBytes(pp, pbytes, n)
        par.scalarsFromBytes(sp, sbytes, n)
        same = n > 1 and pbytes[:96] == pbytes[96:192]
        out = par.msm(sp, pp, n) if same else par.msmUnsafe(sp, pp, n)
        res = out["result"]
    return {"x": res.x, "y": res.y}
