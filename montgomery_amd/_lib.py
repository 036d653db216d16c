"""ctypes binding of libmsm_hip.so (the C ABI in include/msm_hip.h).

There is no CPU fallback: if the HIP extension is missing or no GPU is usable, loading / creating
a context raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSM_HIP_LIB: A/B experiments with alternative builds of the same extension (tools/ab_time.py)
LIB_PATH = os.environ.get("MSM_HIP_LIB") or os.path.join(_HERE, "libmsm_hip.so")

MSM_OK, MSM_ERR_ARG, MSM_ERR_HIP, MSM_ERR_POINT, MSM_ERR_NO_POINTS, MSM_ERR_NO_DEVICE, MSM_ERR_SCALAR, MSM_ERR_INTERNAL = range(8)
CURVE_BLS12_377_G1 = 0
CURVE_ED_ON_BLS12_377 = 1
CURVE_BLS12_381_G1 = 2
CURVE_PALLAS = 3
ABI_VERSION = 6   # MSM_ABI_VERSION of the include/msm_hip.h this binding was written against
N_PHASES = 8
PHASE_NAMES = ("total", "upload", "digits", "sort", "accumulate", "reduce", "final", "accumulate_round1")

OP_MUL, OP_SQR, OP_ADD, OP_SUB, OP_INV, OP_TO_MONT, OP_FROM_MONT, OP_INV_FERMAT, OP_INV_KALISKI, OP_INV_WORDSLICED = range(10)

# every symbol include/msm_hip.h declares
EXPORTS = (
    "msm_ctx_create", "msm_ctx_destroy", "msm_last_error", "msm_set_points", "msm_run", "msm_window_sums",
    "msm_combine", "msm_combine_curve", "msm_plan", "msm_generate_points", "msm_generate_scalars", "msm_get_points", "msm_test_fp",
    "msm_test_glv", "msm_test_batch_add", "msm_test_batch_inverse",
    "msm_ctx_create_multi", "msm_ctx_device_count", "msm_pointset_create", "msm_pointset_select", "msm_pointset_destroy",
    "msm_device_alloc", "msm_device_free", "msm_device_upload",
    "msm_test_fp_raw", "msm_test_curve_op", "msm_test_batch_add_mode",
    "msm_run_placed", "msm_combine_groups", "msm_test_bucket_reduce", "msm_set_workspace_limit",
    "msm_precompute", "msm_tables_info", "msm_tables_range", "msm_set_tables_limit", "msm_reserve",
    "msm_abi_version", "msm_abi_struct_bytes",
)


class MsmOpts(C.Structure):
    _fields_ = [("c", C.c_int32), ("unsafe", C.c_int32), ("k_lo", C.c_int32), ("k_hi", C.c_int32), ("serial", C.c_int32),
                ("no_glv", C.c_int32), ("strict", C.c_int32), ("point_lo", C.c_uint32), ("by_window", C.c_int32), ("no_tables", C.c_int32),
                ("bucket_shard", C.c_int32), ("bucket_shards", C.c_int32), ("merged_sums", C.c_int32), ("reserved_", C.c_int32)]


class MsmResult(C.Structure):
    _fields_ = [
        ("x", C.c_uint8 * 48),
        ("y", C.c_uint8 * 48),
        ("is_infinity", C.c_int32),
        ("c", C.c_int32),
        ("K", C.c_int32),
        ("rounds", C.c_int32),
        ("phase_ms", C.c_float * N_PHASES),
        ("n_pairs", C.c_uint64),
        ("max_bucket", C.c_uint64),
        ("n_pairs_algo", C.c_uint64),
        ("tables", C.c_int32),
        ("reserved_", C.c_int32),
    ]


class MsmError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"msm error {code}: {message}")
        self.code = code


_lib = None


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process, whichever of torch and this package is imported first.

    The ROCm wheels of torch bundle their own libamdhip64.so (SONAME libamdhip64.so.7, found through the RPATH of torch's
    libraries under the name `libamdhip64.so`); libmsm_hip.so asks for `libamdhip64.so.7`.  With torch imported first the
    dynamic loader hands this library the runtime torch has already mapped (same SONAME).  The other way round it would map
    the system runtime for this library and then, for torch, the bundled file as a SECOND runtime -- and torch reports
    "No HIP GPUs are available".  So if a torch installation with a bundled runtime exists and is not loaded yet, that runtime
    is mapped here first: this library binds to it by SONAME, and a later `import torch` finds the very file already loaded.
    torch itself is never imported by this package."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        try:
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
        except OSError:
            pass   # the library then binds to the system runtime, as a process without torch does


def load() -> C.CDLL:
    """Load the HIP extension; raises if it has not been built (python __graft_entry__.py / make)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int32
    # a library built from another version of include/msm_hip.h keeps its symbol names but not its struct layouts
    if not hasattr(lib, "msm_abi_version"):
        raise ImportError(f"{LIB_PATH} predates msm_abi_version(): rebuild it (`make`)")
    lib.msm_abi_version.restype = C.c_uint32
    lib.msm_abi_struct_bytes.argtypes = [C.c_int]
    lib.msm_abi_struct_bytes.restype = C.c_uint32
    got = (lib.msm_abi_version(), lib.msm_abi_struct_bytes(0), lib.msm_abi_struct_bytes(1))
    want = (ABI_VERSION, C.sizeof(MsmOpts), C.sizeof(MsmResult))
    if got != want:
        raise ImportError(f"{LIB_PATH}: ABI (version, sizeof msm_opts, sizeof msm_result) = {got}, this binding expects {want}; "
                          "rebuild the library and the binding from the same include/msm_hip.h")
    lib.msm_ctx_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    lib.msm_ctx_create.restype = C.c_int
    lib.msm_ctx_destroy.argtypes = [vp]
    lib.msm_ctx_destroy.restype = None
    lib.msm_last_error.argtypes = [vp]
    lib.msm_last_error.restype = C.c_char_p
    lib.msm_set_points.argtypes = [vp, vp, u64, C.c_int, C.c_int]
    lib.msm_run.argtypes = [vp, vp, u64, C.c_int, C.POINTER(MsmOpts), C.POINTER(MsmResult)]
    lib.msm_window_sums.argtypes = [vp, vp, u64, C.c_int, C.POINTER(MsmOpts), vp, C.POINTER(MsmResult)]
    lib.msm_combine.argtypes = [vp, vp, i32, i32, C.POINTER(MsmResult)]
    lib.msm_combine_curve.argtypes = [i32, vp, i32, i32, C.POINTER(MsmResult)]
    lib.msm_combine_curve.restype = i32
    lib.msm_combine_groups.argtypes = [i32, vp, i32, i32, i32, C.POINTER(MsmResult)]
    lib.msm_run_placed.argtypes = [vp, C.POINTER(vp), u64, C.POINTER(MsmOpts), C.POINTER(MsmResult)]
    lib.msm_plan.argtypes = [vp, u64, C.POINTER(MsmOpts), C.POINTER(i32), C.POINTER(i32)]
    lib.msm_generate_points.argtypes = [vp, u64, u64, vp]
    lib.msm_generate_scalars.argtypes = [vp, u64, u64, vp, vp]
    lib.msm_get_points.argtypes = [vp, u64, u64, vp]
    lib.msm_test_fp.argtypes = [vp, C.c_int, vp, vp, vp, u64]
    lib.msm_test_glv.argtypes = [vp, vp, vp, u64]
    lib.msm_test_batch_add.argtypes = [vp, vp, vp, vp, u64]
    lib.msm_test_batch_inverse.argtypes = [vp, vp, vp, u64, C.c_uint32]
    lib.msm_ctx_create_multi.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(i32), i32]
    lib.msm_ctx_device_count.argtypes = [vp]
    lib.msm_pointset_create.argtypes = [vp, C.POINTER(i32)]
    lib.msm_pointset_select.argtypes = [vp, i32]
    lib.msm_pointset_destroy.argtypes = [vp, i32]
    lib.msm_device_alloc.argtypes = [vp, u64, C.POINTER(vp)]
    lib.msm_device_free.argtypes = [vp, vp]
    lib.msm_device_upload.argtypes = [vp, vp, vp, u64]
    lib.msm_set_workspace_limit.argtypes = [vp, u64]
    lib.msm_precompute.argtypes = [vp, u64, C.POINTER(MsmOpts)]
    lib.msm_tables_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(u64)]
    lib.msm_tables_range.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    lib.msm_set_tables_limit.argtypes = [vp, u64]
    lib.msm_reserve.argtypes = [vp, u64, C.POINTER(MsmOpts)]
    lib.msm_test_fp_raw.argtypes = [vp, C.c_int, vp, vp, vp, u64]
    lib.msm_test_curve_op.argtypes = [vp, C.c_int, vp, vp, vp, u64]
    lib.msm_test_batch_add_mode.argtypes = [vp, vp, vp, vp, u64, C.c_int, C.c_uint32]
    lib.msm_test_bucket_reduce.argtypes = [vp, vp, i32, C.c_uint32, C.c_int, C.c_int, vp, C.POINTER(C.c_float)]
    for name in EXPORTS:
        if name not in ("msm_ctx_destroy", "msm_last_error", "msm_abi_version", "msm_abi_struct_bytes"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib
