"""The C-ABI library loads and exports every symbol include/msm_hip.h declares (no compute: no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "msm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msm_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from montgomery_amd import _lib

    names = declared_functions()
    assert len(names) >= 10
    assert sorted(_lib.EXPORTS) == names, "montgomery_amd/_lib.py EXPORTS is out of sync with include/msm_hip.h"
    assert os.path.exists(_lib.LIB_PATH), "HIP extension not built: run `python -c 'import __graft_entry__ as g; g.build()'`"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"libmsm_hip.so does not export {n}"


def test_result_struct_layout_matches_header():
    from montgomery_amd._lib import MsmOpts, MsmResult

    # msm_opts: 14 x 32-bit fields (12 until ABI 5; merged_sums + one reserved word since 6); msm_result: 96 B + 4 x int32 + 8 floats + 3 x u64 + 2 x int32
    assert ctypes.sizeof(MsmOpts) == 56
    text = open(os.path.join(ROOT, "include", "msm_hip.h")).read()
    body = re.sub(r"/\*.*?\*/", "", text[text.index("typedef struct msm_opts {"):text.index("} msm_opts;")], flags=re.S)
    fields = re.findall(r"u?int32_t\s+([a-z_0-9, ]+);", body)
    names = [n.strip() for f in fields for n in f.split(",")]
    assert names == [f[0] for f in MsmOpts._fields_], (names, MsmOpts._fields_)
    assert ctypes.sizeof(MsmResult) == 96 + 16 + 32 + 24 + 8


def test_abi_version_is_checked_at_load():
    """The library reports the header version and struct sizes it was built with (no GPU needed); the Python loader compares
    them with its own and refuses a library from another version of include/msm_hip.h instead of misreading its structs."""
    from montgomery_amd import _lib

    text = open(os.path.join(ROOT, "include", "msm_hip.h")).read()
    ver = int(re.search(r"#define\s+MSM_ABI_VERSION\s+(\d+)", text).group(1))
    assert ver == _lib.ABI_VERSION
    lib = _lib.load()
    assert lib.msm_abi_version() == ver
    assert lib.msm_abi_struct_bytes(0) == ctypes.sizeof(_lib.MsmOpts)
    assert lib.msm_abi_struct_bytes(1) == ctypes.sizeof(_lib.MsmResult)
    # a binding written against another version must be refused
    saved = (_lib._lib, _lib.ABI_VERSION)
    try:
        _lib._lib, _lib.ABI_VERSION = None, ver + 1
        try:
            _lib.load()
        except ImportError as e:
            assert "ABI" in str(e)
        else:
            raise AssertionError("a library of another ABI version was accepted")
    finally:
        _lib._lib, _lib.ABI_VERSION = saved


def test_no_cpu_fallback_without_gpu():
    """Creating a context must fail loudly when no GPU is usable (this container has none)."""
    import torch

    from montgomery_amd import MsmError
    from montgomery_amd.api import MsmContext

    if torch.cuda.is_available():
        return
    try:
        MsmContext()
    except MsmError as e:
        assert e.code in (2, 5)
    else:
        raise AssertionError("MsmContext() succeeded without a GPU: a fallback path exists")


def test_product_does_not_import_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "montgomery_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                for line in src.splitlines():
                    s = line.strip()
                    if s.startswith(("import ", "from ", "#include")):
                        assert "oracle" not in s, f"{f}: {s}"


def test_plain_c_host_links_and_fails_loudly_without_gpu():
    """examples/msm_demo.c links against the C ABI alone; on a box without a GPU it must stop with an error
    (there is no CPU fallback), on a GPU box tests/test_napi.py::test_plain_c_host_of_the_abi runs it for real."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from conftest import build_if_missing

    build_if_missing("all", "montgomery_amd/libmsm_hip.so")
    build_if_missing("demo", "examples/msm_demo")
    out = subprocess.run([os.path.join(root, "examples", "msm_demo"), "10"], capture_output=True, text=True, timeout=120)
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        assert out.returncode == 0, out.stdout + out.stderr
    else:
        assert out.returncode == 1 and "no CPU fallback" in out.stderr, out.stdout + out.stderr
