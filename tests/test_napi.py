"""Node N-API shim + JS facade: builds, loads in the image's node, fails loudly
without a GPU; on a GPU box it runs the reference's ZPrize self-tests and the golden vectors through JS."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")


@pytest.fixture(scope="module")
def addon():
    if NODE is None or not os.path.exists("/usr/include/node/node_api.h"):
        pytest.skip("node / node_api.h not present")
    from conftest import build_if_missing

    build_if_missing("all", "montgomery_amd/libmsm_hip.so")
    build_if_missing("napi", "montgomery_amd/msm_hip.node")
    return os.path.join(ROOT, "montgomery_amd", "msm_hip.node")


def test_addon_loads_and_exports(addon):
    out = subprocess.run([NODE, "-e", "const m=require('./js/montgomery-hip.js');console.log(Object.keys(m.hip).sort().join(','))"],
                         cwd=ROOT, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    names = out.stdout.strip().split(",")
    for n in ("createContext", "destroyContext", "setPoints", "msm", "plan", "deviceAlloc", "deviceUpload", "deviceFree", "msmDevice",
              "fieldOp", "batchInverse", "glvDecompose", "batchAdd", "OP_MUL", "OP_INV", "ABI_VERSION"):
        assert n in names


def test_addon_fails_loudly_without_gpu(addon):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    code = ("const m=require('./js/montgomery-hip.js');"
            "try{m.Weierstrass.create(m.bls12377Params);console.log('CREATED')}catch(e){console.log('THROWN '+e.message)}")
    out = subprocess.run([NODE, "-e", code], cwd=ROOT, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.startswith("THROWN") and "msm error 5" in out.stdout


@pytest.mark.gpu
def test_js_facade_on_gpu(addon):
    out = subprocess.run([NODE, "js/test-compute-msm.js"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout


@pytest.mark.gpu
def test_js_benchmark_with_the_reference_protocol(addon):
    """js/bench-msm.js: the call sequence of scripts/msm-weierstrass.ts:12-51 on this facade, unchanged call by call
    (startThreads, `let [ptr] = await randomPointsFast(N)`, 15 runs of msmUnsafe, Projective.toAffine / Affine.toBigint)."""
    import json

    out = subprocess.run([NODE, "js/bench-msm.js", "14"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["consistent"] and rep["runs"] == 10 and rep["n"] == 14 and rep["median_ms"] > 0
    # the reference's `log`, printed by the script as the reference prints it: the parameters, then >= 6 "label... x.xms" lines
    lines = out.stdout.splitlines()
    assert sum(1 for l in lines if l.rstrip().endswith("ms") and "... " in l and not l.startswith("msm (n=")) >= 6, out.stdout
    assert any(l.startswith("msm total... ") for l in lines)


@pytest.mark.gpu
def test_js_field_operator_table_replays_the_golden_vectors(addon):
    """Field.multiply / square / add / subtract / inverse / batchInverse, Scalar.decompose and Affine.batchAdd through the N-API
    boundary (the reference's fine wasm exports, src/field-msm.ts:190-243, src/scalar-glv.ts:105-128) against
    tests/golden/fp377.json, glv377.json and point_add377.json."""
    import json

    out = subprocess.run([NODE, "js/test-field.js"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["ok"] and rep["fp_cases"] >= 30 and rep["glv_cases"] >= 60 and rep["batch_add_cases"] >= 5, rep


@pytest.mark.gpu
def test_js_scalars_are_device_resident(addon):
    """The JS facade keeps scalars in HBM from scalarsFromBytes / randomScalars on (src/parallel.ts:119-133): the timed msm
    of js/bench-msm.js at 2^20 must be the resident-scalar time, i.e. close to the Python facade's run_device on the same box."""
    import json
    import statistics
    import time

    from montgomery_amd.api import MsmContext

    out = subprocess.run([NODE, "js/bench-msm.js", "20"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    js_ms = json.loads(out.stdout.strip().splitlines()[-1])["median_ms"]
    ctx = MsmContext()
    n = 1 << 20
    ctx.generate_points(n, seed=1)
    dev, _ = ctx.generate_scalars(n, seed=2)
    for _ in range(5):
        ctx.run_device(dev, n)
    ts = []
    for _ in range(10):
        t = time.perf_counter()
        ctx.run_device(dev, n)
        ts.append((time.perf_counter() - t) * 1e3)
    ctx.close()
    py_ms = statistics.median(ts)
    # a host Buffer per call cost +0.9 ms of 3.9 at 2^20 (round 3); resident scalars must be within 10 % of the ctypes path
    assert js_ms <= 1.10 * py_ms, (js_ms, py_ms)


@pytest.mark.gpu
@pytest.mark.parametrize("curve", [0, 1, 2, 3])
def test_plain_c_host_of_the_abi(curve):
    """examples/msm_demo.c: a C program on the C ABI alone (what a cgo / JNI binding would do): generates inputs on
    the GPU, checks that the MSM does not depend on the window size nor on one-window sharding + msm_combine_curve."""
    import subprocess

    exe = os.path.join(ROOT, "examples", "msm_demo")
    from conftest import build_if_missing

    build_if_missing("demo", "examples/msm_demo")
    out = subprocess.run([exe, "14", str(curve)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK: result independent" in out.stdout
