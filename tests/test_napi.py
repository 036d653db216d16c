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
    for n in ("createContext", "destroyContext", "setPoints", "msm", "plan"):
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
