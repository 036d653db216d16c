import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_if_missing(target: str, path: str) -> None:
    """`make target` only when its product is absent.  The test process may already have libmsm_hip.so loaded: a
    timestamp-triggered relink of the library underneath it must never happen as a side effect of a test."""
    import subprocess

    if not os.path.exists(os.path.join(ROOT, path)):
        subprocess.check_call(["make", "-C", ROOT, "-s", target])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def c_oracle():
    from oracle import c_oracle as co

    co.load()
    return co


@pytest.fixture(scope="session")
def gpu_ctx():
    """One BLS12-377 context for the whole GPU session.  Fails loudly (no CPU fallback) without a GPU."""
    from montgomery_amd.api import MsmContext

    ctx = MsmContext()
    yield ctx
    ctx.close()
