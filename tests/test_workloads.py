"""montgomery_amd/workloads.py (host side, no GPU): the synthetic scalar distributions bench.py's `skewed` leg and
tests/test_gpu_skew.py run -- shapes, determinism, and that every scalar stays below 2^252 (< q for all four curves)."""
import numpy as np

from montgomery_amd import workloads


def test_shapes_determinism_and_bounds():
    n = 4096
    for kind in workloads.KINDS:
        a = workloads.scalars(kind, n, seed=5)
        b = workloads.scalars(kind, n, seed=5)
        assert a.shape == (n, 32) and a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
        assert np.array_equal(a, b)
        assert int(a[:, 31].max()) < 16                      # below 2^252
    assert not np.array_equal(workloads.scalars("uniform", n, seed=5), workloads.scalars("uniform", n, seed=6))


def test_prover_shape_and_one_scalar():
    n = 1 << 16
    s = workloads.scalars("prover", n, seed=1)
    zero = (s.sum(axis=1) == 0)
    one = (s[:, 0] == 1) & (s[:, 1:].sum(axis=1) == 0)
    small = (s[:, 2:].sum(axis=1) == 0) & ~zero & ~one
    assert abs(zero.mean() - 0.40) < 0.02 and abs(one.mean() - 0.20) < 0.02 and abs(small.mean() - 0.10) < 0.02
    o = workloads.scalars("one", n, seed=2)
    assert (o == o[0]).all() and o[0].any()
