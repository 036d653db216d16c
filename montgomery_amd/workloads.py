"""Synthetic scalar distributions for benchmarks and tests (host side, numpy): what a prover hands an MSM is rarely uniform.

The reference walks any bucket-size distribution through the same rounds (src/msm-batched-affine.ts:204,243-263: "until
maxBucketSize"); these sets put numbers on how far from the uniform time a skewed one runs (bench.py `skewed`, tests/test_gpu_skew.py):
  uniform   n scalars below 2^252 (< q for every curve here)
  prover    the shape of a witness vector: 40 % zeros, 20 % ones, 10 % below 2^16, the rest uniform
            (scripts/zprize23/submission-bls377.ts:45-57 draws such inputs for the reference's own timing)
  one       ONE scalar repeated: every entry of a window falls into the same bucket
"""
from __future__ import annotations

import numpy as np

KINDS = ("uniform", "prover", "one")


def scalars(kind: str, n: int, seed: int = 1) -> np.ndarray:
    """n x 32 bytes (uint8, little-endian 256-bit scalars below 2^252)."""
    rng = np.random.default_rng(seed)
    if kind == "one":
        s = rng.integers(0, 256, size=32, dtype=np.uint8)
        s[31] &= 0x0F
        return np.tile(s, (n, 1))
    out = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    out[:, 31] &= 0x0F
    if kind == "uniform":
        return out
    if kind != "prover":
        raise ValueError(f"unknown scalar distribution {kind!r}")
    u = rng.random(n)
    zero = u < 0.40
    one = (u >= 0.40) & (u < 0.60)
    small = (u >= 0.60) & (u < 0.70)
    out[zero | one, :] = 0
    out[one, 0] = 1
    out[small, 2:] = 0
    return out
