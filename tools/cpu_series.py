"""CPU baseline series: the C port of the oracle on the host cores of the GPU box at
N = 2^14, 2^16, 2^18, 2^20 (the sizes the reference's own benchmark scripts run, scripts/msm-weierstrass.ts), median of
a few runs after a discarded warm-up like scripts/evaluate-util.ts:3-20.  Writes one JSON object.
usage: python tools/cpu_series.py [out.json]"""
import json, os, statistics, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import c_oracle as co, msm_oracle as O

C = O.BLS12_377
base, _ = O.random_points_bls377("cpu-series", 512)
pb = O.points_to_bytes(base, 48)
out = {"kind": "port", "what": "oracle/msm_oracle.c, BLS12-377 G1 batched-affine MSM, reference window table", "cpu_count": os.cpu_count(), "series": []}
rng = np.random.default_rng(20261002)
for lg in (14, 16, 18, 20):
    n = 1 << lg
    pts = pb * (n // 512)
    sc = rng.integers(0, 256, size=n * 32, dtype=np.uint8)
    sc[31::32] &= 0x0F       # < 2^252 < q
    scb = sc.tobytes()
    times, threads = [], 0
    for rep in range(6):
        t = time.perf_counter()
        _, threads = co.msm_bls377(pts, scb, 0)
        times.append(time.perf_counter() - t)
    times = times[1:]
    out["series"].append({"log2_n": lg, "threads": threads, "median_ms": statistics.median(times) * 1e3,
                          "std_ms": statistics.stdev(times) * 1e3, "points_per_s": n / statistics.median(times)})
    print(out["series"][-1], flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w"), indent=1)
