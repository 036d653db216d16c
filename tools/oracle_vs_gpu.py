"""The C oracle (oracle/msm_oracle.c, all host cores, windows in parallel teams) against the GPU on generated inputs:
python tools/oracle_vs_gpu.py LOG2N [LOG2N ...]   (checker tooling, not part of the product)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montgomery_amd.api import MsmContext
from oracle import c_oracle

ctx = MsmContext()
for lg in [int(x) for x in sys.argv[1:]] or [16, 18, 20]:
    n = 1 << lg
    ctx.generate_points(n, seed=31 + lg)
    pts = ctx.get_points(0, n)
    dev, sc = ctx.generate_scalars(n, seed=77 + lg, to_host=True)
    got, _ = ctx.run_device(dev, n)
    t = time.perf_counter(); ref, th = c_oracle.msm_bls377(pts, sc, 0); dt = time.perf_counter() - t
    print(f"2^{lg}: oracle on {th} threads {dt:.2f} s, equal to the GPU result: {got.as_tuple() == ref}", flush=True)
