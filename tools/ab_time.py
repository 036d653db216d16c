"""A/B timing of alternative builds of libmsm_hip.so (each in its own process, interleaved).
usage: python tools/ab_time.py LOG2N lib1.so lib2.so ...   ('-' = the in-tree build)"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
from montgomery_amd.api import MsmContext
lg = int(sys.argv[1]); n = 1 << lg
import os
cc = int(os.environ.get('MSM_C', '0')) or None
ctx = MsmContext(int(os.environ.get('AB_CURVE', '0')))
ctx.generate_points(n, seed=7)
dev, _ = ctx.generate_scalars(n, seed=9)
serial = bool(int(os.environ.get('AB_SERIAL', '0')))
ctx.run_device(dev, n, c=cc)
best = None
for i in range(4):
    t = time.perf_counter(); r, info = ctx.run_device(dev, n, c=cc, serial=serial); dt = time.perf_counter() - t
    if best is None or dt < best[0]: best = (dt, info)
print(json.dumps({"ms": best[0] * 1e3, "x": hex(r.x)[:18], "phase": {k: round(v, 2) for k, v in best[1]["phase_ms"].items()}, "c": best[1]["c"]}))
''' % ROOT

def main():
    lg = sys.argv[1]
    libs = sys.argv[2:]
    for rep in range(int(os.environ.get('AB_REPS', '2'))):
        for lib in libs:
            env = dict(os.environ)
            if lib != "-":
                env["MSM_HIP_LIB"] = os.path.abspath(lib)
            out = subprocess.run([sys.executable, "-c", CHILD, lg], env=env, capture_output=True, text=True)
            print(lib, out.stdout.strip() or out.stderr[-400:], flush=True)

main()
