"""Uniform against skewed scalar distributions (montgomery_amd/workloads.py) on one build or several:
    python tools/skew_time.py LOG2N [lib.so ...]      ('-' or nothing = the in-tree build)
Per distribution: one warm-up call, five timed calls (median), the result checked against the known discrete logs."""
import json, os, statistics, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json, statistics
sys.path.insert(0, %r)
from montgomery_amd.api import MsmContext
from montgomery_amd import workloads
from oracle import c_oracle, msm_oracle as O
lg = int(sys.argv[1]); n = 1 << lg
C = O.BLS12_377
ctx = MsmContext()
a = ctx.generate_points(n, seed=7, want_scalars=True, raw=True)
dev = ctx.device_alloc(32 * n)
out = {}
for kind in workloads.KINDS:
    s = workloads.scalars(kind, n, seed=11)
    sb = s.tobytes()
    ctx.device_upload(dev, sb)
    exp = O.aff_scale(c_oracle.dot_mod(a, sb, n, C.q), (C.gx, C.gy), C.p)
    r, info = ctx.run_device(dev, n, no_tables=True)
    ok = r.as_tuple() == exp
    ms = []
    for i in range(5):
        t = time.perf_counter(); r, info = ctx.run_device(dev, n, no_tables=True); ms.append((time.perf_counter() - t) * 1e3)
    out[kind] = {"ms": round(statistics.median(ms), 3), "ok": ok, "max_bucket": info["max_bucket"], "rounds": info["rounds"], "c": info["c"],
                 "phase": {k: round(v, 2) for k, v in info["phase_ms"].items()}}
    del s, sb
print(json.dumps(out))
''' % ROOT

def main():
    lg = sys.argv[1]
    libs = sys.argv[2:] or ["-"]
    for lib in libs:
        env = dict(os.environ)
        if lib != "-":
            env["MSM_HIP_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "-c", CHILD, lg], env=env, capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-600:]
        print(lib, lg, line, flush=True)

main()
