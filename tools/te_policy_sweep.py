import sys, time, os, subprocess, json
sys.path.insert(0, "/root/repo")
CHILD = r'''
import sys, time, json
sys.path.insert(0, "/root/repo")
from montgomery_amd import _lib
from montgomery_amd.api import MsmContext
lg = int(sys.argv[1]); n = 1 << lg
ctx = MsmContext(_lib.CURVE_ED_ON_BLS12_377)
ctx.generate_points(n, seed=7)
dev, _ = ctx.generate_scalars(n, seed=9)
ctx.run_device(dev, n)
best = 1e9
for i in range(4):
    t = time.perf_counter(); r, info = ctx.run_device(dev, n); best = min(best, time.perf_counter() - t)
print(json.dumps({"ms": round(best * 1e3, 3), "phase": {k: round(v, 2) for k, v in info["phase_ms"].items()}}))
'''
for lg in (16, 18, 20, 22):
    for v in ("", "MSM_PBL=4", "MSM_PBL=16", "MSM_FINISH_MAX=8", "MSM_FINISH_MAX=16", "MSM_FINISH_MAX=64", "MSM_TAIL_MIN=262144", "MSM_TC=2", "MSM_TC=8", "MSM_GROUPS=2"):
        env = dict(os.environ)
        if v:
            k, val = v.split("="); env[k] = val
        out = subprocess.run([sys.executable, "-c", CHILD, str(lg)], env=env, capture_output=True, text=True)
        print(lg, f"{v:22s}", out.stdout.strip()[:200] or out.stderr[-200:], flush=True)
