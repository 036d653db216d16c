"""Team shape of the CPU port (oracle/msm_oracle.c) on the host cores this process is granted: windows side by side
(ORACLE_WPAR of them at a time) x threads per window.  No torch in this process: once torch is imported its OpenMP
runtime grants nested teams one thread.   usage: python tools/cpu_teams.py [LOG2N]"""
import os, statistics, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, statistics
sys.path.insert(0, %r)
import numpy as np
from oracle import c_oracle as co, msm_oracle as O
lg = int(sys.argv[1]); n = 1 << lg
base, _ = O.random_points_bls377("cpu-teams", 512)
pts = O.points_to_bytes(base, 48) * (n // 512)
sc = np.random.default_rng(7).integers(0, 256, size=n * 32, dtype=np.uint8); sc[31::32] &= 0x0F
scb = sc.tobytes()
ts = []
for rep in range(4):
    t = time.perf_counter(); _, thr = co.msm_bls377(pts, scb, 0); ts.append(time.perf_counter() - t)
print(thr, round(statistics.median(ts[1:]), 3), [round(x, 3) for x in ts], n / statistics.median(ts[1:]))
''' % ROOT
lg = sys.argv[1] if len(sys.argv) > 1 else "20"
print("cpu_count", os.cpu_count(), "quota", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)
for wpar in ("", "1", "2", "4", "8"):
    env = dict(os.environ)
    if wpar: env["ORACLE_WPAR"] = wpar
    out = subprocess.run([sys.executable, "-c", CHILD, lg], env=env, capture_output=True, text=True)
    print("wpar", wpar or "default", "->", out.stdout.strip() or out.stderr[-300:], flush=True)
