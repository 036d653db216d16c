"""Run-to-run drift of the 2^20 MSM: 24 back-to-back calls over eight scalar sets, wall time and phases of each (the chip gives
back clock under sustained load: the 15-run protocol of bench.py reads 3 - 5 % above a best-of-four): python tools/var20.py"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from montgomery_amd.api import MsmContext
n = 1 << 20
ctx = MsmContext()
ctx.generate_points(n, seed=20261022)
dev = torch.device("cuda", 0)
scal = [torch.empty(n * 32, dtype=torch.uint8, device=dev) for _ in range(8)]
for i, t in enumerate(scal): ctx.generate_scalars(n, seed=3000 + i, into=t.data_ptr())
torch.cuda.synchronize()
for i in range(5): ctx.run_device(scal[i % 8].data_ptr(), n)
for i in range(24):
    ts = time.perf_counter(); r, info = ctx.run_device(scal[i % 8].data_ptr(), n); dt = (time.perf_counter() - ts) * 1e3
    p = info["phase_ms"]
    print(f"set {i % 8} wall {dt:.3f} total {p['total']:.3f} sort {p['sort']:.3f} acc {p['accumulate']:.3f} r1 {p['accumulate_round1']:.3f} red {p['reduce']:.3f} maxb {info.get('max_bucket')} rounds {info.get('rounds')} pairs {info.get('n_pairs')}")
