"""Context life cycle: device and pinned memory come back when a context is destroyed.  python tools/leak_check.py [cycles]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from montgomery_amd import _lib
from montgomery_amd.api import MsmContext

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.cuda.init()
free0 = torch.cuda.mem_get_info()[0]
lo = free0
for i in range(cycles):
    ctx = MsmContext([_lib.CURVE_BLS12_377_G1, _lib.CURVE_ED_ON_BLS12_377, _lib.CURVE_PALLAS][i % 3])
    n = (1 << 21) + 77 * i
    ctx.generate_points(n, seed=i)
    dev, host = ctx.generate_scalars(n, seed=i + 1, to_host=True)
    a, _ = ctx.run_device(dev, n)
    b, _ = ctx.run(host)                      # staged upload: pinned chunks, copy streams
    assert a.as_tuple() == b.as_tuple()
    if i % 2: ctx.set_workspace_limit(64 << 20); ctx.run_device(dev, n)
    ctx.close()
    free = torch.cuda.mem_get_info()[0]
    lo = min(lo, free)
    if i == 2: base = free                    # the runtime's own one-time pools (code objects, signals) are in by now
    if i % 5 == 0: print(i, "free", free >> 20, "MiB  delta", (free0 - free) >> 20, "MiB", flush=True)
free = torch.cuda.mem_get_info()[0]
print("one-time MiB:", (free0 - base) >> 20, " growth over the later cycles MiB:", (base - free) >> 20)
sys.exit(0 if base - free < (16 << 20) else 1)
