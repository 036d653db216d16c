"""Several settings of the tuning build's MSM_* knobs timed in ONE process on the same resident inputs, interleaved.
usage: MSM_HIP_LIB=ab_builds/libmsm_tune.so python tools/knob_matrix.py LOG2N [REPS] -- "K1=v K2=v" "K3=v" ...
       ("" = the defaults; AB_SERIAL=1 in a setting runs the window groups one after the other; MSM_C forces the window)
The knobs are read with getenv at call time (builds made with -DMSM_TUNING only), so one context serves every setting."""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from montgomery_amd.api import MsmContext  # noqa: E402


def main():
    args = sys.argv[1:]
    cut = args.index("--")
    lg = int(args[0])
    reps = int(args[1]) if cut > 1 else 5
    settings = args[cut + 1:] or [""]
    n = 1 << lg
    ctx = MsmContext(int(os.environ.get("AB_CURVE", "0")))
    ctx.generate_points(n, seed=7)
    devs = [ctx.generate_scalars(n, seed=9 + i)[0] for i in range(2)]
    ctx.run_device(devs[0], n)
    times = {s: [] for s in settings}
    phases = {}
    ref = None
    for rep in range(reps + 1):
        for s in settings:
            kv = dict(x.split("=", 1) for x in s.split())
            for k, v in kv.items():
                os.environ[k] = v
            serial = bool(int(kv.get("AB_SERIAL", "0")))
            cc = int(kv.get("MSM_C", "0")) or None
            t = time.perf_counter()
            r, info = ctx.run_device(devs[rep % 2], n, c=cc, serial=serial)
            dt = (time.perf_counter() - t) * 1e3
            for k in kv:
                del os.environ[k]
            if rep % 2 == 0:
                if ref is None:
                    ref = r.as_tuple()
                elif r.as_tuple() != ref:
                    print("RESULT DIFFERS under", repr(s), flush=True)
            if rep:   # the first pass of every setting is a warm-up (workspace growth)
                times[s].append(dt)
                phases[s] = info["phase_ms"]
    for s in settings:
        t = times[s]
        print(json.dumps({"setting": s, "median_ms": round(statistics.median(t), 2), "min_ms": round(min(t), 2),
                          "max_ms": round(max(t), 2), "phase": {k: round(v, 2) for k, v in phases[s].items()}}), flush=True)
    ctx.close()


main()
