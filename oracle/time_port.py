"""Times the C port of the oracle (msm_oracle.c) in a process of its own -- test infrastructure, run only by bench.py's
cpu_baseline leg.  A separate process because an imported torch brings its own OpenMP runtime, under which the port's
nested teams (windows side by side, a team of threads each) are granted one thread per team.

usage: python oracle/time_port.py DIR BUDGET_SECONDS LOG2N[:REPS] [LOG2N[:REPS] ...]
DIR holds points.bin (96 B per point, affine little-endian) and scalars.bin (32 B each); every size takes the first 2^k of
them.  Prints one JSON object: per size the timed runs (after one untimed warm-up at 2^16), the result and the threads used.
REPS (default 3) bounds the repeats of one size; a size given with REPS is run at least once whatever the budget says (the one
call at the headline size, ~70 s at 2^26 on 14 threads)."""
import json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle


def main():
    d, budget = sys.argv[1], float(sys.argv[2])
    sizes = [(int(a.split(":")[0]), int(a.split(":")[1]) if ":" in a else 3) for a in sys.argv[3:]]
    import mmap

    fp, fs = open(os.path.join(d, "points.bin"), "rb"), open(os.path.join(d, "scalars.bin"), "rb")
    import ctypes as C

    # 6.4 GB of points at 2^26: mapped (copy-on-write, so that ctypes may take views of it), never copied
    pts = mmap.mmap(fp.fileno(), 0, access=mmap.ACCESS_COPY)
    sc = mmap.mmap(fs.fileno(), 0, access=mmap.ACCESS_COPY)
    view = lambda m, nbytes: (C.c_uint8 * nbytes).from_buffer(m)
    lib = c_oracle.load()
    t_all = time.perf_counter()
    w = min(1 << 16, len(sc) // 32)
    c_oracle.msm_bls377(view(pts, 96 * w), view(sc, 32 * w), 0)   # thread pool, page faults
    out = {"quota": lib.oracle_cpu_quota(), "cpu_count": os.cpu_count(), "series": []}
    for lg, reps in sizes:
        n = 1 << lg
        times, ref, threads = [], None, 1
        for rep in range(reps):
            t0 = time.perf_counter()
            ref, threads = c_oracle.msm_bls377(view(pts, 96 * n), view(sc, 32 * n), 0)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all + times[-1] > budget:   # the next repeat would not fit the bounded sample
                break
        out["series"].append({"log2_n": lg, "times_s": times, "threads": threads, "window_bits": lib.oracle_window_size(lg),
                              "result": None if ref is None else [hex(ref[0]), hex(ref[1])]})
    print(json.dumps(out))


main()
