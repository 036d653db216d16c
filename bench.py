#!/usr/bin/env python3
"""Headline benchmark: BLS12-377 G1 MSM throughput (points/s) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2n 26] [--c C]

One "step" = one MSM over resident base points with FRESH scalars already in HBM (the reference's
protocol: points pre-loaded, new random scalars every run, scripts/msm-weierstrass.ts:12-51).
N = 1: the whole MSM on one GPU.  N > 1, one rank per GPU: the same MSM sharded across the ranks.  The headline is
`--split windows` (BASELINE configs[4], the north star: every rank holds all points and scalars and computes the
window sums P_k of its windows; windows are independent until src/msm-batched-affine.ts:312-333); the other split
(`points`: every rank runs all K windows on its share of the points and needs only that share of the scalars) is
timed by the same run and reported as `other_splits`.  Either way ONE RCCL all-gather of K x 144 bytes per rank,
rank 0 does the Horner combination.  Total work is fixed as N grows: scaling = "strong".
Launch: under torch.distributed.run (the driver's form: WORLD_SIZE must equal --gpus, anything else is an error),
or plainly as `python bench.py --gpus N`: the script then starts the N ranks itself, as fresh child processes of
torch.distributed.run, BEFORE it imports torch or touches the GPU, relays their output and exits with their code.

Prints ONE JSON line on rank 0 with the driver's contract plus
  `roofline`      dominant kernel k_batch_add, HIP-event timed inside the library on its own stream,
  `cpu_baseline`  the C port of the oracle on the host cores this box grants (its cgroup CPU quota), bounded sample, in a child process,
  `verified`      the result of the LAST timed step checked outside the timed region against the known
                  discrete logs of the generated points: sum s_i P_i == (sum s_i a_i mod q) G
                  (the reference compares every size it benchmarks, scripts/msm-weierstrass.ts:97-107),
  `median_ms` / `std_ms` over the timed steps (scripts/evaluate-util.ts:3-20) and `pcie_inclusive`.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes of one affine pair addition in k_batch_add: read two 96-byte points, write one
PAIR_ALGO_BYTES = 288
# 32x32->64 multiply-adds of one pair addition: 5 multiplications (325 each) + 1 squaring (247), 13 x 30-bit limbs
PAIR_MADS = 5 * 325 + 247
SORT_ALGO_BYTES = 12   # per (entry, window): digit read by k_hist, digit read by k_scatter_lds, payload write
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
# v_mad_u64_u32 lane-ops/s, measured by tools/ubench_int2.hip (profiles/r02_ubench_int.txt): 33.4e12 with eight resident
# waves per SIMD (4.6 real cycles per wave-instruction at 2.35 GHz); 28.2e12 at the two waves per SIMD k_batch_add runs with
INT_MAD_PEAK = 33.4e12
INT_MAD_AT_2_WAVES = 28.2e12
# Clock the chip holds under k_batch_add (GRBM_GUI_ACTIVE / 8 / wall time, profiles/r02_pmc_2p24.json: 1.79 GHz in the
# regular rounds, 2.00 in the gather round, half the pairs each) against the 2.39 GHz the micro-benchmark kernels hold
HELD_CLOCK_GHZ = 1.89
UBENCH_CLOCK_GHZ = 2.39
# SURVEY.md section 8(d) prices a 377-bit multiplication at 300 word multiplications (6 x 300 = 1 800 per pair addition); the
# kernel's own count for 13 limbs is PAIR_MADS.  `frac` uses PAIR_MADS, `frac_1800_basis` the survey's figure.
PAIR_MADS_SURVEY = 1800
# HBM bytes per pair addition of the tree kernel: from the PMC passes at the headline size committed under profiles/
# (tools/pmc_headline.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled: gfx950 halves wide coalesced
# reads); the round-3 constant (2^24, regular rounds only) if that file is missing
PAIR_TRAFFIC_BYTES_PMC = 488
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_2p26.json")


def pmc_summary(curve, log2n, window_bits):
    """The committed PMC summary of THIS configuration (tools/pmc_headline.sh: separate rocprofv3 --pmc passes over this
    command), or None: counters taken at another size, curve or window say nothing about this run."""
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
        if curve != "bls12-377" or log2n != 26 or d.get("window_bits") != window_bits:
            return None
        return d
    except (OSError, KeyError, ValueError):
        return None


MAX_SCALAR_SETS = 8            # distinct 2^n x 32-byte scalar sets kept in HBM; steps cycle through them


def expected_from_logs(curve_name, a_host, s_host, n):
    """(sum s_i a_i mod q) G through the checker: C oracle for the dot product, Python oracle for the one scalar
    multiplication.  Returns the affine point (None = identity)."""
    from oracle import c_oracle
    from oracle import msm_oracle as O

    if curve_name == "ed377":
        C = O.ED_ON_BLS12_377
        k = c_oracle.dot_mod(a_host, s_host, n, C.q)
        return O.te_to_affine(O.te_scale(k, O.te_from_affine((C.gx, C.gy), C), C), C)
    C = {"bls12-377": O.BLS12_377, "bls12-381": O.BLS12_381}[curve_name]
    k = c_oracle.dot_mod(a_host, s_host, n, C.q)
    return O.aff_scale(k, (C.gx, C.gy), C.p)


def C_uint8_view(arr):
    """a C-contiguous numpy uint8 array as a ctypes array over the same memory (no copy: 2 GB at 2^26)"""
    import ctypes

    flat = arr.reshape(-1)
    return (ctypes.c_uint8 * flat.size).from_buffer(flat)


def cpu_baseline(ctx, log2n_sample, seed, log2n_headline=None):
    """Times oracle/msm_oracle.c (kind "port") on the host cores over the first 2^k resident points, k = 20 and
    log2n_sample: one untimed warm-up call (thread pool, page faults), then repeated timed calls per size -- median and
    sample standard deviation as everywhere else.  The port runs in a child process (oracle/time_port.py) on the same bytes:
    under the OpenMP runtime torch brings into THIS process its nested teams get one thread each (round 2 and the first
    lines of round 3 timed it on 7-8 threads that way).  The reference's WASM path cannot run here."""
    import subprocess
    import tempfile

    sizes = sorted({min(20, log2n_sample), log2n_sample})
    # ... and ONE call at the headline size (~70 s at 2^26 on the 14 - 16 threads the boxes grant): the port on the very
    # configuration `value` is quoted on
    one_call = log2n_headline if log2n_headline and log2n_headline > sizes[-1] else None
    n_top = 1 << (one_call or sizes[-1])
    pts = ctx.get_points(0, n_top)
    _, sc = ctx.generate_scalars(n_top, seed=seed, to_host=True)
    with tempfile.TemporaryDirectory(prefix="msm_cpu_") as d:
        with open(os.path.join(d, "points.bin"), "wb") as f:
            f.write(pts)
        with open(os.path.join(d, "scalars.bin"), "wb") as f:
            f.write(sc)
        env = {k: v for k, v in os.environ.items() if not k.startswith("OMP_")}
        out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "time_port.py"), d, "30"] + [str(lg) for lg in sizes]
                             + ([f"{one_call}:1"] if one_call else []), env=env, capture_output=True, text=True, timeout=1200)
    del pts
    if out.returncode != 0:
        raise RuntimeError("oracle/time_port.py failed: " + out.stderr[-500:])
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    series, threads = [], 1
    for e in rep["series"]:
        n = 1 << e["log2_n"]
        got, _ = ctx.run(sc[:32 * n])
        ref = None if e["result"] is None else (int(e["result"][0], 16), int(e["result"][1], 16))
        assert got.as_tuple() == ref, "GPU result differs from the CPU oracle on the cpu_baseline sample"
        times = e["times_s"]
        threads = e["threads"]
        series.append({"log2_n": e["log2_n"], "runs": len(times), "median_s": statistics.median(times),
                       "std_s": statistics.stdev(times) if len(times) > 1 else None, "points_per_s": n / statistics.median(times),
                       "threads": threads, "window_bits": e["window_bits"]})
    top = [e for e in series if e["log2_n"] == sizes[-1]][0]   # `value`: the repeated sample (median of 3); the one call at the headline size is series[-1]
    threads = top["threads"]
    quota = rep.get("quota") or 0
    return {
        "value": top["points_per_s"],
        "unit": "points/s",
        "cores": threads,
        "host_cpus": os.cpu_count(),
        "cpu_quota": quota or None,
        "threads": threads,
        "kind": "port",
        "per_thread": top["points_per_s"] / threads,
        "series": series,
        "sample": f"BLS12-377 G1 MSMs over the first 2^{top['log2_n']} resident points (reference window table, c = {top['window_bits']}), "
                  f"{top['runs']} timed calls after a warm-up, median {top['median_s']:.2f} s on {threads} OpenMP threads"
                  + (f" (the box grants {quota} CPUs through its cgroup quota; {os.cpu_count()} logical CPUs are visible)" if quota else "")
                  + " (windows side by side, a team of threads each; inside a window entries split across the team for slicing / "
                  "sorting, buckets for the accumulation and reduction, as the reference's SPMD threads); GPU result on the same "
                  "inputs checked equal. A correctness checker first: the reference publishes 6.8e4 points/s per wasm thread at "
                  "2^16 on a laptop (doc/zprize23.md:119-123); compare per_thread"
                  + (f".  series[-1]: ONE call at the headline size 2^{series[-1]['log2_n']} (c = {series[-1]['window_bits']}): "
                     f"{series[-1]['median_s']:.1f} s = {series[-1]['points_per_s']:.3g} points/s" if one_call else ""),
    }


def step_stats(step_ms):
    return {"median_ms": statistics.median(step_ms), "std_ms": statistics.stdev(step_ms) if len(step_ms) > 1 else 0.0,
            "min_ms": min(step_ms), "max_ms": max(step_ms)}


def timed_config(curve_name, log2n, torch, steps=10, warmup=5, c=0):
    """One more BASELINE config timed in-process with the protocol of scripts/msm-weierstrass.ts:12-51 (15 runs, the
    first 5 discarded, median and sample std-dev): its own context, points P_i = a_i G generated on the GPU, fresh
    resident scalars per run, last result checked against the known discrete logs.  Returns the `other_configs` entry."""
    from montgomery_amd import _lib
    from montgomery_amd.api import MsmContext

    n = 1 << log2n
    te = curve_name == "ed377"
    ctx = MsmContext(_lib.CURVE_ED_ON_BLS12_377 if te else _lib.CURVE_BLS12_377_G1, device=0)
    a_host = ctx.generate_points(n, seed=20261002 + log2n, want_scalars=True, raw=True)
    cc, K = ctx.plan(n, c or None)
    dev = torch.device("cuda", 0)
    n_sets = min(steps + warmup, MAX_SCALAR_SETS)
    scal = [torch.empty(n * 32, dtype=torch.uint8, device=dev) for _ in range(n_sets)]
    for i, t in enumerate(scal):
        ctx.generate_scalars(n, seed=3000 + i, into=t.data_ptr())
    torch.cuda.synchronize()
    for i in range(warmup):
        ctx.run_device(scal[i % n_sets].data_ptr(), n, c=cc)
    torch.cuda.synchronize()
    infos, step_ms, last, last_set = [], [], None, 0
    t0 = time.perf_counter()
    for i in range(steps):
        last_set = (warmup + i) % n_sets
        ts = time.perf_counter()
        last, info = ctx.run_device(scal[last_set].data_ptr(), n, c=cc)
        step_ms.append((time.perf_counter() - ts) * 1e3)
        infos.append(info)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the same MSMs on the plain path (no window tables: K sets of buckets and a Horner step, the path of rounds 1-4), for the record
    plain = None
    if infos[-1]["tables"]:
        p_ms, p_info = [], None
        for i in range(warmup + steps):
            ts = time.perf_counter()
            p_last, p_info = ctx.run_device(scal[i % n_sets].data_ptr(), n, no_tables=True)
            if i >= warmup:
                p_ms.append((time.perf_counter() - ts) * 1e3)
        same = (p_last.x, p_last.y) == (last.x, last.y)   # the last timed step of either loop ran on the same scalar set
        plain = {"ms": statistics.median(p_ms), "std_ms": statistics.stdev(p_ms), "window_bits": p_info["c"], "windows": p_info["K"],
                 "equals_tables_result": bool(same)}
    _, s_host = ctx.generate_scalars(n, seed=3000 + last_set, to_host=True, raw=True)
    exp = expected_from_logs(curve_name, a_host, s_host, n)
    verified = ((last.x, last.y) == exp) if te else (last.as_tuple() == exp)
    acc_ms = sum(x["phase_ms"]["accumulate"] for x in infos)
    pairs = sum(x["n_pairs_algo"] for x in infos)   # algorithmic pair additions (msm_result.n_pairs_algo)
    algo_bytes, mads = (384, 9 * 153) if te else (PAIR_ALGO_BYTES, PAIR_MADS)
    mad_rate = pairs * mads / (acc_ms * 1e-3)
    hbm = pairs * algo_bytes / (acc_ms * 1e-3) / 1e9
    tables_gib = ctx.tables_info()[2] / 2 ** 30
    ctx.close()
    del scal
    return {
        "workload": f"{'ed-on-bls12-377' if te else 'bls12-377-g1'}-msm-2^{log2n}",
        "window_bits": cc, "windows": K, "steps": steps, "warmup": warmup,
        "window_tables": ({"tables": K, "gib": tables_gib,
                           "note": "K resident tables 2^(c k) P of the point set (built once per set, like the point conversion): all "
                                   "windows share one set of buckets; `plain_path` = the same MSMs without them"} if infos[-1]["tables"] else None),
        "plain_path": plain,
        "ms": statistics.median(step_ms), "std_ms": statistics.stdev(step_ms), "ms_per_step": dt / steps * 1e3,
        "points_per_s": n * steps / dt, "verified": bool(verified),
        "roofline": {"kernel": "k_te_add" if te else "k_batch_add", "bound": "int-alu", "achieved": mad_rate, "peak": INT_MAD_PEAK,
                     "unit": "v_mad_u64_u32 lane-ops/s", "frac": mad_rate / INT_MAD_PEAK,
                     "hbm": {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_pair_add": algo_bytes}},
        "phase_ms": {k: sum(x["phase_ms"][k] for x in infos) / len(infos) for k in infos[0]["phase_ms"]},
    }


def bench_ed377(args, torch):
    """BASELINE configs[3]: 2^20 Ed-on-BLS12-377 MSM (msmBasic path) on one GPU.  Points: N distinct subgroup points
    P_i = a_i G generated on the GPU (resident); scalars: uniform < q, fresh per step, resident in HBM."""
    from montgomery_amd import _lib
    from montgomery_amd.api import MsmContext

    n = 1 << args.log2n
    ctx = MsmContext(_lib.CURVE_ED_ON_BLS12_377, device=0)
    a_host = ctx.generate_points(n, seed=20261002, want_scalars=True, raw=True)
    c, K = ctx.plan(n, args.c or None)
    dev = torch.device("cuda", 0)
    n_sets = min(args.steps + args.warmup, MAX_SCALAR_SETS)
    scal = [torch.empty(n * 32, dtype=torch.uint8, device=dev) for _ in range(n_sets)]
    for i, t in enumerate(scal):
        ctx.generate_scalars(n, seed=1000 + i, into=t.data_ptr())
    torch.cuda.synchronize()
    for i in range(args.warmup):
        ctx.run_device(scal[i % n_sets].data_ptr(), n, c=c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    infos, step_ms, last, last_set = [], [], None, 0
    for i in range(args.steps):
        last_set = (args.warmup + i) % n_sets
        ts = time.perf_counter()
        last, info = ctx.run_device(scal[last_set].data_ptr(), n, c=c)
        step_ms.append((time.perf_counter() - ts) * 1e3)
        infos.append(info)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _, s_host = ctx.generate_scalars(n, seed=1000 + last_set, to_host=True, raw=True)
    verified = (last.x, last.y) == expected_from_logs("ed377", a_host, s_host, n)
    acc_ms = sum(x["phase_ms"]["accumulate"] for x in infos)
    pairs = sum(x["n_pairs_algo"] for x in infos)
    # one unified extended addition: two 128-byte nodes in, one out; 9 multiplications of 9 limbs (2*81 - 9 MADs)
    algo_bytes, mads = 384, 9 * 153
    achieved = pairs * algo_bytes / (acc_ms * 1e-3) / 1e9
    out = {
        "metric": "Ed-on-BLS12-377 MSM throughput", "value": n * args.steps / dt, "unit": "points/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"ed-on-bls12-377-msm-2^{args.log2n}", "log2_n": args.log2n, "window_bits": c, "windows": K,
                   "parallelism": "single-gpu"},
        "verified": bool(verified),
        **step_stats(step_ms),
        "roofline": {"kernel": "k_te_add (bucket tree, unified extended additions)", "bound": "hbm", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_pair_add": algo_bytes,
                     "pair_adds_per_step": pairs / len(infos), "pair_adds_issued_per_step": sum(x["n_pairs"] for x in infos) / len(infos),
                     "int_mad": {"achieved": pairs * mads / (acc_ms * 1e-3), "peak": INT_MAD_PEAK,
                                 "frac": pairs * mads / (acc_ms * 1e-3) / INT_MAD_PEAK}},
        "phase_ms": {k: sum(x["phase_ms"][k] for x in infos) / len(infos) for k in infos[0]["phase_ms"]},
    }
    print(json.dumps(out), flush=True)
    ctx.close()
    if not verified:
        sys.exit("bench: the Edwards MSM result failed the known-discrete-log check")


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: N fresh ranks under torch.distributed.run (the driver's own command
    line), started before this process has imported torch or made any HIP call -- a process that has initialised the GPU
    must not be replaced or forked into ranks.  The children inherit stdout / stderr, so rank 0's JSON line is this
    command's JSON line; returns their exit code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)   # 15 runs, the first 5 discarded: scripts/msm-weierstrass.ts:27-50
    ap.add_argument("--log2n", type=int, default=26)
    ap.add_argument("--c", type=int, default=0)
    ap.add_argument("--cpu-log2n", type=int, default=22)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-headline", action="store_true", help="cpu_baseline: skip the one call of the CPU port at the headline size (~70 s at 2^26)")
    ap.add_argument("--no-verify", action="store_true", help="skip the known-discrete-log check of the last timed result")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-scalar (PCIe-inclusive) leg")
    ap.add_argument("--no-skewed", action="store_true", help="skip the leg over skewed scalar distributions (prover-shaped, one scalar repeated)")
    ap.add_argument("--no-tables-leg", action="store_true", help="skip the leg that builds window tables for the headline size and times the MSM on them")
    ap.add_argument("--no-c16", action="store_true", help="skip the serialised step at c = 16 (same_kernel_at_c16): counter runs "
                                                          "then hold MSMs of one plan only")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the in-process runs of BASELINE configs[1] (2^20 BLS12-377) and configs[3] (2^20 Ed-on-BLS12-377)")
    ap.add_argument("--split", choices=["auto", "windows", "points"], default="windows",
                    help="N > 1: the headline sharding -- by scalar window (default: BASELINE configs[4]) or by points; "
                         "auto = montgomery_amd.distributed.choose_split.  The other one is timed too (other_splits)")
    ap.add_argument("--no-other-splits", action="store_true", help="N > 1: time the headline split only")
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1 only: take the SHARDED code path with a process group of one rank (RCCL initialisation with a "
                         "device id, device all-reduce / all-gather, both splits) -- what a one-GPU box can exercise of the N > 1 path")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (functional check of the sharded path on one GPU)")
    ap.add_argument("--curve", choices=["bls12-377", "bls12-381", "ed377"], default="bls12-377",
                    help="ed377 = BASELINE configs[3]: twisted Edwards msmBasic path (single GPU, use --log2n 20); "
                         "bls12-381 = the same batched-affine path over the BLS12-381 G1 constants (no CPU baseline leg)")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("bench: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))       # nothing has touched the GPU yet: torch is not even imported
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit(f"bench: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: refusing to measure a different job "
                 f"than the one asked for (launch with --nproc-per-node {args.gpus}, or without a launcher)")

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    sharded = world > 1 or (args.force_dist and args.gpus == 1)
    if sharded:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # --force-dist without a launcher: a rendezvous of one
            import socket

            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
            s_.close()
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if local_rank >= torch.cuda.device_count():   # several ranks on one GPU: a functional check only, and only over gloo
            if args.dist_backend == "nccl":
                sys.exit(f"bench: rank {rank} has no GPU of its own ({torch.cuda.device_count()} visible, {world} ranks): RCCL "
                         "needs one device per rank (use --dist-backend gloo for a functional run on fewer GPUs)")
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)
    else:
        torch.cuda.set_device(0)

    from montgomery_amd.api import AffineResult, MsmContext
    from montgomery_amd.distributed import (choose_split, choose_window, point_shards, sharded_msm, sharded_msm_buckets, sharded_msm_points,
                                            window_shards)

    n = 1 << args.log2n
    if args.curve == "ed377":
        return bench_ed377(args, torch)
    from montgomery_amd import _lib as _abi

    is381 = args.curve == "bls12-381"
    verify = not args.no_verify
    ctx = MsmContext(_abi.CURVE_BLS12_381_G1 if is381 else _abi.CURVE_BLS12_377_G1, device=local_rank)
    # identical points on every rank; rank 0 keeps their discrete logs for the check of the result
    a_host = ctx.generate_points(n, seed=20261002, want_scalars=(verify and rank == 0), raw=True)
    c, K = ctx.plan(n, args.c or None)
    split = "none" if not sharded else (choose_split(n, world, K) if args.split == "auto" else args.split)

    def plan_for(how):
        """window size of one sharding (montgomery_amd.distributed.choose_window): the points split plans for a rank's share
        of the points, the window split wants K divisible by the rank count; --c overrides both"""
        if args.c or not sharded:
            return ctx.plan(n, args.c or None)
        def plan(m, cc, no_tables=False, merged=False):
            return ctx.plan(m, cc, no_tables=no_tables, merged=merged)

        return choose_window(plan, n, world, how)

    if sharded:
        c, K = plan_for(split)
    shards = window_shards(K, world)

    dev = torch.device("cuda", local_rank)
    # fresh scalars per step, generated on the GPU before the timed region (resident in HBM); at most MAX_SCALAR_SETS
    # distinct sets (2 GB each at 2^26), the steps cycle through them
    n_sets = min(args.steps + args.warmup, MAX_SCALAR_SETS)
    scal = [torch.empty(n * 32, dtype=torch.uint8, device=dev) for _ in range(n_sets)]
    for i, t in enumerate(scal):
        ctx.generate_scalars(n, seed=1000 + i, into=t.data_ptr())

    ddev = dev if args.dist_backend == "nccl" else "cpu"
    exchanges = {}

    def step(i, how, c=None, K=None):
        if not sharded:
            return ctx.run_device(scal[i % n_sets].data_ptr(), n, c=c_main)
        from montgomery_amd.distributed import PARTIAL_BYTES, ShardExchange

        if (how, K) not in exchanges:   # one pinned row + device twin + gathered tensor for every step of this sharding
            exchanges[(how, K)] = ShardExchange(PARTIAL_BYTES * K, ddev)
        exchange = exchanges[(how, K)]
        box = {}

        def my_window_sums(lo, hi):
            parts, box["info"] = ctx.window_sums(scal[i % n_sets].data_ptr(), n, lo, hi, c=c, on_device=True)
            return parts

        def my_point_sums(first, count):   # all K windows over this rank's share of the points: only its scalars are read
            # (merged: only combined afterwards -- the call may run on the window tables of this rank's range of the points, which
            # the library builds when the range comes back, i.e. during the warm-up steps)
            parts, box["info"] = ctx.window_sums(scal[i % n_sets].data_ptr() + 32 * first, count, 0, K, c=c, on_device=True,
                                                 point_lo=first, merged=True)
            return parts

        def my_bucket_sums(r, w):   # all K windows over all points, this rank's range of every window's buckets
            parts, box["info"] = ctx.window_sums(scal[i % n_sets].data_ptr(), n, 0, K, c=c, on_device=True, bucket_shard=(r, w))
            return parts

        tm = {}
        if how == "points":
            out = sharded_msm_points(my_point_sums, n, K, c, device=ddev, curve=ctx.curve, timing=tm, exchange=exchange)
        elif how == "buckets":
            out = sharded_msm_buckets(my_bucket_sums, K, c, device=ddev, curve=ctx.curve, timing=tm, exchange=exchange)
        else:
            out = sharded_msm(my_window_sums, K, c, device=ddev, curve=ctx.curve, timing=tm, exchange=exchange)
        if box.get("info") is not None:
            box["info"]["all_gather_ms"] = tm.get("all_gather_ms")
        res = None
        if out is not None:
            xy = out[1]
            res = AffineResult(x=xy[0] if xy else 0, y=xy[1] if xy else 0, isZero=xy is None)
        return res, box.get("info")

    def sync():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    c_main = c

    def timed_loop(how, c=None, K=None):
        """W untimed + K timed steps of one sharding, bracketed by barrier + device synchronisation; the time is the MAX over
        the ranks.  Returns (seconds, per-step ms, infos, last result, its scalar set, per-rank summaries)."""
        shards = window_shards(K, world) if K else None
        for i in range(args.warmup):
            step(i, how, c, K)
        sync()
        t0 = time.perf_counter()
        infos, step_ms = [], []
        last, last_set = None, 0
        for i in range(args.steps):
            last_set = (args.warmup + i) % n_sets
            ts = time.perf_counter()
            last, info = step(args.warmup + i, how, c, K)    # msm_run / the all-gather are synchronous: per-step wall time is meaningful
            step_ms.append((time.perf_counter() - ts) * 1e3)
            infos.append(info)
        sync()
        dt = time.perf_counter() - t0
        ranks_info = None
        if sharded:
            tmax = torch.tensor([dt], dtype=torch.float64, device=ddev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
            # outside the timed region: what every rank spent where, so that a scaling run explains itself
            mine = [x for x in infos if x]
            summary = {"rank": rank, "shard": (list(point_shards(n, world)[rank]) if how == "points" else [rank, world] if how == "buckets" else list(shards[rank])),
                       "steps_with_work": len(mine)}
            if mine:
                summary["phase_ms"] = {k: sum(x["phase_ms"][k] for x in mine) / len(mine) for k in mine[0]["phase_ms"]}
                ag = [x["all_gather_ms"] for x in mine if x.get("all_gather_ms") is not None]
                summary["all_gather_ms"] = sum(ag) / len(ag) if ag else None
                summary["step_ms"] = sum(step_ms) / len(step_ms)
            gathered = [None] * world
            dist.all_gather_object(gathered, summary)
            ranks_info = gathered
        return dt, step_ms, infos, last, last_set, ranks_info

    def check(last, last_set):
        """the checker leg (outside every timed region): a result against the known discrete logs of the generated points"""
        _, s_host = ctx.generate_scalars(n, seed=1000 + last_set, to_host=True, raw=True)   # the same stream, read back
        exp = expected_from_logs(args.curve, a_host, s_host, n)
        return bool(last is not None and last.as_tuple() == exp)

    failed = None
    dt, step_ms, infos, last, last_set, ranks_info = timed_loop(split, c, K)
    other_splits = None
    if sharded and not args.no_other_splits:
        other_splits = []
        for other in [x for x in ("windows", "points", "buckets") if x != split]:
            o_c, o_K = plan_for(other)
            o_dt, o_step_ms, _, o_last, o_set, o_ranks = timed_loop(other, o_c, o_K)
            if rank == 0:
                other_splits.append({"split": other, "window_bits": o_c, "windows": o_K,
                                     "value": n * args.steps / o_dt, "unit": "points/s", "ms_per_step": o_dt / args.steps * 1e3,
                                     **step_stats(o_step_ms), "verified": check(o_last, o_set) if verify else None, "ranks": o_ranks})

    if rank == 0:
        infos = [x for x in infos if x]
        acc_ms = sum(x["phase_ms"]["accumulate"] for x in infos)
        # ALGORITHMIC pair additions price the roofline (sum over the non-empty buckets of size - 1: msm_result.n_pairs_algo);
        # the tree also issues the additions of its padding lanes (n_pairs, ~2.5 % more at 2^26)
        pairs = sum(x["n_pairs_algo"] for x in infos)
        pairs_issued = sum(x["n_pairs"] for x in infos)
        launches = sum(x["rounds"] for x in infos) or 1     # tree rounds of ALL window groups (summed by the library)
        phase = {k: sum(x["phase_ms"][k] for x in infos) / max(len(infos), 1) for k in infos[0]["phase_ms"]} if infos else {}
        achieved = pairs * PAIR_ALGO_BYTES / (acc_ms * 1e-3) / 1e9 if acc_ms else 0.0
        # ---- checker leg (outside the timed region): the last timed result against the known discrete logs -------
        verified = check(last, last_set) if verify else None
        # Big inputs run as two window groups on two streams, so the event-timed launch durations above are those of
        # kernels SHARING the GPU.  One extra, untimed step with the groups serialised gives the exclusive figures.
        excl = None
        pcie = None
        if not sharded:
            # the same kernel under the round-3 plan (c = 16, K = 8: a third more pair additions, none of them through the
            # chunk-ordered or descriptor paths) when the library picks a bigger window: the per-pair rate of the tree kernel
            # where nothing but the kernel itself is in the way
            c16 = None
            if c > 16 and not args.c and not args.no_c16:
                ctx.run_device(scal[0].data_ptr(), n, c=16, serial=True)   # untimed: another plan's buffers are allocated here
                _, yi = ctx.run_device(scal[0].data_ptr(), n, c=16, serial=True)
                y_ms, yp = yi["phase_ms"]["accumulate"], yi["n_pairs_algo"]
                c16 = {"window_bits": 16, "windows": yi["K"], "accumulate_ms": y_ms, "pair_adds": yp,
                       "int_mad_frac": yp * PAIR_MADS / (y_ms * 1e-3) / INT_MAD_PEAK, "ns_per_pair_add": y_ms * 1e6 / yp,
                       "note": "one serialised step at c = 16 (the plan of rounds 1-3): more pair additions, all of them in the "
                               "index-free rounds the kernel is fastest in; the shipped plan trades per-pair rate for 25 % fewer"}
            _, xi = ctx.run_device(scal[0].data_ptr(), n, c=c, serial=True)
            x_ms = xi["phase_ms"]["accumulate"]
            xp = xi["n_pairs_algo"]
            excl = {
                "same_kernel_at_c16": c16,
                "ns_per_pair_add": x_ms * 1e6 / xp,
                "accumulate_ms": x_ms,
                "pair_adds": xp,
                "pair_adds_issued": xi["n_pairs"],
                "achieved": xp * PAIR_ALGO_BYTES / (x_ms * 1e-3) / 1e9,
                "frac": xp * PAIR_ALGO_BYTES / (x_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "int_mad_frac": xp * PAIR_MADS / (x_ms * 1e-3) / INT_MAD_PEAK,
                "int_mad_frac_1800_basis": xp * PAIR_MADS_SURVEY / (x_ms * 1e-3) / INT_MAD_PEAK,
                "int_mad_frac_at_2_waves_per_simd": xp * PAIR_MADS / (x_ms * 1e-3) / INT_MAD_AT_2_WAVES,
                "int_mad_frac_at_held_clock": xp * PAIR_MADS / (x_ms * 1e-3) / (INT_MAD_PEAK * HELD_CLOCK_GHZ / UBENCH_CLOCK_GHZ),
                "held_clock_ghz": HELD_CLOCK_GHZ,
                "phase_ms": xi["phase_ms"],
                # the counting sort (histogram + scans + scatter): 2 N K entries, each read twice as a 4-byte digit and
                # written once as a 4-byte payload to a random slot of its bucket
                "scatter": {
                    "bound": "hbm",
                    "algorithmic_bytes_per_entry": SORT_ALGO_BYTES,
                    "entries": 2 * n * K,
                    "ms": xi["phase_ms"]["sort"],
                    "achieved": 2 * n * K * SORT_ALGO_BYTES / (xi["phase_ms"]["sort"] * 1e-3) / 1e9,
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": 2 * n * K * SORT_ALGO_BYTES / (xi["phase_ms"]["sort"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                },
            }
        if not sharded and not args.no_pcie:
            # The same MSM with the scalars handed over as a HOST buffer (pageable memory; 2^n x 32 bytes cross PCIe inside the
            # call, behind the computation from 2^24 points up): never `value`.  Same protocol as the headline: 15 calls, the
            # first 5 discarded, median and sample standard deviation; every result must equal the device-resident one.
            _, s_host = ctx.generate_scalars(n, seed=1000, to_host=True, raw=True)
            ref0, _ = ctx.run_device(scal[0].data_ptr(), n, c=c)
            p_ms, p_up = [], []
            for i in range(15):
                tp = time.perf_counter()
                got, pi = ctx.run(s_host, c=args.c or None)   # the library plans a host-scalar call itself (ranges of the points)
                if i >= 5:
                    p_ms.append((time.perf_counter() - tp) * 1e3)
                    p_up.append(pi["phase_ms"]["upload"])
                if got.as_tuple() != ref0.as_tuple():
                    failed = "bench: the MSM over host scalars differs from the one over the same scalars resident in HBM"
            del s_host
            pcie = {"median_ms": statistics.median(p_ms), "std_ms": statistics.stdev(p_ms), "min_ms": min(p_ms), "runs": len(p_ms),
                    "points_per_s": n / (statistics.median(p_ms) * 1e-3), "upload_ms": statistics.median(p_up),
                    "equals_device_resident": failed is None,
                    "note": "MSMs with host-resident (pageable) scalars: the upload runs in the background, range by range of the "
                            "points, under the window groups of the ranges that have arrived; upload_ms = wall time of the transfer"}
        tables_leg = None
        if not sharded and not args.no_tables_leg and not is381 and not infos[-1].get("tables"):
            # The same MSM on WINDOW TABLES (msm_precompute): K resident tables 2^(c k) P of the point set, all windows of a window
            # group in one set of buckets.  Off by default at this size (six tables of 2^26 points are 96 GiB for ~1.5 %); the leg
            # builds them once, times 10 calls after 3 discarded ones and checks the result against the plain path.
            ctx.set_tables_limit(1 << 42)
            tb = time.perf_counter()
            tc, tK, tbytes = ctx.precompute(n, c=c)
            build_s = time.perf_counter() - tb
            if tK:
                t_ms = []
                for i in range(13):
                    tp = time.perf_counter()
                    t_last, t_info = ctx.run_device(scal[i % n_sets].data_ptr(), n, c=c)
                    if i >= 3:
                        t_ms.append((time.perf_counter() - tp) * 1e3)
                p_last, _ = ctx.run_device(scal[12 % n_sets].data_ptr(), n, c=c, no_tables=True)
                tables_leg = {"median_ms": statistics.median(t_ms), "std_ms": statistics.stdev(t_ms), "points_per_s": n / (statistics.median(t_ms) * 1e-3),
                              "tables": tK, "window_bits": tc, "gib": tbytes / 2 ** 30, "build_s": build_s, "ran_on_tables": bool(t_info["tables"]),
                              "equals_plain_path": t_last.as_tuple() == p_last.as_tuple(),
                              "note": "one-time per point set, like the point conversion the reference's protocol also leaves out of the "
                                      "timed region (scripts/msm-weierstrass.ts:19,32); never `value` at this size"}
                if not tables_leg["equals_plain_path"]:
                    failed = "bench: the MSM on window tables differs from the plain path"
            ctx.set_tables_limit(0)
        skewed = None
        if not sharded and not is381 and not args.no_skewed:
            # Skewed scalar distributions (montgomery_amd/workloads.py) over the same resident points, outside `value`: a prover's
            # witness-shaped set (40 % zeros, 20 % ones, 10 % below 2^16, the rest uniform) and ONE scalar repeated (every entry of a
            # window in one bucket).  The reference walks any bucket-size distribution through the same rounds
            # (src/msm-batched-affine.ts:204,243-263); here the sort cuts heavy bins into parts (sort_kernels.h) and deep buckets
            # run through the tail rounds.  Five timed calls after a warm-up each, median, result checked against the known logs.
            from montgomery_amd import workloads

            uniform_ms = statistics.median(step_ms)
            skewed = []
            for kind in ("prover", "one"):
                s_np = workloads.scalars(kind, n, seed=4242)
                scal[0].copy_(torch.from_numpy(s_np.reshape(-1)))
                torch.cuda.synchronize()
                plain = not infos[-1].get("tables")   # the path the headline ran on (the tables leg above may have left tables behind)
                k_last, k_info = ctx.run_device(scal[0].data_ptr(), n, c=c_main, no_tables=plain)
                k_ms = []
                for _ in range(5):
                    tk = time.perf_counter()
                    k_last, k_info = ctx.run_device(scal[0].data_ptr(), n, c=c_main, no_tables=plain)
                    k_ms.append((time.perf_counter() - tk) * 1e3)
                k_ok = None
                if verify:
                    k_ok = bool(k_last.as_tuple() == expected_from_logs(args.curve, a_host, C_uint8_view(s_np), n))
                    if not k_ok:
                        failed = f"bench: the MSM over the '{kind}' scalar distribution failed the known-discrete-log check"
                skewed.append({"scalars": kind, "median_ms": statistics.median(k_ms), "min_ms": min(k_ms), "max_ms": max(k_ms),
                               "ratio_to_uniform": statistics.median(k_ms) / uniform_ms, "largest_bucket": k_info["max_bucket"],
                               "tree_rounds": k_info["rounds"], "pair_adds": k_info["n_pairs_algo"], "verified": k_ok})
                del s_np
            ctx.generate_scalars(n, seed=1000, into=scal[0].data_ptr())   # set 0 as the other legs know it
        pmc = pmc_summary(args.curve, args.log2n, c)
        if excl and pmc and pmc.get("scatter_phase"):
            excl["scatter"]["hbm_bytes_per_entry_pmc"] = pmc["scatter_phase"]["hbm_bytes_per_entry"]
            excl["scatter"]["digits_ms"] = excl["phase_ms"]["digits"]
        mad_rate = (excl["int_mad_frac"] * INT_MAD_PEAK) if excl else (pairs * PAIR_MADS / (acc_ms * 1e-3) if acc_ms else 0.0)
        out = {
            "metric": f"{'BLS12-381' if is381 else 'BLS12-377'} G1 MSM throughput",
            "value": n * args.steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.curve}-g1-msm-2^{args.log2n}",
                "log2_n": args.log2n,
                "window_bits": c,
                "windows": K,
                "parallelism": "single-gpu" if not sharded else (
                    f"window-shard x{world}, one RCCL all-gather of {K}x144 B" if split == "windows" else
                    f"bucket-shard x{world} (all {K} windows on 1/{world} of every window's buckets per rank), one RCCL all-gather of {K}x144 B per rank"
                    if split == "buckets" else
                    f"points-shard x{world} (all {K} windows on n/{world} points per rank), one RCCL all-gather of {K}x144 B per rank"),
                "points": "P_i = a_i*G generated on GPU (resident)",
                "scalars": f"uniform < q, fresh per step ({n_sets} distinct sets cycled), resident in HBM before the timed region",
            },
            "window_tables": bool(infos and infos[-1].get("tables")),
            "verified": verified,
            "verified_how": "last timed result == (sum s_i a_i mod q) G from the known discrete logs of the generated points "
                            "(dot product: oracle/msm_oracle.c, scalar multiplication: oracle/msm_oracle.py), outside the timed region",
            **step_stats(step_ms),
            "roofline": {
                # arithmetic intensity 1 872 MADs / 288 B = 6.5 per byte against a machine balance of 4.2: the binding roof
                # of the dominant kernel is the integer multiply-add issue rate, so that is what `frac` prices
                "kernel": "k_batch_add (bucket accumulation tree, all rounds)",
                "bound": "int-alu",
                "achieved": mad_rate,
                "peak": INT_MAD_PEAK,
                "unit": "v_mad_u64_u32 lane-ops/s",
                "frac": mad_rate / INT_MAD_PEAK,
                "frac_1800_basis": mad_rate / PAIR_MADS * PAIR_MADS_SURVEY / INT_MAD_PEAK,
                "frac_basis": ("exclusive (window groups serialised, one untimed step; a launch that has the chip to itself walks its "
                               "pairs in batches of 128 per lane, the overlapped run `value` is timed on in batches of 512: "
                               "`int_mad_overlapped` is the figure of that geometry)" if excl else "overlapped streams")
                              + "; algorithmic pair additions (sum over non-empty buckets of size - 1) x 1 872 multiply-adds "
                                "(13-limb count); frac_1800_basis prices them at SURVEY section 8(d)'s 300 per multiplication",
                "mads_per_pair_add": PAIR_MADS,
                "traffic": (pairs / launches * pmc["per_pair_addition"]["all_rounds"]["hbm_bytes_per_pair_add"]) if pmc else None,
                "traffic_note": ("HBM bytes per launch = algorithmic pair additions per launch x the measured bytes per pair addition of "
                                 + os.path.relpath(PMC_FILE, ROOT) + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this "
                                 "command at this size, curve and window; not collected inside this run)") if pmc else
                                "null: the committed PMC summary was taken at another size, curve or window (profiles/r06_pmc_2p26.json: "
                                "BLS12-377, 2^26, 21-bit windows)",
                "traffic_bytes_per_pair_add": pmc["per_pair_addition"]["all_rounds"]["hbm_bytes_per_pair_add"] if pmc else None,
                "pair_adds_per_step": pairs / max(len(infos), 1),
                "pair_adds_issued_per_step": pairs_issued / max(len(infos), 1),
                "avg_launch_ms": acc_ms / launches,
                "launches_per_step": launches / max(len(infos), 1),
                "hbm": {
                    "achieved": achieved,
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": pairs / launches * PAIR_ALGO_BYTES,
                    "algorithmic_bytes_per_pair_add": PAIR_ALGO_BYTES,
                },
                "int_mad_overlapped": {
                    "achieved": pairs * PAIR_MADS / (acc_ms * 1e-3) if acc_ms else 0.0,
                    "frac": (pairs * PAIR_MADS / (acc_ms * 1e-3) / INT_MAD_PEAK) if acc_ms else 0.0,
                    "note": "event-timed launch durations of the two window-group streams, which share the chip: per-launch rates",
                },
                "exclusive": excl,
                "note": "achieved = pair additions x 1 872 multiply-adds / event-timed accumulation time with the two window-group "
                        "streams serialised ('exclusive'); hbm = the same with 288 algorithmic bytes per pair addition "
                        "(DESIGN.md section 5)",
            },
            "phase_ms": phase,
            "split": split,
            "ranks": ranks_info,
            "other_splits": other_splits,
            "pcie_inclusive": pcie,
            "window_tables_leg": tables_leg,
            "skewed": skewed,
            "result_is_infinity": bool(last.isZero) if last is not None else None,
        }
        if not sharded and not args.no_cpu_baseline and not is381:
            out["cpu_baseline"] = cpu_baseline(ctx, min(args.cpu_log2n, args.log2n), seed=777,
                                           log2n_headline=None if args.no_cpu_headline else args.log2n)
        if not sharded and not is381 and not args.no_other_configs and args.log2n != 20:
            # the other size BASELINE.json's metric names and configs[3], timed by the same process (value stays the headline size)
            out["other_configs"] = [timed_config("bls12-377", 20, torch), timed_config("ed377", 20, torch)]
        print(json.dumps(out), flush=True)
        if verified is False or any(o["verified"] is False for o in (other_splits or [])):
            failed = "bench: the MSM result failed the known-discrete-log check"
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if failed:
        sys.exit(failed)


if __name__ == "__main__":
    main()
