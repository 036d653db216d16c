#!/usr/bin/env python3
"""Headline benchmark: BLS12-377 G1 MSM throughput (points/s) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2n 26] [--c C]

One "step" = one MSM over resident base points with FRESH scalars already in HBM (the reference's
protocol: points pre-loaded, new random scalars every run, scripts/msm-weierstrass.ts:12-51).
N = 1: the whole MSM on one GPU.  N > 1 (launched by torch.distributed.run, one rank per GPU): the
same MSM sharded by scalar window -- every rank holds all points and scalars, computes the window
sums P_k of its windows, ONE RCCL all-gather of K x 144 bytes, rank 0 does the Horner combination
(SURVEY.md section 8e).  Total work is fixed as N grows: scaling = "strong".

Prints ONE JSON line on rank 0 with the driver's contract plus `roofline` (dominant kernel
k_batch_add, HIP-event timed inside the library on its own stream) and `cpu_baseline` (the C port
of the oracle on the host cores, bounded sample).
"""
import argparse
import json
import os
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
INT_MAD_PEAK = 30.3e12         # v_mad_u64_u32 lane-ops/s measured on MI355X by tools/ubench_int.hip (profiles/r01_ubench_int.txt)
# HBM bytes per pair addition of the regular tree rounds from the PMC passes committed in profiles/r01_pmc_2p24.json
# (separate --pmc FETCH_SIZE / WRITE_SIZE runs at 2^24): FETCH_SIZE 169 B x 2 (gfx950 halves wide coalesced reads)
# + WRITE_SIZE 147 B.  Round 1 (random 64-byte-sector gathers) reads 381 B uncorrected + writes 148 B per pair.
PAIR_TRAFFIC_BYTES_PMC = 2 * 169 + 147


def cpu_baseline(ctx, log2n_sample, seed):
    """Times oracle/msm_oracle.c (kind "port") on the host cores over the first 2^log2n_sample
    resident points.  The reference's WASM path cannot run here."""
    from oracle import c_oracle

    n = 1 << log2n_sample
    pts = ctx.get_points(0, n)
    _, sc = ctx.generate_scalars(n, seed=seed, to_host=True)
    c_oracle.load()
    t0 = time.perf_counter()
    ref, threads = c_oracle.msm_bls377(pts, sc, 0)
    dt = time.perf_counter() - t0
    dev, _ = ctx.generate_scalars(n, seed=seed)
    got, _ = ctx.run_device(dev, n)
    assert got.as_tuple() == ref, "GPU result differs from the CPU oracle on the cpu_baseline sample"
    return {
        "value": n / dt,
        "unit": "points/s",
        "cores": threads,
        "kind": "port",
        "sample": f"one 2^{log2n_sample}-point BLS12-377 G1 MSM (first 2^{log2n_sample} of the resident points, reference window table), "
                  f"{dt:.2f} s, OpenMP over windows; GPU result on the same inputs checked equal",
    }


def bench_ed377(args, torch):
    """BASELINE configs[3]: 2^20 Ed-on-BLS12-377 MSM (msmBasic path) on one GPU.  Points: N distinct subgroup points
    P_i = a_i G generated on the GPU (resident); scalars: uniform < q, fresh per step, resident in HBM."""
    from montgomery_amd import _lib
    from montgomery_amd.api import MsmContext

    n = 1 << args.log2n
    ctx = MsmContext(_lib.CURVE_ED_ON_BLS12_377, device=0)
    ctx.generate_points(n, seed=20261002)
    c, K = ctx.plan(n, args.c or None)
    dev = torch.device("cuda", 0)
    scal = [torch.empty(n * 32, dtype=torch.uint8, device=dev) for _ in range(args.steps + args.warmup)]
    for i, t in enumerate(scal):
        ctx.generate_scalars(n, seed=1000 + i, into=t.data_ptr())
    torch.cuda.synchronize()
    for i in range(args.warmup):
        ctx.run_device(scal[i].data_ptr(), n, c=c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    infos = []
    for i in range(args.steps):
        _, info = ctx.run_device(scal[args.warmup + i].data_ptr(), n, c=c)
        infos.append(info)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    acc_ms = sum(x["phase_ms"]["accumulate"] for x in infos)
    pairs = sum(x["n_pairs"] for x in infos)
    # one unified extended addition: two 128-byte nodes in, one out; 9 multiplications of 9 limbs (2*81 - 9 MADs)
    algo_bytes, mads = 384, 9 * 153
    achieved = pairs * algo_bytes / (acc_ms * 1e-3) / 1e9
    out = {
        "metric": "Ed-on-BLS12-377 MSM throughput", "value": n * args.steps / dt, "unit": "points/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"ed-on-bls12-377-msm-2^{args.log2n}", "log2_n": args.log2n, "window_bits": c, "windows": K,
                   "parallelism": "single-gpu"},
        "roofline": {"kernel": "k_te_add (bucket tree, unified extended additions)", "bound": "hbm", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_pair_add": algo_bytes,
                     "int_mad": {"achieved": pairs * mads / (acc_ms * 1e-3), "peak": INT_MAD_PEAK,
                                 "frac": pairs * mads / (acc_ms * 1e-3) / INT_MAD_PEAK}},
        "phase_ms": {k: sum(x["phase_ms"][k] for x in infos) / len(infos) for k in infos[0]["phase_ms"]},
    }
    print(json.dumps(out), flush=True)
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=26)
    ap.add_argument("--c", type=int, default=0)
    ap.add_argument("--cpu-log2n", type=int, default=22)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (functional check of the sharded path on one GPU)")
    ap.add_argument("--curve", choices=["bls12-377", "bls12-381", "ed377"], default="bls12-377",
                    help="ed377 = BASELINE configs[3]: twisted Edwards msmBasic path (single GPU, use --log2n 20); "
                         "bls12-381 = the same batched-affine path over the BLS12-381 G1 constants (no CPU baseline leg)")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if local_rank >= torch.cuda.device_count():   # several ranks on one GPU: only meaningful with --dist-backend gloo
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    from montgomery_amd.api import AffineResult, MsmContext
    from montgomery_amd.distributed import sharded_msm, window_shards

    n = 1 << args.log2n
    if args.curve == "ed377":
        return bench_ed377(args, torch)
    from montgomery_amd import _lib as _abi

    is381 = args.curve == "bls12-381"
    ctx = MsmContext(_abi.CURVE_BLS12_381_G1 if is381 else _abi.CURVE_BLS12_377_G1, device=local_rank)
    ctx.generate_points(n, seed=20261002)   # identical on every rank
    c, K = ctx.plan(n, args.c or None)
    shards = window_shards(K, world)

    dev = torch.device("cuda", local_rank)
    n_bufs = args.steps + args.warmup
    # fresh scalars per step, generated on the GPU before the timed region (resident in HBM)
    scal = [torch.empty(n * 32, dtype=torch.uint8, device=dev) for _ in range(n_bufs)]
    for i, t in enumerate(scal):
        ctx.generate_scalars(n, seed=1000 + i, into=t.data_ptr())
    def step(i):
        if world == 1:
            return ctx.run_device(scal[i].data_ptr(), n, c=c)
        box = {}

        def my_window_sums(lo, hi):
            parts, box["info"] = ctx.window_sums(scal[i].data_ptr(), n, lo, hi, c=c, on_device=True)
            return parts

        out = sharded_msm(my_window_sums, K, c, device=dev if args.dist_backend == "nccl" else "cpu", curve=ctx.curve)
        res = None
        if out is not None:
            xy = out[1]
            res = AffineResult(x=xy[0] if xy else 0, y=xy[1] if xy else 0, isZero=xy is None)
        return res, box.get("info")

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    infos = []
    last = None
    for i in range(args.steps):
        last, info = step(args.warmup + i)
        infos.append(info)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        infos = [x for x in infos if x]
        acc_ms = sum(x["phase_ms"]["accumulate"] for x in infos)
        pairs = sum(x["n_pairs"] for x in infos)
        launches = sum(x["rounds"] for x in infos) or 1
        phase = {k: sum(x["phase_ms"][k] for x in infos) / max(len(infos), 1) for k in infos[0]["phase_ms"]} if infos else {}
        achieved = pairs * PAIR_ALGO_BYTES / (acc_ms * 1e-3) / 1e9 if acc_ms else 0.0
        # Big inputs run as two window groups on two streams, so the event-timed launch durations above are those of
        # kernels SHARING the GPU.  One extra, untimed step with the groups serialised gives the exclusive figures.
        excl = None
        if world == 1:
            _, xi = ctx.run_device(scal[0].data_ptr(), n, c=c, serial=True)
            x_ms = xi["phase_ms"]["accumulate"]
            excl = {
                "accumulate_ms": x_ms,
                "achieved": xi["n_pairs"] * PAIR_ALGO_BYTES / (x_ms * 1e-3) / 1e9,
                "frac": xi["n_pairs"] * PAIR_ALGO_BYTES / (x_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "int_mad_frac": xi["n_pairs"] * PAIR_MADS / (x_ms * 1e-3) / INT_MAD_PEAK,
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
                    "note": "random 4-byte stores: every payload write dirties its own 64-byte sector, so the algorithmic "
                            "rate understates the sectors moved; hidden under the accumulation of the other window group",
                },
            }
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
                "parallelism": "single-gpu" if world == 1 else f"window-shard x{world}, one RCCL all-gather of {K}x144 B",
                "points": "P_i = a_i*G generated on GPU (resident)",
                "scalars": "uniform < q, fresh per step, resident in HBM before the timed region",
            },
            "roofline": {
                "kernel": "k_batch_add (bucket accumulation tree, all rounds)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pairs / launches * PAIR_TRAFFIC_BYTES_PMC,
                "traffic_note": "bytes per launch = pair additions per launch x 485 B/pair from the committed PMC passes "
                                "(profiles/r01_pmc_2p24.json, regular rounds); not collected inside this run",
                "algorithmic_bytes_per_launch": pairs / launches * PAIR_ALGO_BYTES,
                "algorithmic_bytes_per_pair_add": PAIR_ALGO_BYTES,
                "pair_adds_per_step": pairs / max(len(infos), 1),
                "avg_launch_ms": acc_ms / launches,
                "launches_per_step": launches / max(len(infos), 1),
                "int_mad": {
                    "achieved": pairs * PAIR_MADS / (acc_ms * 1e-3) if acc_ms else 0.0,
                    "peak": INT_MAD_PEAK,
                    "unit": "v_mad_u64_u32 lane-ops/s",
                    "frac": (pairs * PAIR_MADS / (acc_ms * 1e-3) / INT_MAD_PEAK) if acc_ms else 0.0,
                },
                "exclusive": excl,
                "note": "the kernel is integer-ALU bound (no MFMA path exists for carry-propagated big integers); 'hbm' is the "
                        "nearer of the two allowed labels, int_mad carries the ALU roofline; 'exclusive' = same kernels with the "
                        "two window-group streams serialised (one untimed step)",
            },
            "phase_ms": phase,
            "result_is_infinity": bool(last.isZero) if last is not None else None,
        }
        if world == 1 and not args.no_cpu_baseline and not is381:
            out["cpu_baseline"] = cpu_baseline(ctx, min(args.cpu_log2n, args.log2n), seed=777)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
