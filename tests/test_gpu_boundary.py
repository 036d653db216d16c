"""Boundary behaviour of the C ABI / facade that the reference has by construction and round 1 lacked:
independent pointer handles (every `getPointer` / `randomScalars` of the reference is its own allocation,
src/parallel.ts:97-133, src/curve-random.ts:151-194), a distinct status for out-of-range scalars, and the
device-list context of SURVEY.md section 8(b).  Needs an MI355X: `-m gpu`."""
import pytest

from oracle import msm_oracle as O

pytestmark = pytest.mark.gpu

C = O.BLS12_377
G = (C.gx, C.gy)


def test_scalar_handles_are_independent_allocations():
    """a = randomScalars(n); b = randomScalars(n); msm(a) must use a's data (round 1 computed with b's), also after
    a bigger handle was made and dropped in between."""
    from montgomery_amd.api import BLS12_377_PARAMS, Weierstrass

    cv = Weierstrass.create(BLS12_377_PARAMS)
    par, ctx = cv.Parallel, cv.context
    n = 3000
    pp = par.randomPointsFast(n, seed=5)
    a_logs = O.scalars_from_bytes(ctx.generate_points(n, seed=5, want_scalars=True))   # same seed: the same points
    a = par.randomScalars(n, seed=101)
    b = par.randomScalars(n, seed=202)
    assert a.dev_ptr != b.dev_ptr
    big = par.randomScalars(50 * n, seed=303)     # grows nothing that a or b point into
    big.close()
    _, sa = ctx.generate_scalars(n, seed=101, to_host=True)
    _, sb = ctx.generate_scalars(n, seed=202, to_host=True)
    ea = O.aff_scale(sum(x * y for x, y in zip(a_logs, O.scalars_from_bytes(sa))) % C.q, G, C.p)
    eb = O.aff_scale(sum(x * y for x, y in zip(a_logs, O.scalars_from_bytes(sb))) % C.q, G, C.p)
    assert ea != eb
    assert par.msm(a, pp, n)["result"].as_tuple() == ea
    assert par.msm(b, pp, n)["result"].as_tuple() == eb
    assert par.msm(a, pp, n)["result"].as_tuple() == ea
    a.close(); b.close(); pp.close()
    cv.context.close()


def test_point_handles_coexist():
    """Two `pointPtr`s of one curve object stay valid side by side; msm() uses the one it is given."""
    from montgomery_amd import MsmError
    from montgomery_amd.api import BLS12_377_PARAMS, Weierstrass

    cv = Weierstrass.create(BLS12_377_PARAMS)
    par = cv.Parallel
    p1, _ = O.random_points_bls377("bnd/p1", 40)
    p2, _ = O.random_points_bls377("bnd/p2", 25)
    sc = O.prng_ints("bnd/s", 40, C.q)
    pp1, pp2 = par.getPointer(40 * 96), par.getPointer(25 * 96)
    assert pp1.set_id != pp2.set_id
    par.pointsFromBytes(pp1, O.points_to_bytes(p1, 48), 40)
    par.pointsFromBytes(pp2, O.points_to_bytes(p2, 48), 25)     # must not replace the points behind pp1
    sp = par.getScalarPointer(40 * 32)
    par.scalarsFromBytes(sp, O.scalars_to_bytes(sc), 40)
    assert par.msm(sp, pp1, 40)["result"].as_tuple() == O.msm_naive_affine(sc, p1, C)
    assert par.msm(sp, pp2, 25)["result"].as_tuple() == O.msm_naive_affine(sc[:25], p2, C)
    assert par.msm(sp, pp1, 40)["result"].as_tuple() == O.msm_naive_affine(sc, p1, C)
    with pytest.raises(MsmError):
        par.msm(sp, pp2, 40)           # only 25 points behind this pointer
    pp2.close()
    assert par.msm(sp, pp1, 40)["result"].as_tuple() == O.msm_naive_affine(sc, p1, C)
    cv.context.close()


def test_strict_scalars_status(gpu_ctx):
    """Scalars >= q: reduced mod q by default (same group element), MSM_ERR_SCALAR under msm_opts.strict."""
    import ctypes as CT

    from montgomery_amd import MsmError, _lib
    from montgomery_amd._lib import MsmOpts, MsmResult

    pts, _ = O.random_points_bls377("bnd/strict", 20)
    sc = O.prng_ints("bnd/strict/s", 20, C.q)
    sc[7] += C.q                                    # still < 2^256
    gpu_ctx.set_points(O.points_to_bytes(pts, 48))
    res, _ = gpu_ctx.run(O.scalars_to_bytes(sc))
    assert res.as_tuple() == O.msm_naive_affine([s % C.q for s in sc], pts, C)
    buf = (CT.c_uint8 * (32 * 20)).from_buffer_copy(O.scalars_to_bytes(sc))
    out = MsmResult()
    rc = gpu_ctx._lib.msm_run(gpu_ctx._h, buf, 20, 0, CT.byref(MsmOpts(strict=1)), CT.byref(out))
    assert rc == _lib.MSM_ERR_SCALAR
    assert b"group order" in gpu_ctx._lib.msm_last_error(gpu_ctx._h)
    sc[7] -= C.q
    buf = (CT.c_uint8 * (32 * 20)).from_buffer_copy(O.scalars_to_bytes(sc))
    assert gpu_ctx._lib.msm_run(gpu_ctx._h, buf, 20, 0, CT.byref(MsmOpts(strict=1)), CT.byref(out)) == _lib.MSM_OK
    with pytest.raises(MsmError):
        gpu_ctx.window_sums(O.scalars_to_bytes(sc), 20, 0, 0)      # (0, 0) is no longer "all windows" on the Python side


@pytest.mark.parametrize("lg", [8, 16])
def test_device_list_context_on_one_gpu(gpu_ctx, lg):
    """msm_ctx_create_multi with the device list [0, 0, 0]: three device contexts (here on the same GPU) driven by host
    threads inside the library -- by points (default: every device runs all windows on a third of the points and gets only
    that third of the scalars) and by window (3 / 3 / 2 windows each over all points); the sums are combined on the host.
    Host, device and pre-placed scalars, MSM and window sums, against the single-device context on the same inputs."""
    from montgomery_amd.api import MsmContext

    n = 1 << lg
    multi = MsmContext(devices=[0, 0, 0])
    assert multi.n_devices == 3
    a = O.scalars_from_bytes(multi.generate_points(n, seed=77, want_scalars=True))
    gpu_ctx.generate_points(n, seed=77)
    dev, sb = multi.generate_scalars(n, seed=78, to_host=True)
    single, info1 = gpu_ctx.run(sb, no_tables=True)   # the plain plan: the one a device list shards (window tables are per context)
    exp = O.aff_scale(sum(x * y for x, y in zip(a, O.scalars_from_bytes(sb))) % C.q, G, C.p)
    assert single.as_tuple() == exp and gpu_ctx.run(sb)[0].as_tuple() == exp
    r_host, info = multi.run(sb)
    assert r_host.as_tuple() == exp and info["K"] == info1["K"]
    r_win, info_w = multi.run(sb, by_window=True)
    assert r_win.as_tuple() == exp
    assert info_w["n_pairs"] == info1["n_pairs"] and info_w["K"] == info1["K"]   # the same windows, only on three contexts
    r_dev, _ = multi.run_device(dev, n)
    assert r_dev.as_tuple() == exp
    assert multi.run_device(dev, n, by_window=True)[0].as_tuple() == exp
    # scalars placed by the caller: one device buffer per device with that device's share of the scalars
    shares = [(n * d // 3, n * (d + 1) // 3) for d in range(3)]
    placed = []
    for lo, hi in shares:
        p = multi.device_alloc(max(32 * (hi - lo), 32))
        multi.device_upload(p, sb[32 * lo:32 * hi] or b"\0" * 32)
        placed.append(p)
    assert multi.run_placed(placed, n)[0].as_tuple() == exp
    for p in placed:
        multi.device_free(p)
    K, c = info["K"], info["c"]
    parts, _ = multi.window_sums(sb, n, 0, K)
    assert multi.combine(parts, K, c).as_tuple() == exp
    parts, _ = multi.window_sums(sb, n, 0, K, by_window=True)
    assert multi.combine(parts, K, c).as_tuple() == exp
    # a points shard through the single-device context: windows over [first, first + count) only, groups added by msm_combine_groups
    from montgomery_amd.distributed import combine_groups_host

    groups = b"".join(gpu_ctx.window_sums(sb[32 * lo:32 * hi], hi - lo, 0, K, c=c, point_lo=lo)[0] for lo, hi in shares)
    assert combine_groups_host(groups, 3, K, c) == exp
    # a second point set on the multi context, then back
    sid = multi.pointset_create()
    pts, _ = O.random_points_bls377("bnd/multi", 9)
    multi.set_points(O.points_to_bytes(pts, 48))
    sc = O.prng_ints("bnd/multi/s", 9, C.q)
    assert multi.run(O.scalars_to_bytes(sc))[0].as_tuple() == O.msm_naive_affine(sc, pts, C)
    multi.pointset_select(0)
    assert multi.run(sb)[0].as_tuple() == exp
    multi.pointset_destroy(sid)
    multi.close()


def test_curve_facade_takes_a_device_list():
    """`Weierstrass.create(params, devices=[...])` / `TwistedEdwards.create(params, devices=[...])`: the
    reference-shaped curve object over a device-list context, here [0, 0]."""
    from montgomery_amd import api

    cv = api.Weierstrass.create(api.BLS12_377_PARAMS, devices=[0, 0])
    assert cv.context.n_devices == 2
    pts, _ = O.random_points_bls377("bnd/facade", 33)
    sc = O.prng_ints("bnd/facade/s", 33, C.q)
    with cv.Parallel.getPointer(33 * 96) as pp, cv.Parallel.getScalarPointer(33 * 32) as sp:
        cv.Parallel.pointsFromBytes(pp, O.points_to_bytes(pts, 48), 33)
        cv.Parallel.scalarsFromBytes(sp, O.scalars_to_bytes(sc), 33)
        res = cv.Parallel.msm(sp, pp, 33)["result"]
    assert (res.x, res.y) == O.msm_naive_affine(sc, pts, C)
    cv.context.close()
    E = O.ED_ON_BLS12_377
    te = api.TwistedEdwards.create(api.ED_ON_BLS12_377_PARAMS, devices=[0, 0])
    tp, _ = O.random_points_ed377("bnd/facade/ed", 21)
    ts = O.prng_ints("bnd/facade/ed/s", 21, E.q)
    with te.Parallel.getPointer(21 * 64) as pp, te.Parallel.getScalarPointer(21 * 32) as sp:
        te.Parallel.pointsFromBytes(pp, O.points_to_bytes(tp, 32), 21)
        te.Parallel.scalarsFromBytes(sp, O.scalars_to_bytes(ts), 21)
        res = te.Parallel.msm(sp, pp, 21)["result"]
    assert (res.x, res.y) == O.msm_basic_te(ts, tp)
    te.context.close()


def _sharded_bench(world, log2n, split="windows"):
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--log2n", str(log2n), "--dist-backend", "gloo", "--no-cpu-baseline", "--split", split]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("world", [2, 8])
def test_points_shards_on_one_gpu(world):
    """bench.py --split points: every rank runs all windows on its share of the points (its scalars only), one all-gather,
    rank 0 adds the groups per window (msm_combine_groups) and verifies the result against the discrete logs; the line
    carries what every rank spent where."""
    d = _sharded_bench(world, 18, "points")
    assert d["n_gpus"] == world and d["verified"] is True and d["other_splits"][0]["split"] == "windows", d
    assert f"points-shard x{world}" in d["config"]["parallelism"]
    assert len(d["ranks"]) == world and all(r["shard"][1] == (1 << 18) // world for r in d["ranks"])
    assert all("phase_ms" in r and r["all_gather_ms"] is not None for r in d["ranks"])


@pytest.mark.parametrize("world", [4, 8])
def test_four_and_eight_rank_window_shards_on_one_gpu(world):
    """The driver's SCALE shapes (--gpus 4, --gpus 8: two windows / one window per rank at c = 16, K = 8), all ranks on the
    one GPU of the box with gloo as the process group; rank 0 verifies the combined result against the discrete logs."""
    d = _sharded_bench(world, 18)
    assert d["n_gpus"] == world and d["verified"] is True, d
    assert f"window-shard x{world}" in d["config"]["parallelism"]


def test_two_rank_bench_launches_itself():
    """`python bench.py --gpus 2` with NO launcher: the script starts its two ranks itself (torch.distributed.run children,
    before it has imported torch), here with both ranks on the one GPU of the box and gloo as the process group.  The headline
    is the window split of BASELINE configs[4]; the points split is timed by the same run (`other_splits`); both results are
    verified against the known discrete logs, and every rank reports its phases and its all-gather."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log2n", "20",
           "--dist-backend", "gloo", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["verified"] is True, d
    assert d["config"]["workload"] == "bls12-377-g1-msm-2^20" and "window-shard x2" in d["config"]["parallelism"]
    assert d["split"] == "windows" and [r["shard"] for r in d["ranks"]] == [[0, 4], [4, 8]]
    assert all("phase_ms" in r and r["all_gather_ms"] is not None for r in d["ranks"])
    others = {o["split"]: o for o in d["other_splits"]}
    assert set(others) == {"points", "buckets"} and all(o["verified"] is True and o["value"] > 0 for o in others.values())
    assert [r["shard"] for r in others["points"]["ranks"]] == [[0, 1 << 19], [1 << 19, 1 << 19]]
    assert [r["shard"] for r in others["buckets"]["ranks"]] == [[0, 2], [1, 2]]   # rank g of G: its part of every window's buckets


def test_bench_refuses_a_world_size_other_than_gpus():
    """Under a launcher with WORLD_SIZE != --gpus bench.py must fail, not print a line labelled with the wrong GPU count."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--log2n", "16"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("n", [(1 << 24) + 777, 1 << 25])
def test_host_scalars_pipelined_behind_the_msm(n):
    """From 2^24 points a host scalar buffer crosses PCIe BEHIND the computation: the MSM runs range by range of the points
    (three ranges at 2^24, four from 2^25; ragged sizes included) as their scalars arrive, the sums of the ranges are added
    per window.  Same group element as over the same scalars resident in HBM, for the default window, a forced one and a
    window shard, and against the known discrete logs of the generated points."""
    from montgomery_amd.api import MsmContext
    from oracle import c_oracle

    ctx = MsmContext()
    logs = ctx.generate_points(n, seed=77, want_scalars=True)
    dev, host = ctx.generate_scalars(n, seed=78, to_host=True)
    on_dev, _ = ctx.run_device(dev, n)
    on_host, info = ctx.run(host)
    assert on_host.as_tuple() == on_dev.as_tuple()
    assert info["phase_ms"]["upload"] > 0 and info["n_pairs_algo"] > 0
    k = c_oracle.dot_mod(logs, host, n, C.q)
    assert on_dev.as_tuple() == O.aff_scale(k, (C.gx, C.gy), C.p)
    assert ctx.run(host, c=13)[0].as_tuple() == on_dev.as_tuple()
    a, _ = ctx.window_sums(host, n, 2, 5)            # host scalars through the sharded entry
    b, _ = ctx.window_sums(dev, n, 2, 5, on_device=True)
    K = ctx.plan(n)[1]
    from montgomery_amd.distributed import combine_host

    ident = bytes(144)
    assert combine_host(ident * 2 + a + ident * (K - 5), K, ctx.plan(n)[0]) == combine_host(ident * 2 + b + ident * (K - 5), K, ctx.plan(n)[0])
    ctx.close()


def test_big_host_buffers_cross_in_staged_chunks():
    """Host scalars and wire points above 64 MB go through the library's pinned staging chunks (16 MB each, several host
    threads, the last chunk ragged): the MSM over host scalars must equal the one over the same scalars resident on the
    device, and points uploaded from the host must read back byte for byte."""
    from montgomery_amd.api import MsmContext

    n = (1 << 21) + 12345                       # 67.5 MB of scalars, 202 MB of wire points
    ctx = MsmContext()
    logs = ctx.generate_points(n, seed=31, want_scalars=True)    # P_i = a_i G; the a_i as 32-byte integers
    dev, host = ctx.generate_scalars(n, seed=32, to_host=True)
    on_dev, _ = ctx.run_device(dev, n)
    on_host, _ = ctx.run(host)
    assert on_host.as_tuple() == on_dev.as_tuple()
    from oracle import c_oracle

    k = c_oracle.dot_mod(logs, host, n, C.q)
    assert on_dev.as_tuple() == O.aff_scale(k, G, C.p)
    wire = ctx.get_points(0, n)
    other = MsmContext()
    other.set_points(wire)
    assert other.get_points(0, n) == wire
    other_host, _ = other.run(host)
    assert other_host.as_tuple() == on_dev.as_tuple()
    other.close(); ctx.close()


@pytest.mark.parametrize("curve", ["bls12-377", "ed377"])
def test_workspace_limit_runs_windows_over_point_ranges(curve):
    """msm_set_workspace_limit: with room for a third of a window's buffers every window runs over ranges of the points, the
    sums of the ranges added on the host -- the path 2^29 points take on their own (tools/huge_check.py).  Same result as
    without a limit, more tree rounds."""
    from montgomery_amd import _lib
    from montgomery_amd.api import MsmContext

    te = curve == "ed377"
    ctx = MsmContext(_lib.CURVE_ED_ON_BLS12_377 if te else _lib.CURVE_BLS12_377_G1)
    n = (1 << 18) + 321
    ctx.generate_points(n, seed=61)
    dev, _ = ctx.generate_scalars(n, seed=62)
    c = 12 if te else 13
    ref, info0 = ctx.run_device(dev, n, c=c)
    ctx.set_workspace_limit(40 << 20 if te else 60 << 20)
    got, info1 = ctx.run_device(dev, n, c=c)
    assert got.as_tuple() == ref.as_tuple()
    assert info1["rounds"] > info0["rounds"], (info0["rounds"], info1["rounds"])
    K = info0["K"]
    part, _ = ctx.window_sums(dev, n, 0, K, c=c, on_device=True)   # the shard entry takes the same path
    assert ctx.combine(part, K, c).as_tuple() == ref.as_tuple()
    ctx.set_workspace_limit(0)
    again, info2 = ctx.run_device(dev, n, c=c)
    assert again.as_tuple() == ref.as_tuple() and info2["rounds"] == info0["rounds"]
    ctx.close()


def test_shard_exchange_over_rccl_on_one_rank():
    """The sharded path with its REAL backend: torch.distributed "nccl" (= RCCL) with a world of one rank on the box's GPU --
    process-group initialisation with a device id, the pinned row -> device copy, `all_gather_into_tensor` on device tensors,
    the event-timed collective and the one blocking copy back, for both shardings.  (Two ranks cannot share one GPU under
    RCCL; the multi-rank form runs over gloo above and on the driver's multi-GPU node.)  Fresh child process."""
    import json
    import os
    import socket
    import subprocess
    import sys
    import textwrap

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = textwrap.dedent(f"""
        import json, os, sys
        sys.path.insert(0, {root!r})
        import torch, torch.distributed as dist
        from montgomery_amd.api import MsmContext
        from montgomery_amd.distributed import PARTIAL_BYTES, ShardExchange, sharded_msm, sharded_msm_points
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        n = 1 << 18
        ctx = MsmContext(device=0)
        ctx.generate_points(n, seed=5)
        dev, _ = ctx.generate_scalars(n, seed=6)
        want, info = ctx.run_device(dev, n)
        c, K = info["c"], info["K"]
        ex = ShardExchange(PARTIAL_BYTES * K, torch.device("cuda", 0))
        out = {{}}
        for name in ("windows", "points"):
            tm = {{}}
            if name == "windows":
                r = sharded_msm(lambda lo, hi: ctx.window_sums(dev, n, lo, hi, c=c, on_device=True)[0], K, c, device="cuda:0",
                                curve=ctx.curve, timing=tm, exchange=ex)
            else:
                r = sharded_msm_points(lambda first, cnt: ctx.window_sums(dev + 32 * first, cnt, 0, K, c=c, on_device=True, point_lo=first)[0],
                                       n, K, c, device="cuda:0", curve=ctx.curve, timing=tm, exchange=ex)
            out[name] = {{"ok": r[1] == want.as_tuple(), "all_gather_ms": tm["all_gather_ms"]}}
        out["backend"] = dist.get_backend()
        dist.barrier()
        dist.destroy_process_group()
        ctx.close()
        print(json.dumps(out))
    """)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["windows"]["ok"] and d["points"]["ok"], d
    assert d["windows"]["all_gather_ms"] >= 0 and d["points"]["all_gather_ms"] >= 0


def test_pipelined_host_scalars_error_paths():
    """Failures while host scalars are still crossing PCIe behind the computation: a scalar >= q under msm_opts.strict in the
    LAST range of the points (the call fails with MSM_ERR_SCALAR after all ranges ran), and a workspace limit that forces the
    plain staged upload instead.  The context stays usable and the next call gives the right element."""
    import ctypes as CT

    from montgomery_amd import _lib
    from montgomery_amd._lib import MsmOpts, MsmResult
    from montgomery_amd.api import MsmContext

    n = 1 << 24
    ctx = MsmContext()
    ctx.generate_points(n, seed=91)
    dev, host = ctx.generate_scalars(n, seed=92, to_host=True, raw=True)
    want, _ = ctx.run_device(dev, n)
    bad = (CT.c_uint8 * (32 * n)).from_buffer_copy(host)
    for j in range(32):
        bad[32 * (n - 5) + j] = 0xFF     # 2^256 - 1 >= q, in the last range
    out = MsmResult()
    rc = ctx._lib.msm_run(ctx._h, bad, n, 0, CT.byref(MsmOpts(strict=1)), CT.byref(out))
    assert rc == _lib.MSM_ERR_SCALAR
    got, _ = ctx.run(host)             # the same context, the clean buffer: pipelined again
    assert got.as_tuple() == want.as_tuple()
    # without strict the bad scalar is reduced mod q: same element as the reduced value resident in HBM
    rc = ctx._lib.msm_run(ctx._h, bad, n, 0, CT.byref(MsmOpts()), CT.byref(out))
    assert rc == _lib.MSM_OK
    dev2 = ctx.device_alloc(32 * n)
    ctx.device_upload(dev2, bytes(bad))
    ref, _ = ctx.run_device(dev2, n)
    assert int.from_bytes(bytes(out.x), "little") == ref.x and int.from_bytes(bytes(out.y), "little") == ref.y
    # a tight workspace makes the windows run over ranges of the points of their own: plain staged upload, same result
    ctx.set_workspace_limit(3 << 30)
    got, _ = ctx.run(host)
    assert got.as_tuple() == want.as_tuple()
    ctx.set_workspace_limit(0)
    ctx.close()


def test_bench_sharded_path_over_rccl_with_one_rank():
    """`bench.py --gpus 1 --force-dist`: the N > 1 code path of the bench itself -- RCCL process group with a device id, barrier,
    device all-reduce of the times, `all_gather_object`, the shard exchange on device tensors, both splits timed and verified --
    with the one rank a one-GPU box can give it."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1", "--log2n", "18",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["verified"] is True and d["split"] == "windows", d
    assert "window-shard x1" in d["config"]["parallelism"] and len(d["ranks"]) == 1 and d["ranks"][0]["all_gather_ms"] is not None
    assert d["other_splits"][0]["split"] == "points" and d["other_splits"][0]["verified"] is True


def test_package_imported_before_torch_shares_one_hip_runtime():
    """The ROCm wheels of torch bundle their own libamdhip64; a process that loaded libmsm_hip.so first used to map the system
    runtime, torch then a second one, and torch saw no GPU.  The loader
    now maps the runtime torch would use before the library (montgomery_amd/_lib.py): either import order leaves ONE runtime.
    A child process imports the package and runs an MSM BEFORE it imports torch, then uses a torch tensor as the scalar buffer."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys
sys.path.insert(0, %r)
from montgomery_amd.api import MsmContext
assert "torch" not in sys.modules
ctx = MsmContext()
n = 1 << 12
ctx.generate_points(n, seed=3)
dev, sb = ctx.generate_scalars(n, seed=4, to_host=True)
first, _ = ctx.run_device(dev, n)
import torch
assert torch.cuda.is_available(), "torch sees no GPU after montgomery_amd was imported first"
t = torch.frombuffer(bytearray(sb), dtype=torch.uint8).to("cuda:0")
torch.cuda.synchronize()
second, _ = ctx.run_device(t.data_ptr(), n)
assert second.as_tuple() == first.as_tuple()
maps = open("/proc/self/maps").read()
libs = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l})
assert len(libs) == 1, libs
ctx.close()
print("ok", libs[0])
''' % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0 and out.stdout.strip().startswith("ok"), out.stdout[-1500:] + out.stderr[-3000:]
