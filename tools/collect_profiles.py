"""Copies the judged summaries of one profiling run (gpurun_out/prof_<tag>/) into profiles/ (tracked)."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
for name, dst in (("trace26", f"{tag}_kernel_stats_2p26.csv"), ("trace20", f"{tag}_kernel_stats_2p20.csv"), ("trace_ed20", f"{tag}_kernel_stats_ed377_2p20.csv")):
    f = sorted(glob.glob(f"{src}/{name}/*/*_kernel_stats.csv"), key=os.path.getmtime)   # gpurun merges: keep the newest run
    if f:
        shutil.copy(f[-1], f"profiles/{dst}")
for name in ("bench_2p26.json", "bench_2p20.json", "bench_ed377_2p20.json", "bench_bls381_2p26.json", "bench_bls381_2p20.json",
             "ubench_exec.txt", "ubench_occ.txt", "ubench_bt.txt", "ubench_inv.txt", "ubench_int2.txt", "ubench_mul2.txt", "ubench_mad3.txt",
             "ubench_gather.txt", "ubench_carry.txt", "cpu_baseline.json", "bench_2rank_gloo_2p22.json", "js_bench_2p20.txt", "shard_proxy.txt",
             "skew_time.txt"):
    if os.path.exists(f"{src}/{name}"):
        shutil.copy(f"{src}/{name}", f"profiles/{tag}_{name}")
out = {}
for kind in ("fetch", "write", "sq", "grbm"):
    fs = sorted(glob.glob(f"{src}/pmc_{kind}/*/*_counter_collection.csv"), key=os.path.getmtime)[-1:]
    if not fs:
        continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        d = agg.setdefault(k, {})
        d.setdefault("launches", set()).add(r["Dispatch_Id"])
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k in agg:
        agg[k]["launches"] = len(agg[k]["launches"])
    out[kind] = agg
    if kind == "grbm":
        # the clock each kernel actually held (DVFS give-back, MI355X_MICROARCH.md): GRBM_GUI_ACTIVE / 8 XCDs / wall time,
        # over dispatches of at least 0.3 ms (the quotient reads high on shorter ones)
        clk = collections.OrderedDict()
        for r in csv.DictReader(open(fs[0])):
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or dur < 300000:
                continue
            c = clk.setdefault(r["Kernel_Name"].split("(")[0], [0.0, 0, 0])
            c[0] += float(r["Counter_Value"]); c[1] += dur; c[2] += 1
        out["effective_clock_ghz"] = {k: {"dispatches": n, "wall_ms": round(d / 1e6, 3), "ghz": round(c / 8 / d, 3)} for k, (c, d, n) in clk.items()}
if out:
  json.dump({"command": "rocprofv3 --pmc <counters> (one pass per group: FETCH_SIZE | WRITE_SIZE | SQ_* | GRBM_GUI_ACTIVE) -- python3 bench.py --steps 1 --warmup 0 --log2n 24 --no-cpu-baseline --no-verify (the run holds TWO MSMs: the timed step and the serialised one)",
           "note": "FETCH_SIZE / WRITE_SIZE in KB as reported; gfx950 halves FETCH_SIZE on wide coalesced reads (MI355X_MICROARCH.md). SQ_* cycle counters are in quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs.",
           "counters": out}, open(f"profiles/{tag}_pmc_2p24.json", "w"), indent=1)
print("collected", tag)
