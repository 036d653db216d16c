"""gpurun_out/pmc_<tag>_2p<lg>/ (tools/pmc_headline.sh) -> profiles/<tag>_pmc_2p<lg>.json: per-kernel counter sums of the four
PMC passes, the clock every kernel held, and the figures bench.py quotes per pair addition of the tree kernel:
HBM bytes (FETCH_SIZE x 2 -- gfx950 tallies wide coalesced reads at half their size, MI355X_MICROARCH.md -- plus WRITE_SIZE,
both reported in KB) and SQ_INSTS_VALU / 64 lanes, each divided by the ALGORITHMIC pair additions of the MSMs in the run
(msm_result.n_pairs_algo, read from the bench line of the same run)."""
import collections, csv, glob, json, os, sys

tag, lg = sys.argv[1], int(sys.argv[2])
curve = sys.argv[3] if len(sys.argv) > 3 else "bls12-377"     # ed377: the twisted Edwards tree kernel k_te_add
sfx = "" if curve == "bls12-377" else "_" + curve
src = f"gpurun_out/pmc_{tag}{sfx}_2p{lg}"
TREE = "te::k_te_add<" if curve == "ed377" else "k_batch_add<msm::CvBls377"
out, clocks = {}, {}
for kind in ("fetch", "write", "sq", "grbm"):
    fs = sorted(glob.glob(f"{src}/pmc_{kind}/*/*_counter_collection.csv"), key=os.path.getmtime)[-1:]
    if not fs:
        continue
    agg = collections.OrderedDict()
    clk = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d = agg.setdefault(k, {})
        d.setdefault("launches", set()).add(r["Dispatch_Id"])
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if kind == "grbm" and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and dur >= 300000:
            c = clk.setdefault(k, [0.0, 0, 0])
            c[0] += float(r["Counter_Value"]); c[1] += dur; c[2] += 1
    for k in agg:
        agg[k]["launches"] = len(agg[k]["launches"])
    out[kind] = agg
    if kind == "grbm":
        # (a counter that glitches on one dispatch reads hundreds of GHz: such kernels are left out)
        clocks = {k: {"dispatches": n, "wall_ms": round(d / 1e6, 3), "ghz": round(c / 8 / d, 3)} for k, (c, d, n) in clk.items()
                  if c / 8 / d < 3.0}

# the bench line of the fetch pass: pair additions of the MSMs the run held (timed step + serialised step)
pairs_algo = pairs_issued = msms = entries = None
window_bits = None
for line in open(f"{src}/pmc_fetch.log", errors="ignore"):
    if line.startswith("{"):
        b = json.loads(line)
        msms = b["steps"] + b["warmup"] + (0 if curve == "ed377" else 1)   # + the serialised "exclusive" step
        pairs_algo = b["roofline"]["pair_adds_per_step"] * msms
        pairs_issued = b["roofline"]["pair_adds_issued_per_step"] * msms
        window_bits = b["config"]["window_bits"]
        entries = (1 if curve == "ed377" else 2) * (1 << lg) * b["config"]["windows"] * msms
per_pair = {}
if pairs_algo:
    def tot(kind, ctr, key):
        return sum(v.get(ctr, 0.0) for k, v in out.get(kind, {}).items() if key in k)
    gk, rk = (TREE + "0>", TREE + "1>") if curve == "ed377" else (TREE + ", 0>", TREE + ", 1>")
    for name, key in (("gather_round", gk), ("regular_rounds", rk), ("all_rounds", TREE)):
        fetch, write, valu = tot("fetch", "FETCH_SIZE", key) * 1024, tot("write", "WRITE_SIZE", key) * 1024, tot("sq", "SQ_INSTS_VALU", key)
        per_pair[name] = {"fetch_bytes_x2": 2 * fetch, "write_bytes": write, "valu_wave_insts": valu}
    # pair additions per kind of round: the gather round does half of a bucket's additions (n - 1 of them for n entries split
    # as n/2 in round 1, the rest later), so the split comes from the issued counts of the bench line where available
    half = pairs_algo / 2
    # the half / half split holds for windows of up to 16 bits (round 1 gathers, every other big round is index-free); bigger
    # windows run rounds 1 and 2 in the gather kernel and most later ones through descriptors: only the total is meaningful
    kinds = (("gather_round", half), ("regular_rounds", half), ("all_rounds", pairs_algo)) if window_bits <= 16 else (("all_rounds", pairs_algo),)
    if window_bits > 16:
        per_pair = {"all_rounds": per_pair["all_rounds"], "raw_totals_by_kernel_mode": {k: v for k, v in per_pair.items() if k != "all_rounds"}}
    for name, denom in kinds:
        p = per_pair[name]
        p["pair_adds_basis"] = denom
        p["hbm_bytes_per_pair_add"] = (p["fetch_bytes_x2"] + p["write_bytes"]) / denom
        p["valu_insts_per_pair_add"] = p["valu_wave_insts"] / (denom / 64)
# the scatter phase (digits + sort kernels): HBM bytes per (entry, window) and its share of the wall time
sort_keys = ("k_digits", "k_te_digits", "k_hist", "k_colscan", "k_slice_scan", "k_coarse_offsets", "k_vscan", "k_radix_", "k_fine_hist", "k_scatter_lds",
             "k_chunk_order", "k_pscan", "k_bucket_max", "k_pair_", "k_bin_")
def is_sort(k):
    return any(s in k for s in sort_keys)
scatter = None
if entries:
    fb = sum(v.get("FETCH_SIZE", 0.0) for k, v in out.get("fetch", {}).items() if is_sort(k)) * 1024 * 2
    wb = sum(v.get("WRITE_SIZE", 0.0) for k, v in out.get("write", {}).items() if is_sort(k)) * 1024
    wall = sum(v["wall_ms"] for k, v in clocks.items() if is_sort(k))
    scatter = {"entries": entries, "fetch_bytes_x2": fb, "write_bytes": wb, "hbm_bytes_per_entry": (fb + wb) / entries,
               "wall_ms_of_kernels_over_0p3ms": wall, "note": "digits + histogram + scans + radix passes (+ fills excluded); entries = 2 N K x MSMs in the run"}
os.makedirs("profiles", exist_ok=True)
json.dump({"command": f"rocprofv3 --pmc <one counter group per run: FETCH_SIZE | WRITE_SIZE | SQ_* | GRBM_GUI_ACTIVE> -- python3 bench.py "
                      f"--steps 1 --warmup 0 --log2n {lg}{' --curve ' + curve if sfx else ''} --no-cpu-baseline --no-verify --no-other-configs --no-pcie",
           "note": "FETCH_SIZE / WRITE_SIZE in KB as reported (x 1024 below); FETCH_SIZE is doubled for HBM bytes (gfx950 halves wide "
                   "coalesced reads, MI355X_MICROARCH.md); SQ_* cycle counters in quad-cycles; GRBM_GUI_ACTIVE summed over the 8 XCDs. "
                   "Per-pair figures divide by algorithmic pair additions (half of them in the gather round).",
           "window_bits": window_bits, "msms_in_run": msms, "pair_adds_algorithmic": pairs_algo, "pair_adds_issued": pairs_issued,
           "per_pair_addition": per_pair, "scatter_phase": scatter, "effective_clock_ghz": clocks, "counters": out},
          open(f"profiles/{tag}_pmc{sfx}_2p{lg}.json", "w"), indent=1)
print("collected", tag, lg, json.dumps(per_pair.get("all_rounds", {})))
