// Digits and counting sort of one window group: scalars -> signed window digits -> bucket-sorted payload slots, every
// bucket padded to a multiple of 2^logG slots, plus the offset tables of the tail rounds.
// (reference phases: decompose + slices src/msm-batched-affine.ts:175-203, integrateBucketCounts :423-447, sortPoints :456-502)
#include "msm_internal.h"

using namespace msm;
using namespace msmi;

namespace msmi {

void sort_kernel_attributes() {
  // dynamic LDS above the 64 KB a launch gets by default
  HIPCHK(hipFuncSetAttribute((const void*)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_scatter_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_bin_split, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bin_split_lds(1u << BS_MAX_AB)));
  HIPCHK(hipFuncSetAttribute((const void*)k_bin_pairs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bin_pairs_lds(1u << BS_MAX_FB)));
  HIPCHK(hipFuncSetAttribute((const void*)k_bin_slots, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bin_slots_lds(1u << BS_MAX_FB)));
  const int dig_lds = 16 * (1 << BS_MAX_AB) * 4;   // the fused histogram of up to 16 windows
  HIPCHK(hipFuncSetAttribute((const void*)k_digits<CvBls377>, hipFuncAttributeMaxDynamicSharedMemorySize, dig_lds));
  HIPCHK(hipFuncSetAttribute((const void*)k_digits<CvBls381>, hipFuncAttributeMaxDynamicSharedMemorySize, dig_lds));
  HIPCHK(hipFuncSetAttribute((const void*)k_digits<CvPallas>, hipFuncAttributeMaxDynamicSharedMemorySize, dig_lds));
  HIPCHK(hipFuncSetAttribute((const void*)te::k_te_digits, hipFuncAttributeMaxDynamicSharedMemorySize, dig_lds));
}

// windows [k_lo, k_hi) over n points whose scalars start at d_scalars (n x 8 words); queues everything on w.stream, records
// w.ev[0] (start), w.ev[1] (digits done), w.ev[2] (sort done) and returns after ONE read-back (largest bucket, scan totals ->
// w.h_info: the launch geometry of the tree)
void sort_window_group(msm_ctx* ctx, msm_ctx::Workspace& w, const uint32_t* d_scalars, uint64_t n, const Plan& pl, int k_lo, int k_hi,
                       GroupStats& st, SortOut& so, GroupDigits* share) {
  hipStream_t s = w.stream;
  const bool produce = share && share->produce;              // digits of all groups' windows, nothing else
  const bool consume = share && !share->produce && share->valid;
  const int kc_d = k_hi - k_lo;             // windows the digit kernel slices
  const uint32_t L = pl.L;
  const bool te = ctx->is_te();
  const uint64_t two_n_d = te ? n : 2 * n;  // entries per digit window: both GLV halves, or the plain scalar
  const uint64_t n_entries = (uint64_t)kc_d * two_n_d;
  // On window tables the kc_d digit windows of the group are ONE window for everything behind the digit kernel: the digit
  // array [kc_d][two_n_d] read as one run of entries, entry j = (window, point, half) naming row j of the tables.
  const int kc = pl.tables ? 1 : kc_d;
  const uint64_t two_n = pl.tables ? n_entries : two_n_d;
  const uint64_t nb = (uint64_t)kc * L;

  // padding granule G = 2^g: about 1/8 of the mean bucket population (pads cost G/2 slots per bucket; what is
  // left after g regular rounds, ~8 elements per bucket, is finished without inversions by k_bucket_finish)
  // Bigger buckets leave more: pads are identity pairs that occupy a lane for nothing (mean / (2 * left) of all slots), and
  // k_bucket_finish is cheap next to them.  Measured (profiles/r04_experiments.txt items 6 and 14, best `left` per mean bucket):
  // 64 entries 8 with 16-bit windows (2^20: 3.89 against 3.93 ms) but 16 with the big ones (2^25: 81.5 -> 79.9);
  // 128 .. 256: 16 (2^21 6.92 -> 6.88, 2^22 12.45 -> 12.17, Ed-377 2^20 2.64 -> 2.59, 2^26 145.7 -> 142.5, 2^27 304.9 -> 301.8);
  // from 512: 32 (2^23 22.65 -> 22.18, 2^26 at c = 16 152.7 -> 152.1); 32 entries: 8 (2^24: 16 costs 8 %).
  uint64_t mean = std::max<uint64_t>(1, two_n / (pl.fold ? L / 2 : L));   // (a folded plan's lower windows fill half their buckets)
  // (on window tables at 2^20 points, mean 112 at c = 18: 8 -- a third regular round -- 3.37 against 3.41 ms)
  uint64_t per_bucket_left = mean >= 512 ? 32 : (mean >= 128 || (mean >= 64 && pl.c >= 18 && !pl.tables)) ? 16 : 8;
  MSM_KNOB(per_bucket_left, "MSM_PBL", 1);
  uint32_t logG = 1;
  while (logG < 10 && (1ull << (logG + 1)) * per_bucket_left <= mean) logG++;

  if (!consume && !produce) ctx->ensure(w.dig, n_entries * 4);
  if (produce) {
    share->valid = false;
  } else {
  ctx->ensure(w.counts, nb * 4);
  ctx->ensure(w.cursor, (nb + 1) * 4);   // [nb] = the total (k_pscan_final)
  ctx->ensure(w.tail_off, (size_t)34 * (nb + 1) * 4);
  ctx->ensure(w.info, 64 * 4);
  }

  // Sort path: LDS-privatised histogram / ranking, every pass staged through the LDS so that a wave store is a full segment
  // (sort_kernels.h; a direct scatter sends each 4-byte payload to a line of its own: round 1 wrote 7.7x the algorithmic bytes).
  //   one level  : a window's counters fit the LDS (c <= 16) and the input is small
  //   radix split: c <= 16, big inputs -- 2^(c-8) coarse bins x 128 buckets, two passes over pairs of 4-byte arrays
  //   bin split  : c > 16 (up to c = 24, the largest window make_plan accepts) -- coarse bins x up to 2^12 buckets, two passes
  //                over 8-byte records, the histogram of the first fused into the digit kernel, the second emitting the pairs
  //                of round 1 in an order that keeps its row gathers local
  const int cbits = pl.L_log;   // bits of a bucket index
  const bool fits_lds = (size_t)L * 4 <= 128 * 1024;
  // measured (round 5, with one ds_add per key as the ranking: tools/sortpath_sweep.sh): the split wins from 2^21 entries per
  // window -- 2^20 points: sort 0.31 against 0.36 ms, 2^21: 0.51 / 0.85; 2^19: level, below: the one-level sort (2^16 0.11 / 0.14)
  const bool want_radix = fits_lds && cbits > (int)RX_FINE_BITS && two_n >= (te ? 1ull << 22 : 1ull << 21);
  // (the merged window of a run on window tables has kc = 1: the radix split would give its last pass one block per coarse bin,
  // 2^(c-8) of them for the whole chip -- the bin split cuts it into 2^10 bins whatever the window is)
  const bool want_bins = !fits_lds || (pl.tables && two_n >= (1ull << 22));
  const bool bin_split = !fits_lds || want_bins;
  const bool radix = !bin_split && want_radix && cbits > (int)RX_FINE_BITS && cbits - (int)RX_FINE_BITS <= 8;
  WinSplit ws{};
  uint32_t hb = 1, nbmax = 1;   // bin split: coarse bins per window (stride of the bin tables), most buckets of one bin
  if (bin_split) {
    if (kc > 16) throw MsmFail{MSM_ERR_INTERNAL, "more than 16 windows in a group of a window size above 16"};
    for (int kk = 0; kk < kc; kk++) {
      // bits the digits of this window really have: the top window of a scalar is usually short (msm_kernels.h, WinSplit)
      // (a window below the top one holds signed digits of magnitude <= 2^(c - 1); the top one what is left of the scalar)
      const bool top = k_lo + kk == pl.K - 1;
      const int eff = pl.tables ? cbits : std::max(1, top ? std::min(cbits, pl.bits - (k_lo + kk) * pl.c) : std::min(cbits, pl.c - 1));
      // up to 10 bits: pass A sorts the window outright; else 2^10 coarse bins (16-entry runs of a 16 k tile), 2^11 if the
      // fine part would otherwise exceed 2^12 buckets per bin
      const int ab_big = 10;
      const int abk = eff <= (int)ab_big ? eff : std::max((int)ab_big, eff - (int)BS_MAX_FB);
      if (abk > (int)BS_MAX_AB) throw MsmFail{MSM_ERR_INTERNAL, "window too wide for the bin split"};
      ws.ab[kk] = (uint8_t)abk;
      ws.fb[kk] = (uint8_t)(eff - abk);
      hb = std::max(hb, 1u << abk);
      nbmax = std::max(nbmax, 1u << (eff - abk));
    }
  } else if (radix) {
    if (kc > 16) throw MsmFail{MSM_ERR_INTERNAL, "more than 16 windows in a radix-split group"};
    for (int kk = 0; kk < kc; kk++) { ws.ab[kk] = (uint8_t)(cbits - (int)RX_FINE_BITS); ws.fb[kk] = (uint8_t)RX_FINE_BITS; }
  }
  const uint32_t Hn = radix ? 1u << (cbits - (int)RX_FINE_BITS) : 1u;   // radix split: coarse bins = blocks of its last pass per window
  const uint32_t V = (uint32_t)kc * (bin_split ? hb : Hn);              // bins of the group
  // slices of the entries: block (b, kk) of the histogram and of the first pass owns entries [b * chunk, (b + 1) * chunk)
  uint32_t sortB = 1;
  uint64_t chunk = two_n, pps = n;   // entries per slice and window; points per slice
  if (bin_split) {
    // the digit kernel histograms ALL windows of the group over its slice of the points: four slices per CU
    const uint64_t mult = 4;
    // one slice per 4 096 points at least; the digit kernel wants a few blocks per CU whatever kc_d is, k_slice_scan walks
    // the kc_d * sortB rows of the merged window with 32 lanes per column
    sortB = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)mult * ctx->n_cu, n / 4096));
    // (pass A names an entry by its offset inside the slice in BS_SPAN_LOG bits: from 2^27 points the slices multiply instead of growing)
    const uint64_t max_pps = (1ull << BS_SPAN_LOG) / (te ? 1 : 2);
    sortB = (uint32_t)std::max<uint64_t>(sortB, (n + max_pps - 1) / max_pps);
    pps = (n + sortB - 1) / sortB;
    chunk = (te ? 1 : 2) * pps;
    if (consume) {
      // (the producer sliced all groups' windows at once: same slices, and the stride of the widest window's bins)
      if (share->sortB != sortB || share->pps != pps || share->chunk != chunk || share->hb < hb)
        throw MsmFail{MSM_ERR_INTERNAL, "shared digits do not match the group's sort geometry"};
      hb = share->hb;
    } else {
      ctx->ensure(w.block_hist, (size_t)kc_d * sortB * hb * 4 + 64);
    }
  } else {
    // big inputs: finer slices also keep the round-1 gathers of neighbouring lanes inside one Infinity-Cache-sized
    // range of point rows (measured: 193 -> 183 ms at 2^26); small inputs: fewer, larger blocks (less fixed cost)
    const uint64_t mult = two_n >= (1ull << 27) ? 8 : two_n >= (1ull << 24) ? 4 : 2;   // measured 2^21 .. 2^26
    uint64_t want = std::max<uint64_t>(1, (mult * ctx->n_cu + kc - 1) / kc);
    uint64_t maxb = std::max<uint64_t>(1, two_n / 8192);
    sortB = (uint32_t)std::min<uint64_t>(want, maxb);
    chunk = (two_n + sortB - 1) / sortB;
    if (!produce) ctx->ensure(w.block_hist, (size_t)kc * sortB * L * 4 + 64);
  }
  uint32_t* d_bin_start = nullptr;   // bin split: V + 1 starts of the bins in the record array
  uint32_t *d_extra_first = nullptr, *d_mp_first = nullptr, *d_part_pair_off = nullptr;   // bin split: the parts of heavy bins
  uint32_t part_len = 0, part_grid = 0;
  // Big windows over a big table: round 1 walks the pairs in the order the last pass of the sort emits them (see below); decided
  // here because the count pass of the bin split prepares the parts of heavy bins for that pass.
  bool will_chunk = false;
  {
    // from more than 2^23 rows of 256 bytes = 2 GB (measured, tools/chunk_plain.py on the round-5 tuning build: at
    // 2^23 rows -- 2^23 points, or 2^20 on seven window tables -- the plain slot form is 2.5 % ahead: 20.15 against 20.64 ms,
    // 3.23 / 3.32; at 2^23.8 rows level; at 2^24 rows the tile order wins by 10 %: 38.6 against 43.0)
    const int chunk_rows_log = 23;
    const uint64_t table_rows = pl.tables ? (uint64_t)kc_d * n : n;
    const bool want_chunks = bin_split && !te && pl.c >= 18 && table_rows > (1ull << chunk_rows_log);
    // (round 2 must be an index-free round to read the element records round 1 then writes: logG >= 2)
    will_chunk = bin_split && want_chunks && !te && logG >= 2;
  }
  if (produce && !bin_split) return;   // (only the bin split takes its histograms from the digit kernel)
  if (produce) ctx->ensure(w.dig, n_entries * 4);
  if (share && !share->produce && !bin_split) throw MsmFail{MSM_ERR_INTERNAL, "shared digits for a group that does not take the bin split"};
  // where this group's digits and slice histograms are: its own buffers, or its part of the producer's
  const uint32_t* dig_p = consume ? share->dig + (uint64_t)(k_lo - share->k_lo) * two_n_d : (const uint32_t*)w.dig.p;
  uint32_t* hist_p = consume ? share->hist + (uint64_t)(k_lo - share->k_lo) * sortB * hb : (uint32_t*)w.block_hist.p;

  if (consume) HIPCHK(hipStreamWaitEvent(s, share->ready, 0));   // (the producer's launch is timed by the caller, not as this group's wait)
  HIPCHK(hipEventRecord(w.ev[0], s));
  if (!consume) {
    // digits; the bin split also takes the histogram of its first pass from here (one slice of the points per block)
    const uint32_t grid = bin_split ? sortB : (uint32_t)((n + 255) / 256);
    const uint32_t per = bin_split ? (uint32_t)pps : 256u;
    uint32_t* hist = bin_split ? (uint32_t*)w.block_hist.p : nullptr;
    const size_t lds = bin_split ? (size_t)kc_d * hb * 4 : 0;
    WinSplit ws_d = ws;   // the digit kernel counts per DIGIT window: on tables all of them with the merged window's cut
    if (pl.tables) for (int kk = 0; kk < 16; kk++) ws_d.fb[kk] = ws.fb[0];
    // (a slice per block leaves four blocks per CU: 1024 threads each keep the SIMDs full, 256 left them at four waves)
    const uint32_t threads = bin_split ? 1024u : 256u;
    if (te)
      hipLaunchKernelGGL(te::k_te_digits, dim3(grid), dim3(threads), lds, s, (uint32_t*)w.dig.p, d_scalars, (uint32_t)n, pl.c, pl.K,
                         k_lo, kc_d, pl.strict ? 1 : 0, (uint32_t*)ctx->errflag.p, per, hist, hb, pack_fine_bits(ws_d), pl.b_lo, pl.b_n, pl.bt_lo, pl.bt_n);
    else
      W_LAUNCH(ctx, k_digits, dim3(grid), dim3(threads), lds, s, (uint32_t*)w.dig.p, d_scalars, (uint32_t)n, pl.c, pl.K, k_lo, kc_d,
                         (pl.no_glv ? 0 : 1) | (pl.fold ? 2 : 0), pl.strict ? 1 : 0, (uint32_t*)ctx->errflag.p, per, hist, hb, pack_fine_bits(ws_d), pl.b_lo, pl.b_n, pl.bt_lo, pl.bt_n);
  }
  if (produce) {
    share->k_lo = k_lo;
    share->dig = (const uint32_t*)w.dig.p;
    share->hist = (uint32_t*)w.block_hist.p;
    share->hb = hb;
    share->sortB = sortB;
    share->pps = pps;
    share->chunk = chunk;
    HIPCHK(hipEventRecord(share->ready, s));
    share->valid = true;
    return;
  }
  HIPCHK(hipEventRecord(w.ev[1], s));
  if (!bin_split) {
    hipLaunchKernelGGL(k_hist, dim3(sortB, kc), dim3(SORT_THREADS), (size_t)L * 4, s, (uint32_t*)w.block_hist.p,
                       (const uint32_t*)w.dig.p, two_n, chunk, L);
    hipLaunchKernelGGL(k_colscan, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, s, (uint32_t*)w.block_hist.p,
                       (uint32_t*)w.counts.p, sortB, L, (uint32_t)kc);
  } else {
    // per (window, bin): the slices' prefixes and the total -> the bin starts; pass A; the bucket sizes
    ctx->ensure(w.part, ((size_t)2 * V + 2) * 4);
    uint32_t* d_bin_tot = (uint32_t*)w.part.p;
    d_bin_start = d_bin_tot + V;
    ctx->ensure(w.rec, n_entries * 8);
    // (on tables the slices of all kc_d digit windows are the slices of the one merged window, in the same row order)
    hipLaunchKernelGGL(k_slice_scan, dim3((hb + 31) / 32, kc), dim3(1024), 0, s, hist_p, d_bin_tot,
                       pl.tables ? sortB * (uint32_t)kc_d : sortB, hb, (uint32_t)kc);
    // heavy bins are cut into parts of part_len records, one block each (sort_kernels.h, "Parts of heavy bins"): twice the mean
    // bin, so that uniform digits never see one; a multiple of the tile of pass B.  The table comes out of k_vscan.
    part_len = (uint32_t)std::max<uint64_t>(1u << 16, ((2 * n_entries / V + BP_TILE - 1) / BP_TILE) * BP_TILE);
    part_grid = V + V / 2 + 1;
    ctx->ensure(w.parts, ((size_t)2 * (V + 1) + (V + 2)) * 4);
    d_extra_first = (uint32_t*)w.parts.p;
    d_mp_first = d_extra_first + (V + 1);
    d_part_pair_off = d_mp_first + (V + 1);
    PartTables ptab{d_extra_first, d_mp_first, d_part_pair_off, hb, part_len, {}};
    for (int kk = 0; kk < 16; kk++) ptab.fb[kk] = ws.fb[kk];
    hipLaunchKernelGGL(k_vscan, dim3(1), dim3(SCAN_THREADS), 0, s, d_bin_start, (const uint32_t*)d_bin_tot, V, (uint32_t*)w.info.p, ptab);
    // a block of pass A takes as many consecutive slices of the digit kernel as make two tiles
    const uint32_t per_block = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(sortB, std::max<uint64_t>(1, (1ull << BS_SPAN_LOG) / chunk)),
                                                            std::max<uint64_t>(1, (2 * (uint64_t)BS_TILE + chunk - 1) / chunk));
    hipLaunchKernelGGL(k_bin_split, dim3((sortB + per_block - 1) / per_block, kc_d), dim3(BS_THREADS), bin_split_lds(hb), s, (uint2*)w.rec.p,
                       (const uint32_t*)d_bin_start, (const uint32_t*)hist_p, dig_p, two_n_d, chunk, hb, ws,
                       pl.tables ? 1u : 0u, sortB, per_block);
    // k_bin_count leaves the largest bucket and the pair count in `info` as k_bucket_max does for the other paths, and writes
    // every bucket of the bins it is launched for: `counts` needs a fill only where a window's digits do not span all L buckets
    bool spans_all = true;
    for (int kk = 0; kk < kc; kk++) spans_all &= (1u << ws.ab[kk]) == hb && (int)ws.ab[kk] + (int)ws.fb[kk] == cbits;
    if (!spans_all) HIPCHK(hipMemsetAsync(w.counts.p, 0, nb * 4, s));   // (`info` was cleared by k_vscan)
    ctx->ensure(w.sub, (size_t)(V + 2) * nbmax * 4);   // rows: the parts of multi-part bins, at most V of them
    hipLaunchKernelGGL(k_bin_count, dim3(part_grid), dim3(BC_THREADS), (size_t)nbmax * 4, s, (uint32_t*)w.counts.p, (const uint32_t*)d_bin_start,
                       (const uint2*)w.rec.p, hb, L, ws, (uint32_t*)w.info.p, V, (const uint32_t*)d_extra_first, (const uint32_t*)d_mp_first,
                       part_len, (uint32_t*)w.sub.p, nbmax);
    hipLaunchKernelGGL(k_part_scan, dim3(V), dim3(PSC_THREADS), 0, s, (uint32_t*)w.counts.p, (uint32_t*)w.sub.p, d_part_pair_off,
                       (const uint32_t*)d_extra_first, (const uint32_t*)d_mp_first, hb, L, nbmax, ws, (uint32_t*)w.info.p,
                       will_chunk ? 1u : 0u);
  }
  int RT = 0;
  uint64_t total_slots = 0;
  uint32_t max_bucket = 0;
  {
    // largest bucket -> number of tail rounds RT, then the multi-block scan of RT + 2 quantities.  The scan kernels take RT
    // from the device (pscan_nq), so ONE read-back behind them brings the largest bucket and the totals together.
    if (!bin_split) {
      HIPCHK(hipMemsetAsync(w.info.p, 0, 64 * 4, s));
      hipLaunchKernelGGL(k_bucket_max, dim3((uint32_t)std::min<uint64_t>(1024, (nb + 255) / 256)), dim3(256), 0, s,
                         (const uint32_t*)w.counts.p, (uint32_t)nb, (uint32_t*)w.info.p);
    }
    const uint32_t nblocks = (uint32_t)((nb + PS_SPAN - 1) / PS_SPAN);
    ctx->ensure(w.scan_partial, (size_t)PS_MAX_NQ * nblocks * 4);
    hipLaunchKernelGGL(k_pscan_partial, dim3(nblocks), dim3(PS_BLOCK), 0, s, (const uint32_t*)w.counts.p, (uint32_t)nb, logG,
                       (const uint32_t*)w.info.p, (uint32_t*)w.scan_partial.p, nblocks);
    hipLaunchKernelGGL(k_pscan_top, dim3(1), dim3(SCAN_THREADS), 0, s, (uint32_t*)w.scan_partial.p, nblocks, logG,
                       (uint32_t*)w.info.p);
    hipLaunchKernelGGL(k_pscan_final, dim3(nblocks), dim3(PS_BLOCK), 0, s, (const uint32_t*)w.counts.p, (uint32_t)nb, logG,
                       (const uint32_t*)w.scan_partial.p, nblocks, (uint32_t*)w.cursor.p, (uint32_t*)w.tail_off.p,
                       (const uint32_t*)w.info.p);
    HIPCHK(hipEventRecord(w.ev[7], s));
  }
  // Big windows over a big table: round 1 walks the pairs in the order the last pass of the sort emits them -- tile by tile of
  // a bin's records, which are in point order -- and writes every sum to the element index that comes with the pair, as
  // 64-byte records that round 2 reads back (batch_add.h).  Needed once the 128 slots of a wave span more than ~1 GB of
  // rows (measured: from 2 GB of rows, see below).  Smaller tables take the bucket-ordered slots of k_bin_slots.
  const uint32_t* round1_slots = nullptr;
  const uint32_t* round1_dest = nullptr;
  uint64_t rec_y_off = 0;   // 12-word fields: where the y records of round 1's results start inside w.rows1
  bool chunked = will_chunk;
  // The bin split's last pass needs nothing from the host but room for what it writes, and the padded slots have a bound --
  // every non-empty bucket pads by less than G -- so it is launched BEHIND the scans at once and the read-back of the totals
  // (which the tree's launches wait for) crosses on a side stream while it runs: no idle gap of a host round trip in front of
  // it (25 us of the 3.4 ms of an MSM over 2^20 points).
  // (the parts of a heavy bin pair their entries up on their own: a bucket gains at most one slot per part that holds an odd
  // number of its entries -- sort_kernels.h, "Parts of heavy bins")
  const uint64_t parts_extra = will_chunk ? std::min<uint64_t>(n_entries, (uint64_t)V * nbmax) : 0;
  const uint64_t slots_bound = n_entries + parts_extra + (((uint64_t)1 << logG) - 1) * std::min<uint64_t>(nb, n_entries);
  auto launch_bin_pass = [&](uint64_t slots_cap) {
    if (chunked) {
      const uint64_t pairs_cap = slots_cap / 2 + 1;
      ctx->ensure(w.slots2, pairs_cap * 8);
      ctx->ensure(w.dest, pairs_cap * 4);
      rec_y_off = (pairs_cap * 64 + 255) & ~(uint64_t)255;
      ctx->ensure(w.rows1, 2 * rec_y_off + 256);
      hipLaunchKernelGGL(k_bin_pairs, dim3(part_grid), dim3(BP_THREADS), bin_pairs_lds(nbmax), s, (uint2*)w.slots2.p, (uint32_t*)w.dest.p,
                         (const uint2*)w.rec.p, (const uint32_t*)d_bin_start, (const uint32_t*)w.cursor.p, hb, L, nbmax, ws, V,
                         (const uint32_t*)d_extra_first, (const uint32_t*)d_mp_first, part_len, (const uint32_t*)w.sub.p,
                         (const uint32_t*)d_part_pair_off);
      round1_slots = (const uint32_t*)w.slots2.p;
      round1_dest = (const uint32_t*)w.dest.p;
    } else {
      ctx->ensure(w.slots, std::max<uint64_t>(slots_cap, 2) * 4);
      HIPCHK(hipMemsetAsync(w.slots.p, 0xFF, std::max<uint64_t>(slots_cap, 2) * 4, s));
      round1_slots = (const uint32_t*)w.slots.p;
      hipLaunchKernelGGL(k_bin_slots, dim3(part_grid), dim3(BP_THREADS), bin_slots_lds(nbmax), s, (uint32_t*)w.slots.p,
                         (const uint2*)w.rec.p, (const uint32_t*)d_bin_start, (const uint32_t*)w.cursor.p, hb, L, nbmax, ws, V,
                         (const uint32_t*)d_extra_first, (const uint32_t*)d_mp_first, part_len, (const uint32_t*)w.sub.p);
    }
  };
  if (bin_split) launch_bin_pass(slots_bound);
  {
    HIPCHK(hipStreamWaitEvent(w.side, w.ev[7], 0));
    HIPCHK(hipMemcpyAsync(w.h_info, w.info.p, 64 * 4, hipMemcpyDeviceToHost, w.side));
    HIPCHK(hipStreamSynchronize(w.side));
    total_slots = w.h_info[0];
    max_bucket = w.h_info[1];
    st.n_pairs_algo += (uint64_t)w.h_info[INFO_ALGO_PAIRS] | ((uint64_t)w.h_info[INFO_ALGO_PAIRS + 1] << 32);
    const uint32_t capmax = (max_bucket + (1u << logG) - 1) >> logG;
    while (RT < 32 && (1u << RT) < capmax) RT++;
  }
  st.max_bucket = std::max<uint64_t>(st.max_bucket, max_bucket);
  if (total_slots > slots_bound) throw MsmFail{MSM_ERR_INTERNAL, "the padded slots exceed their bound"};
  if (bin_split && chunked && total_slots < 2) {   // (no entry at all: the plain slot form handles the empty tree)
    chunked = false;
    round1_dest = nullptr;
    rec_y_off = 0;
    launch_bin_pass(2);
  }
  // scatter
  if (!bin_split) {
    ctx->ensure(w.slots, std::max<uint64_t>(total_slots, 2) * 4);
    HIPCHK(hipMemsetAsync(w.slots.p, 0xFF, std::max<uint64_t>(total_slots, 2) * 4, s));
    round1_slots = (const uint32_t*)w.slots.p;
    // The last pass of the radix split gives one block to a coarse bin: a bucket that holds far more than a bin's share of the
    // entries (one scalar repeated: a whole window in one bucket) would make one block walk them all.  The histogram is the
    // one-level sort's, so such an input -- the host has the largest bucket by now -- takes the one-level scatter instead, which
    // splits the ENTRIES across the blocks (its partial-line writes do not matter when most payloads go to a few long runs):
    // one scalar repeated at 2^20 points, sort phase 1.39 -> 0.4x ms.
    const bool heavy = radix && max_bucket >= (1u << 16) && (uint64_t)max_bucket * Hn >= 16 * two_n;
    if (radix && !heavy) {
      // pass A: coarse split into dig2 / idx2; pass B: one block per coarse bin, payloads to their padded slots
      ctx->ensure(w.part, ((size_t)kc * sortB * Hn + 2 * (size_t)V + 2) * 4);
      uint32_t* d_blk_off = (uint32_t*)w.part.p;
      uint32_t* d_vtot = d_blk_off + (size_t)kc * sortB * Hn;
      uint32_t* d_vstart = d_vtot + V;
      ctx->ensure(w.dig2, n_entries * 4);
      ctx->ensure(w.idx2, n_entries * 4);
      const uint64_t co = (uint64_t)kc * (sortB + 1) * Hn;
      hipLaunchKernelGGL(k_coarse_offsets, dim3((uint32_t)((co + 255) / 256)), dim3(256), 0, s, d_blk_off, d_vtot,
                         (const uint32_t*)w.block_hist.p, (const uint32_t*)w.counts.p, sortB, L, Hn, (uint32_t)kc);
      hipLaunchKernelGGL(k_vscan, dim3(1), dim3(SCAN_THREADS), 0, s, d_vstart, (const uint32_t*)d_vtot, V, (uint32_t*)nullptr, PartTables{});
      hipLaunchKernelGGL(k_radix_coarse, dim3(sortB, kc), dim3(RX_THREADS), 0, s, (uint32_t*)w.dig2.p, (uint32_t*)w.idx2.p,
                         (const uint32_t*)d_vstart, (const uint32_t*)d_blk_off, (const uint32_t*)w.dig.p, two_n, chunk, Hn, ws);
      hipLaunchKernelGGL(k_radix_fine, dim3(V), dim3(RXB_THREADS), 0, s, (uint32_t*)w.slots.p, (const uint32_t*)w.cursor.p,
                         (const uint32_t*)d_vstart, (const uint32_t*)w.dig2.p, (const uint32_t*)w.idx2.p, Hn, L, ws);
    } else {
      hipLaunchKernelGGL(k_scatter_lds, dim3(sortB, kc), dim3(SORT_THREADS), (size_t)L * 4, s, (uint32_t*)w.slots.p,
                         (const uint32_t*)w.cursor.p, (const uint32_t*)w.block_hist.p, (const uint32_t*)w.dig.p, two_n,
                         chunk, L);
    }
  }
  HIPCHK(hipEventRecord(w.ev[2], s));
  so.logG = logG;
  so.RT = RT;
  so.total_slots = total_slots;
  so.max_bucket = max_bucket;
  so.round1_slots = round1_slots;
  so.round1_dest = round1_dest;
  so.rec_y_off = rec_y_off;
  so.chunked = chunked;
}

}  // namespace msmi
