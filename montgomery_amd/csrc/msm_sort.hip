// Digits and counting sort of one window group: scalars -> signed window digits -> bucket-sorted payload slots, every
// bucket padded to a multiple of 2^logG slots, plus the offset tables of the tail rounds.
// (reference phases: decompose + slices src/msm-batched-affine.ts:175-203, integrateBucketCounts :423-447, sortPoints :456-502)
#include "msm_internal.h"

using namespace msm;
using namespace msmi;

namespace msmi {

void sort_kernel_attributes() {
  HIPCHK(hipFuncSetAttribute((const void*)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_scatter_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
}

// windows [k_lo, k_hi) over n points whose scalars start at d_scalars (n x 8 words); queues everything on w.stream, records
// w.ev[0] (start), w.ev[1] (digits done), w.ev[2] (sort done) and returns after ONE read-back (largest bucket, scan totals ->
// w.h_info: the launch geometry of the tree)
void sort_window_group(msm_ctx* ctx, msm_ctx::Workspace& w, const uint32_t* d_scalars, uint64_t n, const Plan& pl, int k_lo, int k_hi,
                       GroupStats& st, SortOut& so) {
  hipStream_t s = w.stream;
  const int kc = k_hi - k_lo;
  const uint32_t L = pl.L;
  const uint64_t nb = (uint64_t)kc * L;
  const bool te = ctx->is_te();
  const uint64_t two_n = te ? n : 2 * n;   // entries per window: both GLV halves, or the plain scalar
  const uint64_t n_entries = (uint64_t)kc * two_n;

  // padding granule G = 2^g: about 1/8 of the mean bucket population (pads cost G/2 slots per bucket; what is
  // left after g regular rounds, ~8 elements per bucket, is finished without inversions by k_bucket_finish)
  // Bigger buckets leave more: pads are identity pairs that occupy a lane for nothing (mean / (2 * left) of all slots), and
  // k_bucket_finish is cheap next to them.  Measured (profiles/r04_experiments.txt items 6 and 14, best `left` per mean bucket):
  // 64 entries 8 with 16-bit windows (2^20: 3.89 against 3.93 ms) but 16 with the big ones (2^25: 81.5 -> 79.9);
  // 128 .. 256: 16 (2^21 6.92 -> 6.88, 2^22 12.45 -> 12.17, Ed-377 2^20 2.64 -> 2.59, 2^26 145.7 -> 142.5, 2^27 304.9 -> 301.8);
  // from 512: 32 (2^23 22.65 -> 22.18, 2^26 at c = 16 152.7 -> 152.1); 32 entries: 8 (2^24: 16 costs 8 %).
  uint64_t mean = std::max<uint64_t>(1, two_n / (pl.fold ? L / 2 : L));   // (a folded plan's lower windows fill half their buckets)
  uint64_t per_bucket_left = mean >= 512 ? 32 : (mean >= 128 || (mean >= 64 && pl.c >= 18)) ? 16 : 8;
  MSM_KNOB(per_bucket_left, "MSM_PBL", 1);
  uint32_t logG = 1;
  while (logG < 10 && (1ull << (logG + 1)) * per_bucket_left <= mean) logG++;

  ctx->ensure(w.dig, n_entries * 4);
  ctx->ensure(w.counts, nb * 4);
  ctx->ensure(w.cursor, nb * 4);
  ctx->ensure(w.tail_off, (size_t)34 * (nb + 1) * 4);
  ctx->ensure(w.info, 64 * 4);

  // Sort path: LDS-privatised histogram / ranking, every pass staged through the LDS so that a wave store is a full segment
  // (sort_kernels.h; a direct scatter sends each 4-byte payload to a line of its own: round 1 wrote 7.7x the algorithmic bytes).
  //   one level  : a window's counters fit the LDS (c <= 16) and the input is small
  //   two passes : c <= 16, big inputs -- 2^(c-8) coarse bins x 128 buckets
  //   three passes: c > 16 (up to c = 24, the largest window make_plan accepts) -- coarse bins x mid bins x 128 (256) buckets
  const int cbits = pl.L_log;   // bits of a bucket index
  const bool fits_lds = (size_t)L * 4 <= 128 * 1024;
  long long want_radix = (fits_lds && cbits > (int)RX_FINE_BITS && two_n >= (1ull << 22)) ? 1 : 0;   // measured: wins from N = 2^21 up
  MSM_KNOB(want_radix, "MSM_RADIX", 0);
  const bool radix = fits_lds && want_radix && cbits > (int)RX_FINE_BITS && cbits - (int)RX_FINE_BITS <= 8;
  const bool one_level = fits_lds && !radix;
  const bool three_pass = !fits_lds;
  const uint32_t fb = three_pass ? (cbits >= 23 ? 8u : 7u) : RX_FINE_BITS;   // bucket bits sorted by the last pass (full windows)
  const uint32_t shift = radix ? (uint32_t)cbits - fb : 0;                  // two passes: log2 of the coarse bins
  const uint32_t Hn = 1u << shift;
  const uint32_t Lp = three_pass ? 1u << ((uint32_t)cbits - fb) : Hn;        // fine windows (blocks of the last pass) per window
  const uint32_t V = (uint32_t)kc * Lp;
  WinSplit ws{};
  if (three_pass) {
    if (kc > 16) throw MsmFail{MSM_ERR_INTERNAL, "more than 16 windows in a group of a window size above 16"};
    const int lp_log = cbits - (int)fb;
    for (int kk = 0; kk < kc; kk++) {
      // bits the digits of this window really have: the top window of a scalar is usually short (sort_kernels.h, WinSplit)
      // (a window below the top one holds signed digits of magnitude <= 2^(c - 1); the top one what is left of the scalar)
      const bool top = k_lo + kk == pl.K - 1;
      const int eff = std::max(1, top ? std::min(cbits, pl.bits - (k_lo + kk) * pl.c) : std::min(cbits, pl.c - 1));
      const int fbk = std::max(0, eff - lp_log);
      const int hi = eff - fbk;
      ws.fb[kk] = (uint8_t)fbk;
      ws.ab[kk] = (uint8_t)std::min(8, hi);
      ws.mb[kk] = (uint8_t)(hi - ws.ab[kk]);
    }
  } else if (radix) {
    if (kc > 16) throw MsmFail{MSM_ERR_INTERNAL, "more than 16 windows in a radix-split group"};
    for (int kk = 0; kk < kc; kk++) { ws.ab[kk] = (uint8_t)shift; ws.fb[kk] = (uint8_t)fb; }
  }
  uint32_t sortB = 1;
  uint64_t chunk = two_n;
  {
    // big inputs: finer slices also keep the round-1 gathers of neighbouring lanes inside one Infinity-Cache-sized
    // range of point rows (measured: 193 -> 183 ms at 2^26); small inputs: fewer, larger blocks (less fixed cost)
    uint64_t mult = two_n >= (1ull << 27) ? 8 : two_n >= (1ull << 24) ? 4 : 2;   // measured 2^21 .. 2^26
    if (three_pass) mult = std::min<uint64_t>(mult, 4);   // the chunk-ordered round 1 makes its own locality: fewer, larger slices
    MSM_KNOB(mult, "MSM_SORTB_MULT", 1);
    uint64_t want = std::max<uint64_t>(1, (mult * ctx->n_cu + kc - 1) / kc);
    uint64_t maxb = std::max<uint64_t>(1, two_n / 8192);
    sortB = (uint32_t)std::min<uint64_t>(want, maxb);
    chunk = (two_n + sortB - 1) / sortB;
    ctx->ensure(w.block_hist, (size_t)kc * sortB * (three_pass ? Lp : L) * 4 + 64);
  }
  const uint32_t* d_v2start = nullptr;   // three passes: starts of the fine windows in the record arrays
  const uint32_t* d_rec_dig = nullptr;   // records read by the last pass
  const uint32_t* d_rec_idx = nullptr;

  HIPCHK(hipEventRecord(w.ev[0], s));
  {
    uint32_t grid = (uint32_t)((n + 255) / 256);
    if (te)
      hipLaunchKernelGGL(te::k_te_digits, dim3(grid), dim3(256), 0, s, (uint32_t*)w.dig.p, d_scalars, (uint32_t)n, pl.c, pl.K,
                         k_lo, kc, pl.strict ? 1 : 0, (uint32_t*)ctx->errflag.p);
    else
      W_LAUNCH(ctx, k_digits, dim3(grid), dim3(256), 0, s, (uint32_t*)w.dig.p, d_scalars, (uint32_t)n, pl.c, pl.K, k_lo, kc,
                         (pl.no_glv ? 0 : 1) | (pl.fold ? 2 : 0), pl.strict ? 1 : 0, (uint32_t*)ctx->errflag.p);
  }
  HIPCHK(hipEventRecord(w.ev[1], s));
  if (!three_pass) {
    hipLaunchKernelGGL(k_hist, dim3(sortB, kc), dim3(SORT_THREADS), (size_t)L * 4, s, (uint32_t*)w.block_hist.p,
                       (const uint32_t*)w.dig.p, two_n, chunk, L, ws, 0u, 0u);
    hipLaunchKernelGGL(k_colscan, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, s, (uint32_t*)w.block_hist.p,
                       (uint32_t*)w.counts.p, sortB, L, (uint32_t)kc);
  } else {
    // totals of the fine windows -> their starts (= the starts of the mid and coarse bins too); pass A; pass M; bucket sizes
    const size_t n_off = (size_t)kc * sortB * 256;
    ctx->ensure(w.part, ((size_t)2 * V + 2 + n_off) * 4);
    uint32_t* d_v2tot = (uint32_t*)w.part.p;
    uint32_t* d_vs = d_v2tot + V;
    uint32_t* d_blk_off = d_vs + V + 1;
    ctx->ensure(w.dig2, n_entries * 4);
    ctx->ensure(w.idx2, n_entries * 4);
    ctx->ensure(w.idx3, n_entries * 4);
    hipLaunchKernelGGL(k_hist, dim3(sortB, kc), dim3(SORT_THREADS), (size_t)Lp * 4, s, (uint32_t*)w.block_hist.p,
                       (const uint32_t*)w.dig.p, two_n, chunk, Lp, ws, 1u, 0u);
    hipLaunchKernelGGL(k_colscan, dim3((V + 255) / 256), dim3(256), 0, s, (uint32_t*)w.block_hist.p, d_v2tot, sortB, Lp,
                       (uint32_t)kc);
    hipLaunchKernelGGL(k_coarse_offsets3, dim3((uint32_t)((n_off + 255) / 256)), dim3(256), 0, s, d_blk_off,
                       (const uint32_t*)w.block_hist.p, sortB, Lp, (uint32_t)kc, ws);
    hipLaunchKernelGGL(k_vscan, dim3(1), dim3(SCAN_THREADS), 0, s, d_vs, (const uint32_t*)d_v2tot, V);
    hipLaunchKernelGGL(k_radix_coarse, dim3(sortB, kc), dim3(RX_THREADS), 0, s, (uint32_t*)w.dig2.p, (uint32_t*)w.idx2.p,
                       (const uint32_t*)d_vs, (const uint32_t*)d_blk_off, (const uint32_t*)w.dig.p, two_n, chunk, Lp, 256u, ws);
    // the digits are dead now: the second record array reuses their buffer
    hipLaunchKernelGGL(k_radix_mid, dim3(256, kc), dim3(RXB_THREADS), 0, s, (uint32_t*)w.dig.p, (uint32_t*)w.idx3.p,
                       (const uint32_t*)d_vs, (const uint32_t*)w.dig2.p, (const uint32_t*)w.idx2.p, Lp, ws);
    HIPCHK(hipMemsetAsync(w.counts.p, 0, nb * 4, s));
    hipLaunchKernelGGL(k_fine_hist, dim3(V), dim3(256), 0, s, (uint32_t*)w.counts.p, (const uint32_t*)d_vs,
                       (const uint32_t*)w.dig.p, Lp, L, ws);
    d_v2start = d_vs;
    d_rec_dig = (const uint32_t*)w.dig.p;
    d_rec_idx = (const uint32_t*)w.idx3.p;
  }
  int RT = 0;
  uint64_t total_slots = 0;
  uint32_t max_bucket = 0;
  {
    // largest bucket -> number of tail rounds RT, then the multi-block scan of RT + 2 quantities.  The scan kernels take RT
    // from the device (pscan_nq), so ONE read-back behind them brings the largest bucket and the totals together.
    HIPCHK(hipMemsetAsync(w.info.p, 0, 64 * 4, s));
    hipLaunchKernelGGL(k_bucket_max, dim3((uint32_t)std::min<uint64_t>(1024, (nb + 255) / 256)), dim3(256), 0, s,
                       (const uint32_t*)w.counts.p, (uint32_t)nb, (uint32_t*)w.info.p);
    const uint32_t nblocks = (uint32_t)((nb + PS_SPAN - 1) / PS_SPAN);
    ctx->ensure(w.scan_partial, (size_t)PS_MAX_NQ * nblocks * 4);
    hipLaunchKernelGGL(k_pscan_partial, dim3(nblocks), dim3(PS_BLOCK), 0, s, (const uint32_t*)w.counts.p, (uint32_t)nb, logG,
                       (const uint32_t*)w.info.p, (uint32_t*)w.scan_partial.p, nblocks);
    hipLaunchKernelGGL(k_pscan_top, dim3(1), dim3(SCAN_THREADS), 0, s, (uint32_t*)w.scan_partial.p, nblocks, logG,
                       (uint32_t*)w.info.p);
    hipLaunchKernelGGL(k_pscan_final, dim3(nblocks), dim3(PS_BLOCK), 0, s, (const uint32_t*)w.counts.p, (uint32_t)nb, logG,
                       (const uint32_t*)w.scan_partial.p, nblocks, (uint32_t*)w.cursor.p, (uint32_t*)w.tail_off.p,
                       (const uint32_t*)w.info.p);
    HIPCHK(hipMemcpyAsync(w.h_info, w.info.p, 64 * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    total_slots = w.h_info[0];
    max_bucket = w.h_info[1];
    st.n_pairs_algo += (uint64_t)w.h_info[INFO_ALGO_PAIRS] | ((uint64_t)w.h_info[INFO_ALGO_PAIRS + 1] << 32);
    const uint32_t capmax = (max_bucket + (1u << logG) - 1) >> logG;
    while (RT < 32 && (1u << RT) < capmax) RT++;
  }
  st.max_bucket = std::max<uint64_t>(st.max_bucket, max_bucket);

  // scatter
  ctx->ensure(w.slots, std::max<uint64_t>(total_slots, 2) * 4);
  HIPCHK(hipMemsetAsync(w.slots.p, 0xFF, std::max<uint64_t>(total_slots, 2) * 4, s));
  if (radix) {
    // pass A: coarse split into dig2 / idx2; pass B: one block per virtual window, payloads to their padded slots
    ctx->ensure(w.part, ((size_t)kc * sortB * Hn + 2 * (size_t)V + 2) * 4);
    uint32_t* d_blk_off = (uint32_t*)w.part.p;
    uint32_t* d_vtot = d_blk_off + (size_t)kc * sortB * Hn;
    uint32_t* d_vstart = d_vtot + V;
    ctx->ensure(w.dig2, n_entries * 4);
    ctx->ensure(w.idx2, n_entries * 4);
    const uint64_t co = (uint64_t)kc * (sortB + 1) * Hn;
    hipLaunchKernelGGL(k_coarse_offsets, dim3((uint32_t)((co + 255) / 256)), dim3(256), 0, s, d_blk_off, d_vtot,
                       (const uint32_t*)w.block_hist.p, (const uint32_t*)w.counts.p, sortB, L, Hn, (uint32_t)kc);
    hipLaunchKernelGGL(k_vscan, dim3(1), dim3(SCAN_THREADS), 0, s, d_vstart, (const uint32_t*)d_vtot, V);
    hipLaunchKernelGGL(k_radix_coarse, dim3(sortB, kc), dim3(RX_THREADS), 0, s, (uint32_t*)w.dig2.p, (uint32_t*)w.idx2.p,
                       (const uint32_t*)d_vstart, (const uint32_t*)d_blk_off, (const uint32_t*)w.dig.p, two_n, chunk, Hn, Hn, ws);
    hipLaunchKernelGGL(k_radix_fine, dim3(V), dim3(RXB_THREADS), 0, s, (uint32_t*)w.slots.p, (const uint32_t*)w.cursor.p,
                       (const uint32_t*)d_vstart, (const uint32_t*)w.dig2.p, (const uint32_t*)w.idx2.p, Lp, L, ws);
  } else if (one_level) {
    hipLaunchKernelGGL(k_scatter_lds, dim3(sortB, kc), dim3(SORT_THREADS), (size_t)L * 4, s, (uint32_t*)w.slots.p,
                       (const uint32_t*)w.cursor.p, (const uint32_t*)w.block_hist.p, (const uint32_t*)w.dig.p, two_n,
                       chunk, L, 0u);
  } else {
    hipLaunchKernelGGL(k_radix_fine, dim3(V), dim3(RXB_THREADS), 0, s, (uint32_t*)w.slots.p, (const uint32_t*)w.cursor.p,
                       d_v2start, d_rec_dig, d_rec_idx, Lp, L, ws);
  }
  // Big windows over a big table: walk round 1 chunk by chunk of the point rows (k_chunk_order, sort_kernels.h).  Needed once
  // the 128 slots of a wave span more than ~1 GB of rows: 64 L rows of 256 bytes, i.e. from c = 18 with more than 2^22 points.
  const uint32_t* round1_slots = (const uint32_t*)w.slots.p;
  const uint16_t* round1_oidx = nullptr;
  uint64_t rec_y_off = 0;   // 12-word fields: where the y records of round 1's results start inside w.rows1
  bool chunked = false;   // round 1 walks chunk-ordered pairs and writes element records, round 2 reads them
  {
    long long chunk_rows_log = 22;   // 2^22 rows of 256 bytes = 1 GB
    MSM_KNOB(chunk_rows_log, "MSM_CHUNK_LOG", 10);
    long long want_chunks = (!te && pl.c >= 18 && n > (1ull << chunk_rows_log)) ? 1 : 0;
    MSM_KNOB(want_chunks, "MSM_CHUNKED", 0);
    const uint64_t nch = (n + (1ull << chunk_rows_log) - 1) >> chunk_rows_log;
    // (round 2 must be an index-free round to read the element records round 1 then writes: logG >= 2)
    if (want_chunks && !te && logG >= 2 && total_slots >= 2 && nch >= 2 && nch + 1 <= (uint64_t)CO_MAX_KEYS) {
      const uint64_t n_pairs = total_slots / 2;
      ctx->ensure(w.slots2, total_slots * 4);
      ctx->ensure(w.oidx, n_pairs * 2);
      rec_y_off = (n_pairs * 64 + 255) & ~(uint64_t)255;
      ctx->ensure(w.rows1, 2 * rec_y_off + 256);
      hipLaunchKernelGGL(k_chunk_order, dim3((uint32_t)((n_pairs + CO_PAIRS - 1) / CO_PAIRS)), dim3(CO_THREADS), 0, s,
                         (uint2*)w.slots2.p, (uint16_t*)w.oidx.p, (const uint2*)w.slots.p, n_pairs, (uint32_t)chunk_rows_log,
                         (uint32_t)nch + 1);
      round1_slots = (const uint32_t*)w.slots2.p;
      round1_oidx = (const uint16_t*)w.oidx.p;
      if (MSM_KNOB_SET("MSM_CHUNK_NOSTORE")) round1_oidx = nullptr;   // experiment (wrong sums): chunk-ordered loads, natural stores
      chunked = true;
    }
  }
  HIPCHK(hipEventRecord(w.ev[2], s));
  so.logG = logG;
  so.RT = RT;
  so.total_slots = total_slots;
  so.max_bucket = max_bucket;
  so.round1_slots = round1_slots;
  so.round1_oidx = round1_oidx;
  so.rec_y_off = rec_y_off;
  so.chunked = chunked;
}

}  // namespace msmi
