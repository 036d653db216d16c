// One MSM as window groups on the context's two streams, ranges of the points where a window does not fit or the scalars
// arrive over PCIe, and the fan-out over the devices of a multi-device context.
// (reference: the SPMD threads of src/msm-batched-affine.ts:285-340; windows are independent until :312-333)
#include "msm_internal.h"

using namespace msm;
using namespace msmi;

namespace {

// Partition sums P_k for windows [k_lo, k_hi) over the points [p_lo, p_lo + n) -> h_partials_out[(k - k_lo) * 36 ...]
// scalars: device pointer, n x 8 words.
// k_base: the first window of the CALL (window tables carry weights relative to it: table j = 2^(c j) P serves window k_base + j)
void run_window_group(msm_ctx* ctx, msm_ctx::Workspace& w, const uint32_t* d_scalars_all, uint64_t p_lo, uint64_t n, const Plan& pl,
                      int k_lo, int k_hi, int k_base, uint32_t* h_partials_out, GroupStats& st, uint64_t p_off = 0,
                      GroupDigits* share = nullptr) {
  hipStream_t s = w.stream;
  const uint32_t* d_scalars = d_scalars_all + p_lo * 8;   // scalar i of the call <-> resident point p_off + i
  p_lo += p_off;
  const int kc = k_hi - k_lo;
  SortOut so;
  HIPCHK(hipEventRecord(w.ev[0], s));
  sort_window_group(ctx, w, d_scalars, n, pl, k_lo, k_hi, st, so, share);
  st.max_bucket = std::max<uint64_t>(st.max_bucket, so.max_bucket);
  HIPCHK(hipEventRecord(w.ev[5], s));   // the tree starts here
  TreeOut to;
  if (pl.tables) {
    // one merged window over the tables k_lo - k_base .. k_hi - k_base - 1: its sum carries the windows' weights (relative to the
    // call's first window) already.  It goes into the group's first slot, identities into the others.
    accumulate_window_group(ctx, w, pl, 1, (uint64_t)(k_lo - k_base) * pl.tab_n, so, st, to);
    reduce_buckets(ctx, w, to.fin, to.fin_cap, to.off_fin, to.bucket_proj, pl.L, 1, h_partials_out, pl.merged, pl.c);
    const int pw = ctx->is_te() ? 32 : 36;
    for (int kk = 1; kk < kc; kk++) {
      if (ctx->is_te()) te_host_to_partial(ctx, ctx->hte.zero(), h_partials_out + (size_t)kk * pw);
      else memset(h_partials_out + (size_t)kk * pw, 0, (size_t)pw * 4);
    }
  } else {
    accumulate_window_group(ctx, w, pl, kc, p_lo, so, st, to);
    reduce_buckets(ctx, w, to.fin, to.fin_cap, to.off_fin, to.bucket_proj, pl.L, kc, h_partials_out, pl.merged, pl.c);
  }
  float ms;
  HIPCHK(hipEventElapsedTime(&ms, w.ev[0], w.ev[1])); st.ms_digits += ms;
  HIPCHK(hipEventElapsedTime(&ms, w.ev[1], w.ev[2])); st.ms_sort += ms;
  HIPCHK(hipEventElapsedTime(&ms, w.ev[5], w.ev[3])); st.ms_acc += ms;
  HIPCHK(hipEventElapsedTime(&ms, w.ev[5], w.ev[6])); st.ms_r1 += ms;
  HIPCHK(hipEventElapsedTime(&ms, w.ev[3], w.ev[4])); st.ms_red += ms;
}

// windows [k_lo, k_hi) over the resident points [p_off, p_off + n); scalars[i] belongs to point p_off + i
int window_sums_once(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const msm_opts* opts, int k_lo, int k_hi,
                     const Plan& pl_in, std::vector<uint32_t>& words, msm_result* stats, uint64_t p_off) {
  Plan pl = pl_in;   // (a call that turns out to run over ranges of the points leaves the window tables: see below)
  const uint32_t* d_scal = nullptr;
  HIPCHK(hipEventRecord(ctx->ev[8], ctx->stream));
  // Host scalars of a big call cross PCIe BEHIND the computation, range by range of the points (PieceUpload); everything
  // else is staged before the window groups start.
  std::vector<uint64_t> piece_end;   // pipelined upload: point index where piece q ends (the last = n)
  if (!on_device && n >= (1ull << 24)) {
    // The link moves scalars ~4x as fast as the GPU consumes them (2 GB in ~40 ms against ~154 ms of MSM at 2^26), so
    // every range may be ~4x its predecessor and still arrive before the GPU is done with the one before: 1/16, 3/16, the
    // rest from 2^25 points; 1/8, 3/8, the rest below.  The first range is what the GPU waits for (2-3 ms); few ranges keep
    // the sub-MSMs near full-size efficiency.
    const uint64_t gran = msm_ctx::STAGE_CHUNK / 32;   // scalars per staging chunk
    const int big = n >= (1ull << 25);
    for (int sh : {big ? 4 : 3, big ? 2 : 1}) piece_end.push_back(((n >> sh) / gran) * gran);
    piece_end.push_back(n);
    ctx->ensure(ctx->scal, n * 32);   // before the workspace budget is taken from what the device has free
  }
  std::unique_ptr<PieceUpload> pipe;
  if (piece_end.empty()) stage_scalars(ctx, scalars, n, on_device, &d_scal);
  HIPCHK(hipEventRecord(ctx->ev[9], ctx->stream));
  GroupStats st;
  const int pw = ctx->is_te() ? 32 : 36;
  words.assign((size_t)(k_hi - k_lo) * pw, 0);
  // window groups: as large as the workspace budget allows; for big inputs two of them on two streams.  The streams
  // run in step (both sort, both gather, ...): what the second one buys is two tree kernels sharing the chip -- forward
  // (memory-heavy) and backward (issue-heavy) sweeps of different waves mix, the small last rounds fill each other's
  // idle CUs -- not a sort hidden under an accumulation (a sort started under the other group's tree finds no free
  // registers on any CU and takes four times as long: profiles/r04_experiments.txt item 1)
  if (ctx->ws_limit) {
    ctx->ws_budget = ctx->ws_limit;
  } else if (n >= (1ull << 22)) {
    // big inputs: the budget is what the device has free NOW (point sets, scalar buffers and other contexts have come and
    // gone since the context was made) plus what the workspaces already hold
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    uint64_t held = 0;
    for (auto& w : ctx->ws)
      for (DevBuf* b : w.all) held += b->cap;
    ctx->ws_budget = (uint64_t)((free_b + held) * 0.85L);
  }
  int wpg = std::min(windows_per_group(ctx, n, pl), 128);
  // the radix-split and three-pass sorts describe their windows in a WinSplit of 16 entries (sort_kernels.h): a group that
  // may take one of them holds at most 16 windows (msmProjective with a small explicit window: K = 17 .. 29 at c = 15 .. 9)
  // (the one-level sort of small inputs -- a window's counters fit the LDS and fewer than 2^22 entries per window -- has no
  // such table: Ed-on-BLS12-377 at 2^20 keeps its 18 windows in one group)
  {
    const uint64_t entries = ctx->is_te() ? n : 2 * n;
    const bool fits_lds = ((size_t)pl.L * 4 <= 128 * 1024);
    if (pl.c - 1 > (int)RX_FINE_BITS && (!fits_lds || entries >= (ctx->is_te() ? 1ull << 22 : 1ull << 21))) wpg = std::min(wpg, 16);
    // (on window tables the merged window of a group may take the bin split whatever a single digit window would have taken,
    // and the digit kernel describes the fine bits of at most 16 windows: pack_fine_bits)
    if (pl.tables) wpg = std::min(wpg, 16);
  }
  const int nwin = k_hi - k_lo;
  // measured on MI355X: two groups win 14 % at 2^23 / 2^24, 3 % at 2^22, nothing at 2^21 -- below that the fixed
  // per-group latencies (read-backs, bucket reduction depth) cost more than the overlap returns
  // (on window tables from 2^21: 5.87 -> 5.73 ms, Edwards 3.96 -> 3.72; the plain path at 2^21 prefers one group, 6.71 / 6.88; at
  // 2^20 one group wins on tables too, 3.24 / 3.33 -- round 5, tools/knob_sweep.sh MSM_GROUPS)
  // (round 6: the Edwards path on tables from 2^20 -- fifteen digit windows in one group leave the chip to one stream's ramps:
  // 2.19 - 2.24 -> 2.13 - 2.18 ms; BLS12-377 at 2^20 is level, 3.44 / 3.41, and stays on one group)
  int want_groups = (nwin >= 2 && (n >= (1ull << 22) || (pl.tables && n >= (ctx->is_te() ? 1ull << 20 : 1ull << 21)))) ? 2 : 1;
  MSM_KNOB(want_groups, "MSM_GROUPS", 1);
  wpg = std::max(1, std::min(wpg, (nwin + want_groups - 1) / want_groups));
  struct Group {
    int ka, kb;
    uint64_t p_lo, p_n;
    int piece;   // pipelined upload: the piece whose arrival the group waits for (-1: the scalars are in place)
  };
  std::vector<Group> groups;
  // A single window (the 8-GPU shard) has no second window group to hide its sort and tails under: split it by
  // points instead -- two half-size sub-MSMs of the same window on the two streams, their sums added on the host.
  // The same split serves inputs whose single window no longer fits the workspace budget (2^29 points: 165 GB per window
  // at c = 22 next to a 137 GB row table): every window runs over as many ranges of the points as it takes, one after the
  // other on the two streams, and the sums of its ranges are added on the host.
  uint64_t pieces = 1;
  if (nwin == 1 && want_groups == 1 && !ctx->is_te() && n >= (1ull << 24) && !MSM_KNOB_SET("MSM_GROUPS")) pieces = 2;
  pieces = std::max(pieces, point_pieces(ctx, n, pl));
  if (!piece_end.empty() && point_pieces(ctx, n, pl) > 1) {
    // the workspace forces its own ranges: plain staged upload first (rare: 2^29 points, or a tight msm_set_workspace_limit)
    piece_end.clear();
    stage_scalars(ctx, scalars, n, on_device, &d_scal);
  }
  if (!piece_end.empty()) {
    // pipelined host scalars: per arriving range of the points the usual window groups (two above 2^22 points), in order
    ctx->ensure(ctx->scal, n * 32);
    d_scal = (const uint32_t*)ctx->scal.p;
    uint64_t lo = 0;
    for (size_t q = 0; q < piece_end.size(); q++) {
      const uint64_t cnt = piece_end[q] - lo;
      const int g = (nwin >= 2 && cnt >= (1ull << 22)) ? 2 : 1;
      const int per = std::max(1, std::min(wpg, (nwin + g - 1) / g));
      for (int k = k_lo; k < k_hi; k += per) groups.push_back({k, std::min(k_hi, k + per), lo, cnt, (int)q});
      lo = piece_end[q];
    }
  } else if (pieces > 1) {
    for (int k = k_lo; k < k_hi; k++)
      for (uint64_t q = 0; q < pieces; q++) {
        const uint64_t lo = n * q / pieces, hi = n * (q + 1) / pieces;
        groups.push_back({k, k + 1, lo, hi - lo, -1});
      }
  } else {
    for (int k = k_lo; k < k_hi; k += wpg) groups.push_back({k, std::min(k_hi, k + wpg), 0, n, -1});
  }
  // does more than one group contribute to a window?  Then the sums of its ranges are added on the host below.
  bool split_points = false;
  for (const Group& g : groups) split_points |= g.p_n != n;
  // Window tables address row k * tab_n + i of the points they cover from the entry index alone, which counts from the GROUP's
  // first point and in units of the group's own n: a group over another range of the points (a tight workspace limit, the retry
  // after an out-of-memory error, host scalars arriving range by range) would read other points' rows.  Such a call runs the
  // plain path under the same window -- table 0 is the plain row table -- and hands back one sum per window slot, which the
  // caller's Horner step takes like the one weighted sum of a run on tables.
  if (pl.tables && (split_points || n != pl.tab_n || p_off != pl.tab_lo)) pl.tables = false;
  std::vector<std::vector<uint32_t>> split_part((split_points || pl.tables) ? groups.size() : 0);
  HIPCHK(hipMemsetAsync(ctx->errflag.p, 0, 4, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));   // staged scalars are in place before the group streams start
  if (!piece_end.empty()) {
    std::vector<size_t> ends;
    for (uint64_t e : piece_end) ends.push_back((size_t)e * 32);
    pipe.reset(new PieceUpload(ctx, ctx->scal.p, scalars, n * 32, ends));
  }
  // The two window groups of a call slice the same scalars: one launch of the digit kernel (on the first workspace's stream)
  // writes the digits and slice histograms of both -- one GLV decomposition per scalar instead of two (2^26: the two
  // concurrent launches took 2.3 ms, the one takes 1.5) -- and each group then takes its part (GroupDigits, msm_sort.hip).
  GroupDigits share;
  if (groups.size() == 2 && !ctx->is_te() && groups[0].piece < 0 && groups[1].piece < 0 &&
      groups[0].p_lo == groups[1].p_lo && groups[0].p_n == groups[1].p_n && groups[0].kb == groups[1].ka &&
      groups[1].kb - groups[0].ka <= 16) {   // (the digit kernel describes up to 16 windows: WinSplit)
    share.produce = true;
    share.ready = ctx->ev_dig[1];
    SortOut none;
    GroupStats gs;
    HIPCHK(hipEventRecord(ctx->ev_dig[0], ctx->ws[0].stream));
    sort_window_group(ctx, ctx->ws[0], d_scal + groups[0].p_lo * 8, groups[0].p_n, pl, groups[0].ka, groups[1].kb, gs, none, &share);
    share.produce = false;
  }
  std::atomic<int> next{0};
  GroupStats sts[msm_ctx::N_WS];
  const int nthreads = (opts && opts->serial) ? 1 : std::min<int>(msm_ctx::N_WS, (int)groups.size());
  auto worker = [&](int slot) {
    HIPCHK(hipSetDevice(ctx->device));
    for (;;) {
      int gi = next.fetch_add(1);
      if (gi >= (int)groups.size()) break;
      const int ka = groups[gi].ka, kb = groups[gi].kb;
      std::vector<uint32_t> part((size_t)(kb - ka) * pw);
      Plan pg = pl;
      // a launch that has the chip to itself -- the one-window shard, or every launch of a serialised call (msm_opts.serial,
      // the exclusive timing of the roofline) -- walks its pairs in four short batches instead of one long one (round_geom)
      pg.lone = (groups.size() == 1 && (kb - ka == 1 || pl.tables)) || (opts && opts->serial);
      if (groups[gi].piece >= 0) pipe->wait_piece(groups[gi].piece, ctx->ws[slot].stream);
      run_window_group(ctx, ctx->ws[slot], d_scal, groups[gi].p_lo, groups[gi].p_n, pg, ka, kb, k_lo, part.data(), sts[slot], p_off,
                       share.valid ? &share : nullptr);
      if (split_points || pl.tables) split_part[gi] = part;
      else memcpy(&words[(size_t)(ka - k_lo) * pw], part.data(), part.size() * 4);
    }
  };
  {
    // Whatever either worker throws (HIP failure, bad_alloc, ...) is re-raised here only after BOTH have stopped and both
    // group streams are idle: no queued kernel of a failed call may still run when the context is used again.
    std::exception_ptr err;
    if (nthreads > 1) ctx->helper->run([&] { worker(1); });
    try { worker(0); } catch (...) { err = std::current_exception(); }
    if (nthreads > 1) {
      try { ctx->helper->wait(); } catch (...) { if (!err) err = std::current_exception(); }
    }
    if (err) {
      next.store((int)groups.size());
      for (auto& w : ctx->ws) (void)hipStreamSynchronize(w.stream);
      std::rethrow_exception(err);
    }
  }
  {
    // scalars >= q seen by k_digits: refused under msm_opts.strict (otherwise they were reduced mod q)
    HIPCHK(hipMemcpyAsync(ctx->h_info, ctx->errflag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (pl.strict && (ctx->h_info[0] & 4u)) throw MsmFail{MSM_ERR_SCALAR, "a scalar is >= the group order q (msm_opts.strict)"};
    if (ctx->h_info[0] & 8u) throw MsmFail{MSM_ERR_INTERNAL, "a digit of the folded top window exceeds its bucket range (GLV bound violated)"};
  }
  float upload_ms = -1;
  if (pipe) upload_ms = pipe->finish();   // joins the staging threads; their last copy is done
  if (pl.tables) {
    // every group's first slot holds the sum of its windows WITH their weights: the call's sum is their plain sum, kept in
    // slot 0 (identities elsewhere: the caller's Horner step over such slots would return the same element)
    if (ctx->is_te()) {
      msm_host::Ext6 acc = ctx->hte.zero();
      for (size_t gi = 0; gi < groups.size(); gi++)
        if (!split_part[gi].empty()) acc = ctx->hte.add(acc, te_partial_to_host(ctx, split_part[gi].data()));
      for (int k = k_lo; k < k_hi; k++) te_host_to_partial(ctx, k == k_lo ? acc : ctx->hte.zero(), &words[(size_t)(k - k_lo) * pw]);
    } else {
      msm_host::Proj6 acc = ctx->hc.zero();
      for (size_t gi = 0; gi < groups.size(); gi++)
        if (!split_part[gi].empty()) acc = ctx->hc.add(acc, partial_to_host(ctx, split_part[gi].data()));
      std::fill(words.begin(), words.end(), 0u);
      host_to_partial(ctx, acc, words.data());
    }
  } else if (split_points) {
    // P_k = sum over the ranges of the points (groups of one or several windows each); an all-zero partial (Z = 0) is the
    // identity.  (Plan.merged: a group then carries sum_kk 2^(c kk) P_kk in its first slot and identities in the others --
    // slot-wise sums of such groups are still a valid set of slots for the Horner step.)
    for (int k = k_lo; k < k_hi; k++) {
      uint32_t* out = &words[(size_t)(k - k_lo) * pw];
      if (ctx->is_te()) {
        msm_host::Ext6 acc = ctx->hte.zero();
        for (size_t gi = 0; gi < groups.size(); gi++)
          if (groups[gi].ka <= k && k < groups[gi].kb && !split_part[gi].empty())
            acc = ctx->hte.add(acc, te_partial_to_host(ctx, split_part[gi].data() + (size_t)(k - groups[gi].ka) * pw));
        te_host_to_partial(ctx, acc, out);
      } else {
        msm_host::Proj6 acc = ctx->hc.zero();
        for (size_t gi = 0; gi < groups.size(); gi++)
          if (groups[gi].ka <= k && k < groups[gi].kb && !split_part[gi].empty())
            acc = ctx->hc.add(acc, partial_to_host(ctx, split_part[gi].data() + (size_t)(k - groups[gi].ka) * pw));
        host_to_partial(ctx, acc, out);
      }
    }
  }
  if (share.valid) {
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev_dig[0], ctx->ev_dig[1]));
    st.ms_digits += ms;
  }
  for (int i = 0; i < msm_ctx::N_WS; i++) {
    st.n_pairs += sts[i].n_pairs;
    st.n_pairs_algo += sts[i].n_pairs_algo;
    st.max_bucket = std::max(st.max_bucket, sts[i].max_bucket);
    st.rounds += sts[i].rounds;   // tree rounds (k_batch_add launches) of ALL window groups, like n_pairs and ms_acc
    st.ms_digits += sts[i].ms_digits; st.ms_sort += sts[i].ms_sort; st.ms_acc += sts[i].ms_acc;
    st.ms_red += sts[i].ms_red; st.ms_r1 += sts[i].ms_r1;
  }
  HIPCHK(hipEventRecord(ctx->ev[10], ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (stats) {
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[8], ctx->ev[9]));
    stats->phase_ms[MSM_T_UPLOAD] = upload_ms >= 0 ? upload_ms : ms;   // pipelined: host clock of the background transfer
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[8], ctx->ev[10]));
    stats->phase_ms[MSM_T_TOTAL] = ms;
    stats->phase_ms[MSM_T_DIGITS] = st.ms_digits;
    stats->phase_ms[MSM_T_SORT] = st.ms_sort;
    stats->phase_ms[MSM_T_ACCUMULATE] = st.ms_acc;
    stats->phase_ms[MSM_T_ACC_ROUND1] = st.ms_r1;
    stats->phase_ms[MSM_T_REDUCE] = st.ms_red;
    stats->n_pairs = st.n_pairs;
    stats->n_pairs_algo = st.n_pairs_algo;
    stats->max_bucket = st.max_bucket;
    stats->rounds = st.rounds;
    stats->c = pl.c;
    stats->K = pl.K;
    stats->tables = pl.tables ? 1 : 0;
  }
  return MSM_OK;
}

}  // namespace

namespace msmi {

// Workspace buffers only grow, and a call with another shape (window size, curve of the point set, sort path) leaves buffers
// behind that the next shape does not use: if the device runs out of memory the workspaces are dropped and the call runs
// once more from a clean slate, where the budget model of window_sums_once holds again.
int window_sums_impl(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const msm_opts* opts, int k_lo, int k_hi,
                     const Plan& pl, std::vector<uint32_t>& words, msm_result* stats, uint64_t p_off) {
  // The budget model is an estimate and other contexts may take memory while the call runs, so one clean-slate retry is not a
  // guarantee: every further attempt also halves what the workspaces may take (more window groups, then ranges of the points),
  // which trades time for memory as include/msm_hip.h promises.  The caller's own limit is restored afterwards.
  const uint64_t limit0 = ctx->ws_limit;
  for (int attempt = 0;; attempt++) {
    try {
      const int rc = window_sums_once(ctx, scalars, n, on_device, opts, k_lo, k_hi, pl, words, stats, p_off);
      ctx->ws_limit = limit0;
      return rc;
    } catch (const HipFail& f) {
      if (f.e != hipErrorOutOfMemory || attempt >= 4) {
        ctx->ws_limit = limit0;
        throw;
      }
    } catch (...) {
      ctx->ws_limit = limit0;
      throw;
    }
    (void)hipGetLastError();
    release_workspaces(ctx);
    if (attempt >= 1) ctx->ws_limit = std::max<uint64_t>(ctx->ws_budget / 2, (uint64_t)64 << 20);
  }
}

}  // namespace msmi

namespace {
// Multi-device context: one MSM over the devices of the list, each from its own host thread on its own context.
//   by points (default): device d runs ALL windows [k_lo, k_hi) on its share [n d / G, n (d + 1) / G) of the points and needs
//     only that share of the scalars; the G sums of every window are added on the host (G - 1 projective additions each);
//   by window (msm_opts.by_window): the window range is cut into contiguous shards (windows are independent until the Horner
//     step, src/msm-batched-affine.ts:312-333), every device needs all n scalars.
// Scalars: `placed` != nullptr -- one device pointer per device, already on that device (by points: the device's share);
// else a host buffer, of which every device uploads what it needs, or a device buffer on devices[0], of which the other
// devices first copy their part peer-to-peer -- all devices at once, each from its own thread, under device 0's shard.
int multi_window_sums(msm_ctx* ctx, const void* scalars, const void* const* placed, uint64_t n, int on_device, const msm_opts* opts,
                      int k_lo, int k_hi, const Plan& pl, std::vector<uint32_t>& words, msm_result* stats, uint64_t p_off) {
  const int ndev = 1 + (int)ctx->children.size();
  const int nwin = k_hi - k_lo, pw = ctx->is_te() ? 32 : 36;
  const bool by_window = opts && opts->by_window;
  words.assign((size_t)nwin * pw, 0);
  std::vector<int> lo(ndev, k_lo), hi(ndev, k_hi);
  std::vector<uint64_t> p0(ndev, 0), pn(ndev, n);
  for (int d = 0, k = k_lo; d < ndev; d++) {
    if (by_window) {
      const int cnt = nwin / ndev + (d < nwin % ndev ? 1 : 0);
      lo[d] = k;
      hi[d] = k + cnt;
      k += cnt;
    } else {
      p0[d] = n * (uint64_t)d / ndev;
      pn[d] = n * (uint64_t)(d + 1) / ndev - p0[d];
    }
  }
  std::vector<std::vector<uint32_t>> part(ndev);
  std::vector<msm_result> st(ndev);
  for (auto& r : st) memset(&r, 0, sizeof r);
  auto shard = [&](int d) {
    if (hi[d] <= lo[d] || pn[d] == 0) return;
    msm_ctx* c = d == 0 ? ctx : ctx->children[d - 1];
    HIPCHK(hipSetDevice(c->device));
    const void* sc;
    int dev_side = on_device;
    if (placed) {
      sc = placed[d];
      dev_side = 1;
    } else {
      sc = (const uint8_t*)scalars + p0[d] * 32;
      if (on_device && d > 0) {
        c->ensure(c->scal, pn[d] * 32);
        HIPCHK(hipMemcpyPeerAsync(c->scal.p, c->device, sc, ctx->device, pn[d] * 32, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        sc = c->scal.p;
      }
    }
    window_sums_impl(c, sc, pn[d], dev_side, opts, lo[d], hi[d], pl, part[d], &st[d], p_off + p0[d]);
  };
  std::exception_ptr err;
  for (int d = 1; d < ndev; d++) ctx->fan[d - 1]->run([&, d] { shard(d); });
  try { shard(0); } catch (...) { err = std::current_exception(); }
  for (int d = 1; d < ndev; d++) {
    try { ctx->fan[d - 1]->wait(); } catch (...) { if (!err) err = std::current_exception(); }
  }
  HIPCHK(hipSetDevice(ctx->device));
  if (err) std::rethrow_exception(err);
  if (by_window) {
    for (int d = 0; d < ndev; d++)
      if (hi[d] > lo[d]) memcpy(&words[(size_t)(lo[d] - k_lo) * pw], part[d].data(), part[d].size() * 4);
  } else {
    // P_k = sum over the devices; an all-zero partial (Z = 0) is the identity, a device without points has none at all
    for (int k = 0; k < nwin; k++) {
      if (ctx->is_te()) {
        msm_host::Ext6 acc = ctx->hte.zero();
        for (int d = 0; d < ndev; d++)
          if (!part[d].empty()) acc = ctx->hte.add(acc, te_partial_to_host(ctx, &part[d][(size_t)k * pw]));
        te_host_to_partial(ctx, acc, &words[(size_t)k * pw]);
      } else {
        msm_host::Proj6 acc = ctx->hc.zero();
        for (int d = 0; d < ndev; d++)
          if (!part[d].empty()) acc = ctx->hc.add(acc, partial_to_host(ctx, &part[d][(size_t)k * pw]));
        host_to_partial(ctx, acc, &words[(size_t)k * pw]);
      }
    }
  }
  if (stats) {
    for (int d = 0; d < ndev; d++) {
      stats->n_pairs += st[d].n_pairs;
      stats->n_pairs_algo += st[d].n_pairs_algo;
      stats->rounds += st[d].rounds;
      stats->max_bucket = std::max(stats->max_bucket, st[d].max_bucket);
      for (int j = 0; j < MSM_N_PHASES; j++) stats->phase_ms[j] = std::max(stats->phase_ms[j], st[d].phase_ms[j]);
    }
    stats->c = pl.c;
    stats->K = pl.K;
  }
  return MSM_OK;
}

}  // namespace

namespace msmi {
// the one entry the ABI functions use: single- or multi-device
int any_window_sums(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const msm_opts* opts, int k_lo, int k_hi,
                    const Plan& pl, std::vector<uint32_t>& words, msm_result* stats, const void* const* placed) {
  const uint64_t p_off = opts ? opts->point_lo : 0;
  if (ctx->children.empty())
    return window_sums_impl(ctx, placed ? placed[0] : scalars, n, placed ? 1 : on_device, opts, k_lo, k_hi, pl, words, stats, p_off);
  return multi_window_sums(ctx, scalars, placed, n, on_device, opts, k_lo, k_hi, pl, words, stats, p_off);
}

}  // namespace msmi
