// Accumulation tree of one window group: rounds of batched-affine pair additions (k_batch_add; unified extended additions
// k_te_add on the Edwards path) over the padded slots the sort left, then k_bucket_finish.
// (reference: the rounds of src/msm-batched-affine.ts:243-282 over batchAddNew src/curve-affine.ts:376-522)
#include "msm_internal.h"

using namespace msm;
using namespace msmi;

namespace msmi {

// queues the tree on w.stream behind whatever is there; records w.ev[6] behind round 1 and w.ev[3] behind the last kernel
void accumulate_window_group(msm_ctx* ctx, msm_ctx::Workspace& w, const Plan& pl, int kc, uint64_t row_off, const SortOut& so,
                             GroupStats& st, TreeOut& to) {
  hipStream_t s = w.stream;
  const bool lone = pl.lone;
  const bool te = ctx->is_te();
  const uint32_t L = pl.L;
  const uint64_t nb = (uint64_t)kc * L;
  const size_t elem_bytes = te ? 128 : 96;  // tree node: extended (X, Y, Z, T) x 32 B, or affine (x, y) x 48 B
  const uint32_t logG = so.logG;
  const int RT = so.RT;
  const uint64_t total_slots = so.total_slots;
  const uint32_t max_bucket = so.max_bucket;
  const uint32_t* round1_slots = so.round1_slots;
  const uint32_t* round1_dest = so.round1_dest;
  const uint64_t rec_y_off = so.rec_y_off;
  const bool chunked = so.chunked;

  // accumulation tree
  // Weierstrass: tail rounds run only until no bucket holds more than FINISH_MAX elements; k_bucket_finish ends it
  uint32_t FINISH_MAX = pl.c >= 18 ? 64 : 32;
  MSM_KNOB(FINISH_MAX, "MSM_FINISH_MAX", 1);
  // big windows (millions of small buckets per group): the last descriptor rounds pay a binary search over all buckets per
  // pair and a launch each for a few million pair additions -- k_bucket_finish takes the last two or four elements of every
  // bucket cheaper (2^26, c = 22: 154.3 -> 150.8 ms; profiles/r04_experiments.txt item 8)
  uint32_t tail_min_pairs = pl.c >= 18 ? 1u << 23 : 1u << 21;
  MSM_KNOB(tail_min_pairs, "MSM_TAIL_MIN", 1);
  const bool use_finish = true;
  int r_stop = RT;
  if (use_finish) {
    // a tail round is worth its launch + inversion latency (~0.25 ms) only while it still has a few million pairs;
    // below that, and once no bucket holds more than FINISH_MAX elements, k_bucket_finish takes over
    uint32_t cap_elems = (max_bucket + (1u << logG) - 1) >> logG;   // largest bucket after the regular rounds
    r_stop = 0;
    while (r_stop < RT &&
           (((cap_elems + (1u << r_stop) - 1) >> r_stop) > FINISH_MAX || w.h_info[3 + r_stop + 1] >= tail_min_pairs))
      r_stop++;
  }
  // outputs alternate between two buffers: size each for the largest round it receives
  uint64_t capA = 1, capB = 1;
  {
    int which = 0;
    uint64_t cnt = total_slots;
    for (uint32_t r = 1; r <= logG; r++) {
      cnt /= 2;
      if (r == 1 && chunked) continue;   // element records (w.rows1), not planes
      (which ? capB : capA) = std::max<uint64_t>(which ? capB : capA, cnt);
      which ^= 1;
    }
    for (int r = 1; r <= r_stop; r++) {
      cnt = w.h_info[3 + r];
      (which ? capB : capA) = std::max<uint64_t>(which ? capB : capA, cnt);
      which ^= 1;
    }
  }
  // Idle lanes of a plane-reading round read (and ignore) elements past the end of its input: a round of n_out pairs runs
  // steps * T lanes-steps with T = 256 ceil(ceil(n_out / steps) / 256), so fewer than 257 * steps pairs -- 2 elements of 16
  // bytes per plane each -- lie beyond n_out, whatever the grid (hence the CU count) is.  steps <= 512.
  const size_t tree_slack = (size_t)2 * 257 * 512 * 16 + 4096;
  ctx->ensure(w.bufA, capA * elem_bytes + tree_slack);
  ctx->ensure(w.bufB, capB * elem_bytes + tree_slack);
  uint4* buf[2] = {(uint4*)w.bufA.p, (uint4*)w.bufB.p};
  uint64_t cap[2] = {capA, capB};
  int cur = 0;  // buffer that receives the next round's output
  uint64_t n_in = total_slots;
  int round = 0;
  const uint4* fin = buf[0];
  uint64_t fin_cap = cap[0];
  const uint32_t* off_fin = (const uint32_t*)w.tail_off.p;
  if (total_slots > 0) {
    for (uint32_t r = 1; r <= logG; r++) {
      uint64_t n_out = n_in / 2;
      RoundGeom g = round_geom(ctx, n_out, r == 1 || (r == 2 && chunked) || te, lone);   // no inversion on the Edwards path: always two waves
      const uint64_t sstride = g.T;
      if (!te) ctx->ensure(w.scratch, (size_t)g.steps * NL * sstride * 4);
      BatchArgs a{};
      // (on window tables the payloads count rows of the tables, which live in `rows` or, for a range of the points, in `tabs`)
      a.points = (pl.tables ? pl.tab_rows : (const uint32_t*)ctx->rows.p) + row_off * (te ? (uint64_t)te::TE_ROW_WORDS : (uint64_t)ROW_WORDS);
      a.slots = r == 1 ? round1_slots : (const uint32_t*)w.slots.p;
      a.dest = r == 1 ? round1_dest : nullptr;
      a.in = buf[cur ^ 1];
      a.in_cap = cap[cur ^ 1];
      a.out = buf[cur];
      a.out_cap = cap[cur];
      a.scratch = (uint32_t*)w.scratch.p;
      a.sstride = sstride;
      a.n_out = n_out;
      a.steps = g.steps;
      const bool rows_out = r == 1 && chunked, rows_in = r == 2 && chunked;
      a.y_off = 4 * ctx->nw();
      if (rows_out) a.out_rows = (uint32_t*)w.rows1.p;
      if (rows_in) { a.points = (const uint32_t*)w.rows1.p; a.slots = nullptr; }
      if (rows_out) a.out_y_off = rec_y_off;                      // 12-word fields: x records, then y records (batch_add.h)
      if (rows_in && ctx->nw() == 12) a.y_off = rec_y_off;
      if (r == 1 || rows_in) {
        if (te) hipLaunchKernelGGL(te::k_te_add<MODE_GATHER>, dim3(g.grid), dim3(256), 0, s, a);
        else W_LAUNCH_MODE(ctx, k_batch_add, MODE_GATHER, dim3(g.grid), dim3(256), 0, s, a);
        if (r == 1) HIPCHK(hipEventRecord(w.ev[6], s));
      } else {
        if (te) hipLaunchKernelGGL(te::k_te_add<MODE_REGULAR>, dim3(g.grid), dim3(256), 0, s, a);
        else W_LAUNCH_MODE(ctx, k_batch_add, MODE_REGULAR, dim3(g.grid), dim3(256), 0, s, a);
      }
      st.n_pairs += n_out;
      n_in = n_out;
      round++;
      if (rows_out) continue;   // the plane buffers have not been touched yet
      fin = buf[cur];
      fin_cap = cap[cur];
      cur ^= 1;
    }
    for (int r = 1; r <= r_stop; r++) {
      uint64_t n_out = w.h_info[3 + r];
      RoundGeom g = round_geom(ctx, n_out, te);
      const uint64_t sstride = g.T;
      if (!te) ctx->ensure(w.scratch, (size_t)g.steps * NL * sstride * 4);
      BatchArgs a{};
      a.in = buf[cur ^ 1];
      a.in_cap = cap[cur ^ 1];
      a.out = buf[cur];
      a.out_cap = cap[cur];
      a.scratch = (uint32_t*)w.scratch.p;
      a.sstride = sstride;
      a.n_out = n_out;
      a.steps = g.steps;
      if (n_out) {
        const uint32_t* off_in = (const uint32_t*)w.tail_off.p + (uint64_t)(r - 1) * (nb + 1);
        const uint32_t* off_out = (const uint32_t*)w.tail_off.p + (uint64_t)r * (nb + 1);
        ctx->ensure(w.desc, n_out * 4);
        hipLaunchKernelGGL(k_tail_desc, dim3((uint32_t)((n_out + 255) / 256)), dim3(256), 0, s, (uint32_t*)w.desc.p, off_in,
                           off_out, (uint32_t)nb, (uint32_t)n_out);
        a.desc = (const uint32_t*)w.desc.p;
        if (te) hipLaunchKernelGGL(te::k_te_add<MODE_SEARCH>, dim3(g.grid), dim3(256), 0, s, a);
        else W_LAUNCH_MODE(ctx, k_batch_add, MODE_SEARCH, dim3(g.grid), dim3(256), 0, s, a);
      }
      st.n_pairs += n_out;
      fin = buf[cur];
      fin_cap = cap[cur];
      cur ^= 1;
      round++;
    }
    off_fin = (const uint32_t*)w.tail_off.p + (uint64_t)r_stop * (nb + 1);
  }
  st.rounds += round;
  const uint32_t* bucket_proj = nullptr;
  if (use_finish && total_slots > 0) {
    ctx->ensure(w.bucket_proj, nb * (te ? 4 * te::TL : 3 * NL) * 4);
    // lanes of a wave should have equal trip counts: order the buckets by what they still hold
    const uint32_t* perm = nullptr;
    if (nb >= 4096) {
      ctx->ensure(w.blk_tab2, (nb + 2 * FINISH_BINS) * 4);
      uint32_t* hist = (uint32_t*)w.blk_tab2.p;
      HIPCHK(hipMemsetAsync(hist, 0, 2 * FINISH_BINS * 4, s));
      const uint32_t fgrid = (uint32_t)((nb + FINISH_THREADS - 1) / FINISH_THREADS);
      hipLaunchKernelGGL(k_finish_hist, dim3(fgrid), dim3(FINISH_THREADS), 0, s, off_fin, (uint32_t)nb, hist);
      hipLaunchKernelGGL(k_finish_perm, dim3(fgrid), dim3(FINISH_THREADS), 0, s, off_fin, (uint32_t)nb,
                         (const uint32_t*)hist, hist + FINISH_BINS, hist + 2 * FINISH_BINS);
      perm = hist + 2 * FINISH_BINS;
    }
    if (te)
      hipLaunchKernelGGL(te::k_te_bucket_finish, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, s, (uint32_t*)w.bucket_proj.p,
                         fin, fin_cap, off_fin, (uint32_t)nb, perm);
    else
      W_LAUNCH(ctx, k_bucket_finish, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, s, (uint32_t*)w.bucket_proj.p, fin,
                         fin_cap, off_fin, (uint32_t)nb, perm);
    bucket_proj = (const uint32_t*)w.bucket_proj.p;
  }
  if (total_slots == 0) HIPCHK(hipEventRecord(w.ev[6], s));
  HIPCHK(hipEventRecord(w.ev[3], s));
  to.fin = fin;
  to.fin_cap = fin_cap;
  to.off_fin = off_fin;
  to.bucket_proj = bucket_proj;
}

}  // namespace msmi
