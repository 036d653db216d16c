// Window tables: K resident tables of the point set, table k = 2^(c k) P, so that the K windows of an MSM share one set of
// buckets (k_table_next, msm_kernels.h).  The reference has no counterpart (4 GiB of wasm memory); it is what 288 GB of HBM are
// for: six tables of 2^26 points are 103 GB.  Built once per point set and plan -- by the first default-plan msm_run over the
// whole set, or ahead of it by msm_precompute / msm_reserve -- the way k_points_from_wire precomputes beta x once per set.
#include "msm_internal.h"

using namespace msm;
using namespace msmi;

namespace msmi {

static uint64_t table_bytes(const msm_ctx* ctx, uint64_t n, int K) {
  const uint64_t row_words = ctx->is_te() ? (uint64_t)te::TE_ROW_WORDS : (uint64_t)ROW_WORDS;
  return (uint64_t)K * std::max<uint64_t>(n, 1) * row_words * 4;
}

// tables of the WHOLE point set live in `rows` (table 0 = the plain rows); tables of a range [lo, lo + n) of the points in `tabs`
static void build_tables(msm_ctx* ctx, const Plan& pl, uint64_t lo, uint64_t n) {
  const bool whole = lo == 0 && n == ctx->n_points;
  const uint64_t row_words = ctx->is_te() ? (uint64_t)te::TE_ROW_WORDS : (uint64_t)ROW_WORDS;
  const uint64_t bytes = table_bytes(ctx, n, pl.K);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  auto ensure_or_retry = [&](DevBuf& b) {
    try {
      ctx->ensure(b, bytes);
    } catch (const HipFail& f) {
      // the workspaces of earlier calls only grow: give them back and try once more (the next MSM allocates what it needs)
      if (f.e != hipErrorOutOfMemory) throw;
      (void)hipGetLastError();
      release_workspaces(ctx);
      ctx->release(ctx->scal);
      ctx->ensure(b, bytes);
    }
  };
  uint32_t* rows = nullptr;
  if (whole) {
    ctx->release(ctx->tabs);   // (range tables of this set, if any, are replaced)
    if (ctx->rows.cap < bytes) {
      // a bigger buffer: table 0 (the plain rows) moves over, the old buffer goes back
      DevBuf big;
      ensure_or_retry(big);
      HIPCHK(hipMemcpyAsync(big.p, ctx->rows.p, n * row_words * 4, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      ctx->release(ctx->rows);
      ctx->rows = big;
    }
    rows = (uint32_t*)ctx->rows.p;
  } else {
    ctx->tab_c = ctx->tab_K = 0;
    ensure_or_retry(ctx->tabs);
    rows = (uint32_t*)ctx->tabs.p;
    HIPCHK(hipMemcpyAsync(rows, (const uint32_t*)ctx->rows.p + lo * row_words, n * row_words * 4, hipMemcpyDeviceToDevice, ctx->stream));
  }
  const uint32_t grid = (uint32_t)((n + 255) / 256);
  for (int k = 1; k < pl.K; k++) {
    uint32_t* out = rows + (uint64_t)k * n * row_words;
    const uint32_t* in = rows + (uint64_t)(k - 1) * n * row_words;
    if (ctx->is_te()) hipLaunchKernelGGL(te::k_te_table_next, dim3(grid), dim3(256), 0, ctx->stream, out, in, n, pl.c);
    else W_LAUNCH(ctx, k_table_next, dim3(grid), dim3(256), 0, ctx->stream, out, in, n, pl.c);
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  ctx->tab_c = pl.c;
  ctx->tab_K = pl.K;
  ctx->tab_lo = lo;
  ctx->tab_n = n;
}

// can this call run on window tables at all?
static bool tables_eligible(const msm_ctx* ctx, uint64_t n, const msm_opts* opts, bool placed) {
  if (placed || (opts && (opts->no_tables || opts->bucket_shards > 1))) return false;
  if (!ctx->children.empty()) return false;           // a device list shards by points or windows over plain rows
  const uint64_t lo = opts ? opts->point_lo : 0;
  return lo + n <= ctx->n_points && n >= 4096;
}
static bool whole_set(const msm_ctx* ctx, uint64_t n, const msm_opts* opts) { return n == ctx->n_points && !(opts && opts->point_lo); }
static bool tables_cover(const msm_ctx* ctx, uint64_t n, const msm_opts* opts) {
  return ctx->tab_K && ctx->tab_lo == (opts ? opts->point_lo : 0) && ctx->tab_n == n;
}

// entry indices of a merged window (one per table row and GLV half) travel in 31 bits of the sort's payloads
static bool tables_addressable(const msm_ctx* ctx, uint64_t n, int K) {
  return (uint64_t)K * (ctx->is_te() ? n : 2 * n) < (1ull << 31);
}

// A short top window would pile its entries on the lowest buckets of the merged window (a 2-bit top window: an eighth of all
// entries in four buckets): tables are built by default only for plans whose top window is about as wide as the others.
// (Since round 6 the sort cuts such bins into parts, so this is a matter of the tree's depth, no longer of one block's time.)
static bool plan_suits_tables(const Plan& pl) {
  if (pl.K < 2) return false;
  const int top_bits = pl.fold ? pl.c + 1 : pl.bits - (pl.K - 1) * pl.c;
  return top_bits >= pl.c - 3;
}

// Tables that are not there yet may be built by this call: those of the whole set at once (the first default-plan call over it,
// as since round 5); those of a RANGE of the points when the call comes back for the same range -- the rank of a points-split
// run does, a caller that walks over the shards on one GPU does not, and a build (c doublings and an inversion per point and
// table: nine MSMs' worth at 2^23 points) per call would cost it far more than the tables return.
static bool may_build_for(const msm_ctx* ctx, uint64_t n, const msm_opts* opts, int c) {
  if (whole_set(ctx, n, opts)) return true;
  if (ctx->tab_K && !ctx->tabs.p) return false;   // tables of the whole set stay
  return ctx->cand_n == n && ctx->cand_lo == (opts ? opts->point_lo : 0) && ctx->cand_c == c;
}

int make_run_plan(msm_ctx* ctx, uint64_t n, const msm_opts* opts, bool placed, Plan& pl, bool& tables_wanted, bool note_range) {
  tables_wanted = false;
  if (tables_eligible(ctx, n, opts, placed)) {
    // tables that exist decide: the call uses them if its plan is theirs (an explicit c, or the default plan they were built for)
    if (tables_cover(ctx, n, opts)) {
      Plan pt;
      msm_opts o;
      if (opts) o = *opts; else memset(&o, 0, sizeof o);
      if (!o.c) o.c = ctx->tab_c;
      if (make_plan(ctx, n, &o, pt) == MSM_OK && pt.c == ctx->tab_c && pt.K == ctx->tab_K &&
          (!(opts && opts->c) || opts->c == ctx->tab_c)) {
        // (a default-plan call keeps using tables built by msm_precompute for another c only if that is also what it would pick)
        Plan pd;
        if ((opts && opts->c) || (make_plan(ctx, n, opts, pd, true) == MSM_OK && pd.c == ctx->tab_c)) {
          pl = pt;
          tables_wanted = true;
          return MSM_OK;
        }
      }
    }
    // none yet (or others): a call with the default plan may build them if they fit the limit --
    // opts->c == 0, or the very window the library would pick (a facade that asks msm_plan first and hands its answer back)
    if (!(opts && opts->no_glv)) {
      Plan pt;
      msm_opts o;
      if (opts) o = *opts; else memset(&o, 0, sizeof o);
      const int asked = o.c;
      o.c = 0;
      if (make_plan(ctx, n, &o, pt, true) == MSM_OK && (asked == 0 || asked == pt.c) && plan_suits_tables(pt) &&
          table_bytes(ctx, n, pt.K) <= ctx->tables_limit && tables_addressable(ctx, n, pt.K) &&
          (whole_set(ctx, n, opts) || !(ctx->tab_K && !ctx->tabs.p))) {
        const bool build_now = may_build_for(ctx, n, opts, pt.c);
        if (note_range && !whole_set(ctx, n, opts)) {
          ctx->cand_lo = opts ? opts->point_lo : 0;
          ctx->cand_n = n;
          ctx->cand_c = pt.c;
        }
        // (msm_plan answers for the call that would build them: the plan of a rank's share of a points split is the tables' plan)
        if (build_now || !note_range) {
          pl = pt;
          tables_wanted = true;
          return MSM_OK;
        }
      }
    }
  }
  return make_plan(ctx, n, opts, pl);
}

bool use_window_tables(msm_ctx* ctx, uint64_t n, const msm_opts* opts, Plan& pl, bool may_build) {
  if (!tables_eligible(ctx, n, opts, false) || pl.K < 2) return false;
  const uint64_t lo = opts ? opts->point_lo : 0;
  if (!(tables_cover(ctx, n, opts) && ctx->tab_c == pl.c && ctx->tab_K == pl.K)) {
    if (!may_build || table_bytes(ctx, n, pl.K) > ctx->tables_limit || !tables_addressable(ctx, n, pl.K)) return false;
    if (!whole_set(ctx, n, opts) && ctx->tab_K && !ctx->tabs.p) return false;   // tables of the whole set stay
    build_tables(ctx, pl, lo, n);
  }
  pl.tab_rows = ctx->table_rows();
  pl.tab_lo = ctx->tab_lo;
  pl.tab_n = ctx->tab_n;
  return true;
}

}  // namespace msmi

extern "C" {

int msm_precompute(msm_ctx* ctx, uint64_t n, const msm_opts* opts) {
  if (!ctx) return MSM_ERR_ARG;
  const uint64_t lo = opts ? opts->point_lo : 0;
  if (lo + n > ctx->n_points || n == 0)
    return fail(ctx, MSM_ERR_ARG, "msm_precompute: points [%llu, +%llu) but %llu resident points", (unsigned long long)lo,
                (unsigned long long)n, (unsigned long long)ctx->n_points);
  Plan pl;
  if (make_plan(ctx, n, opts, pl, /*for_tables=*/true)) return fail(ctx, MSM_ERR_ARG, "msm_precompute: bad window size");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    if (!(n == ctx->n_points && lo == 0) && ctx->tab_K && !ctx->tabs.p) ctx->tab_c = ctx->tab_K = 0;   // asked for explicitly: a range replaces the whole set's
    (void)use_window_tables(ctx, n, opts, pl, /*may_build=*/true);   // not an error if they do not fit: the plain path stays
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_tables_info(const msm_ctx* ctx, int32_t* c_out, int32_t* K_out, uint64_t* bytes_out) {
  if (!ctx) return MSM_ERR_ARG;
  if (c_out) *c_out = ctx->tab_c;
  if (K_out) *K_out = ctx->tab_K;
  if (bytes_out) *bytes_out = ctx->tab_K ? table_bytes(ctx, ctx->tab_n, ctx->tab_K) : 0;
  return MSM_OK;
}

int msm_tables_range(const msm_ctx* ctx, uint64_t* point_lo_out, uint64_t* n_out) {
  if (!ctx) return MSM_ERR_ARG;
  if (point_lo_out) *point_lo_out = ctx->tab_K ? ctx->tab_lo : 0;
  if (n_out) *n_out = ctx->tab_K ? ctx->tab_n : 0;
  return MSM_OK;
}

int msm_set_tables_limit(msm_ctx* ctx, uint64_t bytes) {
  if (!ctx) return MSM_ERR_ARG;
  ctx->tables_limit = bytes;
  for (msm_ctx* c : ctx->children) c->tables_limit = bytes;
  return MSM_OK;
}

}  // extern "C"
