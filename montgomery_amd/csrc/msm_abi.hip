// The C ABI of include/msm_hip.h: contexts, resident point sets, msm_run / msm_window_sums / msm_combine, device buffers.
// (reference interface replaced: Curve.Parallel.* of src/parallel.ts:135-145,251-259)
#include "msm_internal.h"
#include <chrono>

using namespace msm;
using namespace msmi;

extern "C" {

uint32_t msm_abi_version(void) { return MSM_ABI_VERSION; }
uint32_t msm_abi_struct_bytes(int which) { return which == 0 ? (uint32_t)sizeof(msm_opts) : which == 1 ? (uint32_t)sizeof(msm_result) : 0u; }

int msm_ctx_create(msm_ctx** out, int curve, int device) {
  if (!out) return MSM_ERR_ARG;
  *out = nullptr;
  if (curve != MSM_CURVE_BLS12_377_G1 && curve != MSM_CURVE_ED_ON_BLS12_377 && curve != MSM_CURVE_BLS12_381_G1 &&
      curve != MSM_CURVE_PALLAS)
    return MSM_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return MSM_ERR_NO_DEVICE;
  msm_ctx* ctx = new (std::nothrow) msm_ctx();
  if (!ctx) return MSM_ERR_INTERNAL;
  ctx->curve = curve;
  ctx->device = device;
  try {
    ctx->helper.reset(new HelperThread());
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    for (auto& e : ctx->ev) HIPCHK(hipEventCreate(&e));
    for (auto& e : ctx->ev_dig) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipHostMalloc((void**)&ctx->h_info, 64 * 4, hipHostMallocDefault));
    for (auto& w : ctx->ws) {
      HIPCHK(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
      HIPCHK(hipStreamCreateWithFlags(&w.side, hipStreamNonBlocking));
      for (auto& e : w.ev) HIPCHK(hipEventCreate(&e));
      HIPCHK(hipHostMalloc((void**)&w.h_info, 64 * 4, hipHostMallocDefault));
      HIPCHK(hipHostMalloc((void**)&w.h_part, 128 * 20 * 36 * 4, hipHostMallocDefault));
    }
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    // leave room for the resident points (144 B/point at 2^26 = 9.7 GB) and fragmentation
    ctx->ws_budget = (uint64_t)(free_b * 0.55);
    ctx->tables_limit = (uint64_t)(total_b * 0.10);   // window tables of one point set (msm_set_tables_limit): up to 2^24 points
    ctx->ensure(ctx->errflag, 16);
    sort_kernel_attributes();
  } catch (const HipFail& f) {
    fprintf(stderr, "msm_ctx_create: HIP error %s at line %d\n", hipGetErrorString(f.e), f.line);
    msm_ctx_destroy(ctx);
    return MSM_ERR_HIP;
  } catch (...) {
    msm_ctx_destroy(ctx);
    return MSM_ERR_INTERNAL;
  }
  ctx->hc.F.init(curve == MSM_CURVE_ED_ON_BLS12_377 ? Fp377::PW : curve_info(curve).pw);   // (the Edwards context uses hte)
  ctx->k_dev_to_host = ctx->hc.F.pow2(2 * ctx->hc.F.radix_bits() - 30 * ctx->nl());   // device radix 2^(30 NL); host radix 2^384 or 2^256
  {
    uint32_t pw[12] = {0};
    for (int i = 0; i < 8; i++) pw[i] = Fp253::PW[i];
    ctx->hte.init(pw, 3021);
    ctx->k_te_to_host = ctx->hte.F.pow2(2 * ctx->hte.F.radix_bits() - 270);
  }
  *out = ctx;
  return MSM_OK;
}

void msm_ctx_destroy(msm_ctx* ctx) {
  if (!ctx) return;
  ctx->fan.clear();
  for (msm_ctx* c : ctx->children) msm_ctx_destroy(c);
  ctx->children.clear();
  (void)hipSetDevice(ctx->device);
  for (size_t i = 0; i < ctx->sets.size(); i++)
    if ((int)i != ctx->cur_set) { ctx->release(ctx->sets[i].rows); ctx->release(ctx->sets[i].tabs); }
  for (void* p : ctx->allocs) (void)hipFree(p);
  ctx->allocs.clear();
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (DevBuf* b : {&ctx->rows, &ctx->tabs, &ctx->scal, &ctx->errflag, &ctx->misc}) ctx->release(*b);
  for (auto& w : ctx->ws) {
    if (w.stream) (void)hipStreamSynchronize(w.stream);
    for (DevBuf* b : w.all) ctx->release(*b);
    if (w.h_info) (void)hipHostFree(w.h_info);
    if (w.h_part) (void)hipHostFree(w.h_part);
    for (auto& e : w.ev) if (e) (void)hipEventDestroy(e);
    if (w.side) { (void)hipStreamSynchronize(w.side); (void)hipStreamDestroy(w.side); }
    if (w.stream) (void)hipStreamDestroy(w.stream);
  }
  if (ctx->h_info) (void)hipHostFree(ctx->h_info);
  if (ctx->stage_pin) (void)hipHostFree(ctx->stage_pin);
  for (auto& st : ctx->stage_stream) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  for (auto& row : ctx->stage_ev) for (auto& e : row) if (e) (void)hipEventDestroy(e);
  for (auto& row : ctx->piece_ev) for (auto& e : row) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->ev_dig) if (e) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* msm_last_error(const msm_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

static int set_points_one(msm_ctx* ctx, const void* points, uint64_t n, int on_device, int check_curve) {
  if (!ctx || (!points && n)) return fail(ctx, MSM_ERR_ARG, "msm_set_points: null argument");
  if (n >= (1ull << 30)) return fail(ctx, MSM_ERR_ARG, "msm_set_points: n must be < 2^30");
  const bool te = ctx->is_te();
  const size_t wire_bytes = 2 * ctx->coord_bytes();   // x || y, little-endian
  const size_t row_words = te ? te::TE_ROW_WORDS : ROW_WORDS;
  try {
    HIPCHK(hipSetDevice(ctx->device));
    ctx->n_points = 0;
    ctx->drop_tables();   // window tables belong to the points they were built from
    ctx->ensure(ctx->rows, std::max<uint64_t>(n, 1) * row_words * 4);
    const uint32_t* d_wire = (const uint32_t*)points;
    if (!on_device && n) {
      ctx->ensure(ctx->misc, n * wire_bytes);
      upload_staged(ctx, ctx->misc.p, points, n * wire_bytes);
      d_wire = (const uint32_t*)ctx->misc.p;
    }
    HIPCHK(hipMemsetAsync(ctx->errflag.p, 0, 4, ctx->stream));
    if (n) {
      uint64_t grid = (n + 255) / 256;
      if (te)
        hipLaunchKernelGGL(te::k_te_points_from_wire, dim3((uint32_t)grid), dim3(256), 0, ctx->stream, (uint32_t*)ctx->rows.p, d_wire,
                           n, check_curve, (uint32_t*)ctx->errflag.p);
      else
        W_LAUNCH(ctx, k_points_from_wire, dim3((uint32_t)grid), dim3(256), 0, ctx->stream, (uint32_t*)ctx->rows.p, d_wire, n,
                           check_curve, (uint32_t*)ctx->errflag.p);
    }
    HIPCHK(hipMemcpyAsync(ctx->h_info, ctx->errflag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    if (!on_device) ctx->release(ctx->misc);
    if (ctx->h_info[0] & 1) return fail(ctx, MSM_ERR_POINT, "msm_set_points: coordinate >= p");
    if (ctx->h_info[0] & 2) return fail(ctx, MSM_ERR_POINT, "msm_set_points: point not on curve");
    ctx->n_points = n;
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

int msm_set_points(msm_ctx* ctx, const void* points, uint64_t n, int on_device, int check_curve) {
  if (!ctx || ctx->children.empty()) return set_points_one(ctx, points, n, on_device, check_curve);
  // multi-device context: every device keeps the whole point set (an MSM then runs over a share of the points per device by
  // default, or over a share of the windows with msm_opts.by_window: either way without moving points)
  try {
    std::vector<uint8_t> host;
    const void* src = points;
    if (on_device && n) {   // the buffer lives on devices[0]: the other devices take it through the host
      host.resize((size_t)n * 2 * ctx->coord_bytes());
      HIPCHK(hipSetDevice(ctx->device));
      HIPCHK(hipMemcpy(host.data(), points, host.size(), hipMemcpyDeviceToHost));
      src = host.data();
    }
    return on_all_devices(ctx, [&](msm_ctx* c) {
      return (c == ctx) ? set_points_one(c, points, n, on_device, check_curve) : set_points_one(c, src, n, 0, check_curve);
    });
  } MSM_CATCH_ALL(ctx)
}

int msm_plan(const msm_ctx* ctx, uint64_t n, const msm_opts* opts, int32_t* c_out, int32_t* K_out) {
  Plan pl;   // plain arithmetic: nothing here can throw
  bool tables_wanted = false;
  // (a context with resident points answers for msm_run over them, window tables included; msm_window_sums and shards of
  // the points always run the plain plan, which is also what a context without points reports)
  // (msm_opts.merged_sums: the plan of msm_window_sums over a range of the points that may run on range tables)
  const bool run_like = ctx && n && ((n == ctx->n_points && !(opts && opts->point_lo)) || (opts && opts->merged_sums));
  int rc = run_like ? make_run_plan(const_cast<msm_ctx*>(ctx), n, opts, false, pl, tables_wanted) : make_plan(ctx, n, opts, pl);
  if (rc) return rc;
  if (c_out) *c_out = pl.c;
  if (K_out) *K_out = pl.K;
  return MSM_OK;
}

int msm_window_sums(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const msm_opts* opts, uint8_t* partials_out,
                    msm_result* stats) {
  if (!ctx || !partials_out || (!scalars && n)) return fail(ctx, MSM_ERR_ARG, "msm_window_sums: null argument");
  if ((opts ? opts->point_lo : 0) + n > ctx->n_points)
    return fail(ctx, MSM_ERR_NO_POINTS, "msm_window_sums: points [%llu, +%llu) but %llu resident points",
                (unsigned long long)(opts ? opts->point_lo : 0), (unsigned long long)n, (unsigned long long)ctx->n_points);
  Plan pl;
  // msm_opts.merged_sums: the caller only combines the sums (msm_combine / msm_combine_groups), so the call may hand them back
  // merged -- and with that run on window tables, those of the whole set or of the range of the points it covers
  const bool merged = opts && opts->merged_sums;
  bool tables_wanted = false;
  if (merged ? make_run_plan(ctx, n, opts, false, pl, tables_wanted, /*note_range=*/true) : make_plan(ctx, n, opts, pl))
    return fail(ctx, MSM_ERR_ARG, "msm_window_sums: bad window size");
  pl.merged = merged;
  int k_lo = opts ? opts->k_lo : 0, k_hi = opts ? opts->k_hi : 0;
  if (k_lo == 0 && k_hi == 0) k_hi = pl.K;
  if (k_lo < 0 || k_hi > pl.K || k_lo >= k_hi) return fail(ctx, MSM_ERR_ARG, "msm_window_sums: bad window shard [%d, %d) of %d", k_lo, k_hi, pl.K);
  if (stats) memset(stats, 0, sizeof(*stats));
  try {
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<uint32_t> words;
    pl.tables = n && tables_wanted && use_window_tables(ctx, n, opts, pl, /*may_build=*/true);
    if (ctx->is_te()) {
      // extended point (X : Y : Z : T) sent as X || Y || Z; the receiver rebuilds T (msm_combine: T Z = X Y)
      if (n) any_window_sums(ctx, scalars, n, on_device, opts, k_lo, k_hi, pl, words, stats);
      const auto& C = ctx->hte;
      msm_host::Fe6 one = {{1, 0, 0, 0, 0, 0}}, t;
      for (int k = 0; k < k_hi - k_lo; k++) {
        const msm_host::Ext6 P = n ? te_partial_to_host(ctx, &words[(size_t)k * 32]) : C.zero();
        C.F.mul(t, P.X, one); fe6_to_bytes(partials_out + (size_t)k * 144, t);
        C.F.mul(t, P.Y, one); fe6_to_bytes(partials_out + (size_t)k * 144 + 48, t);
        C.F.mul(t, P.Z, one); fe6_to_bytes(partials_out + (size_t)k * 144 + 96, t);
      }
      if (stats) { stats->c = pl.c; stats->K = pl.K; }
      return MSM_OK;
    }
    if (n == 0) {
      words.assign((size_t)(k_hi - k_lo) * 36, 0);
    } else {
      any_window_sums(ctx, scalars, n, on_device, opts, k_lo, k_hi, pl, words, stats);
    }
    // to 48-byte canonical integers (leave device Montgomery form on the host)
    for (int k = 0; k < k_hi - k_lo; k++) {
      const uint32_t* w = &words[(size_t)k * 36];
      bool zero_z = true;
      for (int j = 0; j < 12; j++) zero_z &= w[24 + j] == 0;
      msm_host::Proj6 P;
      if (n == 0 || zero_z) P = ctx->hc.zero();
      else P = partial_to_host(ctx, w);
      msm_host::Fe6 one = {{1, 0, 0, 0, 0, 0}}, t;
      ctx->hc.F.mul(t, P.X, one); fe6_to_bytes(partials_out + (size_t)k * 144, t);
      ctx->hc.F.mul(t, P.Y, one); fe6_to_bytes(partials_out + (size_t)k * 144 + 48, t);
      ctx->hc.F.mul(t, P.Z, one); fe6_to_bytes(partials_out + (size_t)k * 144 + 96, t);
    }
    if (stats) { stats->c = pl.c; stats->K = pl.K; }
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

int msm_combine(msm_ctx* ctx, const uint8_t* partials, int32_t K, int32_t c, msm_result* out) {
  // pure host arithmetic: ctx may be NULL (then BLS12-377 G1; msm_combine_curve names the curve without a context)
  if (!partials || !out || K <= 0 || c <= 0) return fail(ctx, MSM_ERR_ARG, "msm_combine: bad argument");
  try {
    if (ctx && ctx->is_te()) {
      int rc = te_combine_impl(partials, K, c, out);
      return rc ? fail(ctx, rc, "msm_combine: coordinate >= p") : MSM_OK;
    }
    return combine_impl(ctx, ctx ? ctx->hc : *static_host_curve(MSM_CURVE_BLS12_377_G1), partials, K, c, out, 1);
  } MSM_CATCH_ALL(ctx)
}

int msm_combine_groups(int curve, const uint8_t* partials, int32_t G, int32_t K, int32_t c, msm_result* out) {
  if (!partials || !out || K <= 0 || c <= 0 || G <= 0) return MSM_ERR_ARG;
  msm_ctx* const no_ctx = nullptr;
  try {
    if (curve == MSM_CURVE_ED_ON_BLS12_377) return te_combine_impl(partials, K, c, out, G);
    const msm_host::Curve6* C = static_host_curve(curve);
    if (!C) return MSM_ERR_ARG;
    return combine_impl(nullptr, *C, partials, K, c, out, G);
  } MSM_CATCH_ALL(no_ctx)
}

int msm_combine_curve(int curve, const uint8_t* partials, int32_t K, int32_t c, msm_result* out) {
  return msm_combine_groups(curve, partials, 1, K, c, out);
}

// msm_run over HOST scalars of a big input: the scalars cross PCIe behind the computation (PieceUpload, msm_upload.hip).  The
// sort of a window needs every digit of its range, so the unit of overlap is a range of the points: the call runs as whole
// MSMs over 1/16, 3/16 and the rest of the points (1/8, 3/8, rest below 2^25) as their scalars arrive -- each with the
// window the library picks for ITS size (round 4 ran all ranges under one window picked for an eighth of the input: the last,
// biggest range then missed the 21-bit plan) -- and adds the three results.  The link moves scalars ~4x as fast as the GPU
// consumes them, so every range may be ~4x its predecessor and still be there in time.
// (the reference's counterpart: scalarsFromBytes into shared wasm memory before the call, src/parallel.ts:119-133)
static int run_piped(msm_ctx* ctx, const void* scalars, uint64_t n, const msm_opts* opts, msm_result* out, const char* who) {
  memset(out, 0, sizeof(*out));
  try {
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipEventRecord(ctx->ev[8], ctx->stream));
    const uint64_t gran = msm_ctx::STAGE_CHUNK / 32;   // scalars per staging chunk
    const int big = n >= (1ull << 25);
    std::vector<uint64_t> piece_end;
    std::vector<int> shifts = {big ? 4 : 3, big ? 2 : 1};
    for (int sh : shifts) piece_end.push_back(((n >> sh) / gran) * gran);
    piece_end.push_back(n);
    ctx->ensure(ctx->scal, n * 32);   // before any workspace is sized from what the device has free
    std::vector<size_t> ends;
    for (uint64_t e : piece_end) ends.push_back((size_t)e * 32);
    PieceUpload pipe(ctx, ctx->scal.p, scalars, n * 32, ends);
    const uint32_t base_lo = opts ? opts->point_lo : 0;
    const size_t Q = piece_end.size();
    struct Range {
      uint64_t lo = 0, cnt = 0;
      Plan pq;
      msm_opts o;
      msm_result st;
      std::vector<uint32_t> words;
    };
    std::vector<Range> R(Q);
    for (size_t q = 0; q < Q; q++) {
      R[q].lo = q ? piece_end[q - 1] : 0;
      R[q].cnt = piece_end[q] - R[q].lo;
      if (opts) R[q].o = *opts; else memset(&R[q].o, 0, sizeof(msm_opts));
      R[q].o.point_lo = base_lo + (uint32_t)R[q].lo;
      memset(&R[q].st, 0, sizeof(msm_result));
      if (R[q].cnt && make_plan(ctx, R[q].cnt, &R[q].o, R[q].pq)) return fail(ctx, MSM_ERR_ARG, "%s: bad window size", who);
      R[q].pq.merged = true;
    }
    // the ranges in order, each as soon as its scalars are in HBM (a second pipeline on the device running the middle range beside
    // its neighbours was tried and bought nothing -- work is conserved: profiles/r05_experiments.txt item 11)
    for (size_t q = 0; q < Q; q++) {
      Range& r = R[q];
      if (r.cnt == 0) continue;
      pipe.wait_piece((int)q, ctx->stream);
      HIPCHK(hipStreamSynchronize(ctx->stream));       // the range's scalars are in HBM
      window_sums_impl(ctx, (const uint32_t*)ctx->scal.p + r.lo * 8, r.cnt, 1, &r.o, 0, r.pq.K, r.pq, r.words, &r.st, r.o.point_lo);
    }
    msm_host::Proj6 acc = ctx->hc.zero();
    for (size_t q = 0; q < Q; q++) {
      const Range& r = R[q];
      if (r.cnt == 0) continue;
      std::vector<msm_host::Proj6> P(r.pq.K);
      for (int k = 0; k < r.pq.K; k++) P[k] = partial_to_host(ctx, &r.words[(size_t)k * 36]);
      acc = ctx->hc.add(acc, horner_points(ctx->hc, P, r.pq.c));
      for (int j = 0; j < MSM_N_PHASES; j++) out->phase_ms[j] += r.st.phase_ms[j];
      out->n_pairs += r.st.n_pairs;
      out->n_pairs_algo += r.st.n_pairs_algo;
      out->rounds += r.st.rounds;
      out->max_bucket = std::max(out->max_bucket, r.st.max_bucket);
      out->c = r.pq.c;   // the plan of the last, biggest range
      out->K = r.pq.K;
    }
    out->phase_ms[MSM_T_UPLOAD] = pipe.finish();   // wall time of the background transfer
    proj_to_result(ctx->hc, acc, out);
    HIPCHK(hipEventRecord(ctx->ev[11], ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[8], ctx->ev[11]));
    out->phase_ms[MSM_T_TOTAL] = ms;
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

static int run_impl(msm_ctx* ctx, const void* scalars, const void* const* placed, uint64_t n, int on_device, const msm_opts* opts,
                    msm_result* out, const char* who) {
  if ((opts ? opts->point_lo : 0) + n > ctx->n_points)
    return fail(ctx, MSM_ERR_NO_POINTS, "%s: points [%llu, +%llu) but %llu resident points", who,
                (unsigned long long)(opts ? opts->point_lo : 0), (unsigned long long)n, (unsigned long long)ctx->n_points);
  if (!placed && !on_device && n >= (1ull << 24) && ctx->children.empty() && !ctx->is_te() && !(opts && opts->c))
    return run_piped(ctx, scalars, n, opts, out, who);
  Plan pl;
  bool tables_wanted = false;   // window tables (msm_tables.hip): the plan is then the one tables want
  if (make_run_plan(ctx, n, opts, placed != nullptr, pl, tables_wanted, /*note_range=*/true)) return fail(ctx, MSM_ERR_ARG, "%s: bad window size", who);
  pl.merged = true;
  memset(out, 0, sizeof(*out));
  out->c = pl.c;
  out->K = pl.K;
  if (n == 0) {
    if (ctx->is_te()) out->y[0] = 1;   // identity (0, 1)
    else out->is_infinity = 1;
    return MSM_OK;
  }
  try {
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<uint32_t> words;
    // window tables (msm_tables.hip): built here on the first default-plan call over the whole point set
    pl.tables = tables_wanted && use_window_tables(ctx, n, opts, pl, /*may_build=*/true);
    any_window_sums(ctx, scalars, n, on_device, opts, 0, pl.K, pl, words, out, placed);
    HIPCHK(hipEventRecord(ctx->ev[10], ctx->stream));
    // (a run on tables leaves the whole sum, weights included, in slot 0 and identities in the others: the Horner step over all
    // K slots returns it unchanged, and is what a call that had to leave the tables -- ranges of the points -- needs)
    const int slots = pl.K;
    if (ctx->is_te()) {
      te_horner_to_affine(ctx, words, slots, pl.c, out);
    } else {
      std::vector<msm_host::Proj6> P(slots);
      for (int k = 0; k < slots; k++) P[k] = partial_to_host(ctx, &words[(size_t)k * 36]);
      horner_to_affine(ctx->hc, P, pl.c, out);
    }
    HIPCHK(hipEventRecord(ctx->ev[11], ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[10], ctx->ev[11]));
    out->phase_ms[MSM_T_FINAL] = ms;
    out->phase_ms[MSM_T_TOTAL] += ms;
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

int msm_run(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const msm_opts* opts, msm_result* out) {
  if (!ctx || !out || (!scalars && n)) return fail(ctx, MSM_ERR_ARG, "msm_run: null argument");
  return run_impl(ctx, scalars, nullptr, n, on_device, opts, out, "msm_run");
}

int msm_run_placed(msm_ctx* ctx, const void* const* dev_scalars, uint64_t n, const msm_opts* opts, msm_result* out) {
  if (!ctx || !out || !dev_scalars) return fail(ctx, MSM_ERR_ARG, "msm_run_placed: null argument");
  if (opts && opts->by_window && !ctx->children.empty())
    return fail(ctx, MSM_ERR_ARG, "msm_run_placed: placed scalars are the shares of a points split (by_window must be 0)");
  const int ndev = 1 + (int)ctx->children.size();
  for (int d = 0; d < ndev; d++)
    if (!dev_scalars[d] && n * (uint64_t)(d + 1) / ndev > n * (uint64_t)d / ndev)
      return fail(ctx, MSM_ERR_ARG, "msm_run_placed: no scalars for device %d", d);
  return run_impl(ctx, nullptr, dev_scalars, n, 1, opts, out, "msm_run_placed");
}

int msm_reserve(msm_ctx* ctx, uint64_t n, const msm_opts* opts) {
  if (!ctx) return MSM_ERR_ARG;
  if ((opts ? opts->point_lo : 0) + n > ctx->n_points)
    return fail(ctx, MSM_ERR_NO_POINTS, "msm_reserve: %llu points asked for, %llu resident", (unsigned long long)n, (unsigned long long)ctx->n_points);
  if (n == 0) return MSM_OK;
  void* dev = nullptr;
  try {
    // one MSM over generated scalars: every buffer the real call will need exists afterwards, and so do the window tables
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMalloc(&dev, n * 32));
    int rc = msm_generate_scalars(ctx, n, 0x5eed, dev, nullptr);
    msm_result r;
    if (rc == MSM_OK) rc = msm_run(ctx, dev, n, 1, opts, &r);
    (void)hipFree(dev);
    return rc;
  } catch (...) {
    if (dev) (void)hipFree(dev);
    return fail(ctx, MSM_ERR_HIP, "msm_reserve: device allocation failed");
  }
}

int msm_get_points(msm_ctx* ctx, uint64_t first, uint64_t count, uint8_t* out_xy) {
  if (!ctx || !out_xy || first + count > ctx->n_points) return fail(ctx, MSM_ERR_ARG, "msm_get_points: bad argument");
  if (ctx->is_te()) {
    try {
      HIPCHK(hipSetDevice(ctx->device));
      std::vector<uint32_t> rows((size_t)count * te::TE_ROW_WORDS);
      if (count)
        HIPCHK(hipMemcpy(rows.data(), (const uint32_t*)ctx->rows.p + first * te::TE_ROW_WORDS, rows.size() * 4, hipMemcpyDeviceToHost));
      msm_host::Fe6 one = {{1, 0, 0, 0, 0, 0}};
      for (uint64_t i = 0; i < count; i++)
        for (int j = 0; j < 2; j++) {
          msm_host::Fe6 t = {{0, 0, 0, 0, 0, 0}};
          const uint32_t* w = &rows[(size_t)i * te::TE_ROW_WORDS + 8 * j];
          for (int q = 0; q < 4; q++) t.v[q] = (uint64_t)w[2 * q] | ((uint64_t)w[2 * q + 1] << 32);
          ctx->hte.F.mul(t, t, ctx->k_te_to_host);
          ctx->hte.F.mul(t, t, one);
          for (int q = 0; q < 4; q++)
            for (int b = 0; b < 8; b++) out_xy[i * 64 + 32 * j + 8 * q + b] = (uint8_t)(t.v[q] >> (8 * b));
        }
    } MSM_CATCH_ALL(ctx)
    return MSM_OK;
  }
  try {
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<uint32_t> rows((size_t)count * ROW_WORDS);
    if (count)
      HIPCHK(hipMemcpy(rows.data(), (const uint32_t*)ctx->rows.p + first * ROW_WORDS, rows.size() * 4, hipMemcpyDeviceToHost));
    const int nw = ctx->nw();
    const size_t cb = ctx->coord_bytes();
    memset(out_xy, 0, (size_t)count * 2 * cb);
    msm_host::Fe6 one = {{1, 0, 0, 0, 0, 0}};
    for (uint64_t i = 0; i < count; i++) {
      const uint32_t* row = &rows[(size_t)i * ROW_WORDS];
      if (row[nw - 1] == INF_WORD) continue;
      for (int j = 0; j < 2; j++) {
        msm_host::Fe6 t;
        words_to_fe6(t, row + nw * j, nw);
        ctx->hc.F.mul(t, t, ctx->k_dev_to_host);  // host Montgomery
        ctx->hc.F.mul(t, t, one);                 // plain
        uint8_t b48[48];
        fe6_to_bytes(b48, t);
        memcpy(out_xy + i * 2 * cb + cb * j, b48, cb);
      }
    }
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

// ---- point-set handles: several resident point sets per context, one of them current ------------------------

static int pointset_select_one(msm_ctx* ctx, int32_t id) {
  if (id < 0 || id >= (int)ctx->sets.size() || (id != 0 && !ctx->sets[id].live))
    return fail(ctx, MSM_ERR_ARG, "msm_pointset_select: no point set %d", (int)id);
  if (id == ctx->cur_set) return MSM_OK;
  ctx->sets[ctx->cur_set].rows = ctx->rows;
  ctx->sets[ctx->cur_set].n = ctx->n_points;
  ctx->sets[ctx->cur_set].tab_c = ctx->tab_c;
  ctx->sets[ctx->cur_set].tab_K = ctx->tab_K;
  ctx->sets[ctx->cur_set].tab_lo = ctx->tab_lo;
  ctx->sets[ctx->cur_set].tab_n = ctx->tab_n;
  ctx->sets[ctx->cur_set].tabs = ctx->tabs;
  ctx->rows = ctx->sets[id].rows;
  ctx->n_points = ctx->sets[id].n;
  ctx->tab_c = ctx->sets[id].tab_c;
  ctx->tab_K = ctx->sets[id].tab_K;
  ctx->tab_lo = ctx->sets[id].tab_lo;
  ctx->tab_n = ctx->sets[id].tab_n;
  ctx->tabs = ctx->sets[id].tabs;
  ctx->cand_n = 0;
  ctx->cur_set = id;
  return MSM_OK;
}

int msm_pointset_create(msm_ctx* ctx, int32_t* id_out) {
  if (!ctx || !id_out) return fail(ctx, MSM_ERR_ARG, "msm_pointset_create: null argument");
  try {
    return on_all_devices(ctx, [&](msm_ctx* c) {
      int id = -1;
      for (size_t i = 1; i < c->sets.size(); i++)
        if (!c->sets[i].live) { id = (int)i; break; }
      if (id < 0) { c->sets.emplace_back(); id = (int)c->sets.size() - 1; }   // children stay in lockstep: same ids
      c->sets[id] = msm_ctx::PointSet();
      c->sets[id].live = true;
      if (c == ctx) *id_out = id;
      return pointset_select_one(c, id);
    });
  } MSM_CATCH_ALL(ctx)
}

int msm_pointset_select(msm_ctx* ctx, int32_t id) {
  if (!ctx) return MSM_ERR_ARG;
  try {
    return on_all_devices(ctx, [&](msm_ctx* c) { return pointset_select_one(c, id); });
  } MSM_CATCH_ALL(ctx)
}

int msm_pointset_destroy(msm_ctx* ctx, int32_t id) {
  if (!ctx) return MSM_ERR_ARG;
  if (id <= 0 || id >= (int)ctx->sets.size() || !ctx->sets[id].live)
    return fail(ctx, MSM_ERR_ARG, "msm_pointset_destroy: no such point set %d (the default set 0 stays)", (int)id);
  try {
    return on_all_devices(ctx, [&](msm_ctx* c) {
      if (c->cur_set == id) pointset_select_one(c, 0);
      (void)hipSetDevice(c->device);
      c->release(c->sets[id].rows);
      c->release(c->sets[id].tabs);
      c->sets[id] = msm_ctx::PointSet();
      return (int)MSM_OK;
    });
  } MSM_CATCH_ALL(ctx)
}

// ---- device buffers for scalar handles ------------------------------------------------------------------------

int msm_device_alloc(msm_ctx* ctx, uint64_t bytes, void** dev_ptr_out) {
  if (!ctx || !dev_ptr_out) return fail(ctx, MSM_ERR_ARG, "msm_device_alloc: null argument");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, std::max<uint64_t>(bytes, 32)));
    ctx->allocs.push_back(p);
    *dev_ptr_out = p;
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_device_free(msm_ctx* ctx, void* dev_ptr) {
  if (!ctx) return MSM_ERR_ARG;
  auto it = std::find(ctx->allocs.begin(), ctx->allocs.end(), dev_ptr);
  if (it == ctx->allocs.end()) return fail(ctx, MSM_ERR_ARG, "msm_device_free: not a buffer of this context");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    ctx->allocs.erase(it);
    HIPCHK(hipFree(dev_ptr));
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_device_upload(msm_ctx* ctx, void* dev_ptr, const void* host, uint64_t bytes) {
  if (!ctx || !dev_ptr || (!host && bytes)) return fail(ctx, MSM_ERR_ARG, "msm_device_upload: null argument");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    if (bytes) {
      upload_staged(ctx, dev_ptr, host, bytes);
      HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_set_workspace_limit(msm_ctx* ctx, uint64_t bytes) {
  if (!ctx) return MSM_ERR_ARG;
  ctx->ws_limit = bytes;
  if (!bytes) {   // back to automatic: small calls use the creation-time rule again
    try {
      HIPCHK(hipSetDevice(ctx->device));
      size_t free_b = 0, total_b = 0;
      HIPCHK(hipMemGetInfo(&free_b, &total_b));
      uint64_t held = 0;
      for (auto& w : ctx->ws)
        for (DevBuf* b : w.all) held += b->cap;
      ctx->ws_budget = (uint64_t)((free_b + held) * 0.55L);
    } MSM_CATCH_ALL(ctx)
  }
  for (msm_ctx* c : ctx->children) {
    int rc = msm_set_workspace_limit(c, bytes);
    if (rc != MSM_OK) return rc;
  }
  return MSM_OK;
}

// ---- multi-device context ---------------------------------------------------------------------------------------

int msm_ctx_create_multi(msm_ctx** out, int curve, const int32_t* devices, int32_t n_devices) {
  if (!out || !devices || n_devices < 1) return MSM_ERR_ARG;
  *out = nullptr;
  msm_ctx* ctx = nullptr;
  int rc = msm_ctx_create(&ctx, curve, devices[0]);
  if (rc != MSM_OK) return rc;
  try {
    for (int i = 1; i < n_devices; i++) {
      msm_ctx* c = nullptr;
      rc = msm_ctx_create(&c, curve, devices[i]);
      if (rc != MSM_OK) {
        msm_ctx_destroy(ctx);
        return rc;
      }
      ctx->children.push_back(c);
      ctx->fan.emplace_back(new HelperThread());
    }
  } catch (...) {
    msm_ctx_destroy(ctx);
    return MSM_ERR_INTERNAL;
  }
  *out = ctx;
  return MSM_OK;
}

int msm_ctx_device_count(const msm_ctx* ctx) { return ctx ? 1 + (int)ctx->children.size() : 0; }

}  // extern "C"
