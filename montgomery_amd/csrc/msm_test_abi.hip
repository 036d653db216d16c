// Operator-level test entries of the C ABI (msm_test_*: the fine-grained operator table of src/field-msm.ts:86-123 as a GPU
// debug surface) and the synthetic input generators.
#include "msm_internal.h"

using namespace msm;
using namespace msmi;

extern "C" {

int msm_test_fp(msm_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, uint64_t n) {
  if (!ctx || !a || !b || !out) return fail(ctx, MSM_ERR_ARG, "msm_test_fp: null argument");
  const size_t nb = ctx->coord_bytes();
  try {
    HIPCHK(hipSetDevice(ctx->device));
    ctx->ensure(ctx->misc, n * nb * 3 + 64);
    uint8_t* d = (uint8_t*)ctx->misc.p;
    HIPCHK(hipMemcpyAsync(d, a, n * nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d + n * nb, b, n * nb, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->is_te())
      hipLaunchKernelGGL(te::k_te_test_fp, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)(d + 2 * n * nb),
                         (const uint32_t*)d, (const uint32_t*)(d + n * nb), (uint32_t)n, op);
    else
      W_LAUNCH(ctx, k_test_fp, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)(d + 2 * n * nb),
                         (const uint32_t*)d, (const uint32_t*)(d + n * nb), (uint32_t)n, op);
    HIPCHK(hipMemcpyAsync(out, d + 2 * n * nb, n * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

int msm_test_batch_inverse(msm_ctx* ctx, const uint8_t* xs, uint8_t* out, uint64_t n, uint32_t per_lane) {
  if (!ctx || !xs || !out || per_lane == 0) return fail(ctx, MSM_ERR_ARG, "msm_test_batch_inverse: bad argument");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nb = ctx->coord_bytes();
    ctx->ensure(ctx->misc, n * 2 * nb + 64);
    uint8_t* d = (uint8_t*)ctx->misc.p;
    HIPCHK(hipMemcpyAsync(d, xs, n * nb, hipMemcpyHostToDevice, ctx->stream));
    uint64_t lanes = (n + per_lane - 1) / per_lane;
    if (ctx->is_te())
      hipLaunchKernelGGL((k_test_batch_inverse<te::CvEdField>), dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, ctx->stream,
                         (uint32_t*)(d + n * nb), (const uint32_t*)d, (uint32_t)n, per_lane);
    else
      W_LAUNCH(ctx, k_test_batch_inverse, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, ctx->stream,
                         (uint32_t*)(d + n * nb), (const uint32_t*)d, (uint32_t)n, per_lane);
    HIPCHK(hipMemcpyAsync(out, d + n * nb, n * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

int msm_test_glv(msm_ctx* ctx, const uint8_t* scalars, uint8_t* out, uint64_t n) {
  if (!ctx || !scalars || !out) return fail(ctx, MSM_ERR_ARG, "msm_test_glv: null argument");
  if (ctx->is_te()) return fail(ctx, MSM_ERR_ARG, "msm_test_glv: the twisted Edwards path has no GLV step");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    ctx->ensure(ctx->misc, n * 72 + 64);
    uint8_t* d = (uint8_t*)ctx->misc.p;
    HIPCHK(hipMemcpyAsync(d, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    W_LAUNCH(ctx, k_test_glv, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)(d + n * 32),
                       (const uint32_t*)d, (uint32_t)n);
    HIPCHK(hipMemcpyAsync(out, d + n * 32, n * 40, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

int msm_test_batch_add(msm_ctx* ctx, const uint8_t* g, const uint8_t* h, uint8_t* out, uint64_t n) {
  if (!ctx || !g || !h || !out || n == 0) return fail(ctx, MSM_ERR_ARG, "msm_test_batch_add: bad argument");
  if (ctx->is_te()) {
    // unified extended addition of the gather round (te_add_rows, src/curve-twisted-edwards.ts:84-165): n pairs of
    // 64-byte affine points in, n affine sums out
    try {
      HIPCHK(hipSetDevice(ctx->device));
      DevBuf rows, wire, slots, outb;
      ctx->ensure(wire, 2 * n * 64);
      ctx->ensure(rows, 2 * n * te::TE_ROW_WORDS * 4);
      ctx->ensure(slots, 2 * n * 4);
      ctx->ensure(outb, n * 128);
      std::vector<uint8_t> inter(2 * n * 64);
      std::vector<uint32_t> sl(2 * n);
      for (uint64_t i = 0; i < n; i++) {
        memcpy(&inter[(2 * i) * 64], g + i * 64, 64);
        memcpy(&inter[(2 * i + 1) * 64], h + i * 64, 64);
        sl[2 * i] = (uint32_t)((2 * i) << 1);
        sl[2 * i + 1] = (uint32_t)((2 * i + 1) << 1);
      }
      HIPCHK(hipMemcpyAsync(wire.p, inter.data(), inter.size(), hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(slots.p, sl.data(), sl.size() * 4, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipMemsetAsync(ctx->errflag.p, 0, 4, ctx->stream));
      hipLaunchKernelGGL(te::k_te_points_from_wire, dim3((uint32_t)((2 * n + 255) / 256)), dim3(256), 0, ctx->stream,
                         (uint32_t*)rows.p, (const uint32_t*)wire.p, 2 * n, 0, (uint32_t*)ctx->errflag.p);
      BatchArgs a{};
      a.points = (const uint32_t*)rows.p;
      a.slots = (const uint32_t*)slots.p;
      a.out = (uint4*)outb.p;
      a.out_cap = n;
      a.n_out = n;
      a.steps = 1;
      hipLaunchKernelGGL(te::k_te_add<MODE_GATHER>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, a);
      std::vector<uint32_t> planes(n * 32);
      HIPCHK(hipMemcpyAsync(planes.data(), outb.p, n * 128, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(hipGetLastError());
      const auto& C = ctx->hte;
      msm_host::Fe6 one = {{1, 0, 0, 0, 0, 0}};
      for (uint64_t e = 0; e < n; e++) {
        msm_host::Fe6 co[3];   // X, Y, Z
        for (int j = 0; j < 3; j++) {
          msm_host::Fe6 t = {{0, 0, 0, 0, 0, 0}};
          for (int pl = 0; pl < 2; pl++)
            for (int q = 0; q < 2; q++) {
              const uint32_t* w = &planes[((uint64_t)(2 * j + pl) * n + e) * 4 + 2 * q];
              t.v[2 * pl + q] = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
            }
          if (msm_host::Field6::ge(t, C.F.p)) C.F.sub_raw(t, t, C.F.p);
          C.F.mul(co[j], t, ctx->k_te_to_host);
        }
        msm_host::Fe6 zi, x, y;
        C.F.inv(zi, co[2]);
        C.F.mul(x, co[0], zi);
        C.F.mul(y, co[1], zi);
        C.F.mul(x, x, one);
        C.F.mul(y, y, one);
        uint8_t xb[48], yb[48];
        fe6_to_bytes(xb, x);
        fe6_to_bytes(yb, y);
        memcpy(out + e * 64, xb, 32);
        memcpy(out + e * 64 + 32, yb, 32);
      }
      for (DevBuf* b : {&rows, &wire, &slots, &outb}) ctx->release(*b);
    } MSM_CATCH_ALL(ctx)
    return MSM_OK;
  }
  try {
    HIPCHK(hipSetDevice(ctx->device));
    // rows for 2n points: pair e = (row 2e, row 2e + 1), gathered through identity payload slots
    DevBuf rows, wire, slots, outb, scr;
    const size_t pb = 2 * ctx->coord_bytes();   // wire point; a tree node has the same size
    ctx->ensure(wire, 2 * n * pb);
    ctx->ensure(rows, 2 * n * ROW_WORDS * 4);
    ctx->ensure(slots, 2 * n * 4);
    ctx->ensure(outb, n * pb);
    std::vector<uint8_t> inter(2 * n * pb);
    std::vector<uint32_t> sl(2 * n);
    for (uint64_t i = 0; i < n; i++) {
      memcpy(&inter[(2 * i) * pb], g + i * pb, pb);
      memcpy(&inter[(2 * i + 1) * pb], h + i * pb, pb);
      sl[2 * i] = (uint32_t)((2 * i) << 2);
      sl[2 * i + 1] = (uint32_t)((2 * i + 1) << 2);
    }
    HIPCHK(hipMemcpyAsync(wire.p, inter.data(), 2 * n * pb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(slots.p, sl.data(), 2 * n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->errflag.p, 0, 4, ctx->stream));
    W_LAUNCH(ctx, k_points_from_wire, dim3((uint32_t)((2 * n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)rows.p,
                       (const uint32_t*)wire.p, 2 * n, 0, (uint32_t*)ctx->errflag.p);
    RoundGeom gm = round_geom(ctx, n);
    gm.steps = (uint32_t)std::min<uint64_t>(n, 3);  // exercise the shared inversion with a few pairs per lane
    uint64_t threads = (n + gm.steps - 1) / gm.steps;
    gm.grid = (uint32_t)((threads + 255) / 256);
    gm.T = (uint64_t)gm.grid * 256;
    ctx->ensure(scr, (size_t)gm.steps * NL * gm.T * 4);
    BatchArgs a{};
    a.points = (const uint32_t*)rows.p;
    a.slots = (const uint32_t*)slots.p;
    a.y_off = 4 * ctx->nw();
    a.out = (uint4*)outb.p;
    a.out_cap = n;
    a.scratch = (uint32_t*)scr.p;
    a.sstride = gm.T;
    a.n_out = n;
    a.steps = gm.steps;
    W_LAUNCH_MODE(ctx, k_batch_add, MODE_GATHER, dim3(gm.grid), dim3(256), 0, ctx->stream, a);
    std::vector<uint32_t> planes(n * pb / 4);
    HIPCHK(hipMemcpyAsync(planes.data(), outb.p, n * pb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    for (uint64_t e = 0; e < n; e++) plane_element_to_wire(ctx, planes.data(), n, e, out + e * pb);
    for (DevBuf* b : {&rows, &wire, &slots, &outb, &scr}) ctx->release(*b);
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

static int generate_points_one(msm_ctx* ctx, uint64_t n, uint64_t seed, uint8_t* a_out) {
  try {
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->is_te()) return msm_gen::generate_points_te(ctx, n, seed, a_out);
    return msm_gen::generate_points(ctx, n, seed, a_out);
  } MSM_CATCH_ALL(ctx)
}

int msm_test_fp_raw(msm_ctx* ctx, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, uint64_t n) {
  if (!ctx || !a || !b || !out) return fail(ctx, MSM_ERR_ARG, "msm_test_fp_raw: null argument");
  if (op != MSM_OP_MUL && op != MSM_OP_SQR) return fail(ctx, MSM_ERR_ARG, "msm_test_fp_raw: op must be MSM_OP_MUL or MSM_OP_SQR");
  const size_t nb = (size_t)ctx->nl() * 4;
  try {
    HIPCHK(hipSetDevice(ctx->device));
    ctx->ensure(ctx->misc, n * nb * 3 + 64);
    uint8_t* d = (uint8_t*)ctx->misc.p;
    HIPCHK(hipMemcpyAsync(d, a, n * nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d + n * nb, b, n * nb, hipMemcpyHostToDevice, ctx->stream));
    const dim3 grid((uint32_t)((n + 255) / 256));
    if (ctx->is_te())
      hipLaunchKernelGGL(te::k_te_test_fp_raw, grid, dim3(256), 0, ctx->stream, (uint32_t*)(d + 2 * n * nb), (const uint32_t*)d,
                         (const uint32_t*)(d + n * nb), (uint32_t)n, op);
    else
      W_LAUNCH(ctx, k_test_fp_raw, grid, dim3(256), 0, ctx->stream, (uint32_t*)(d + 2 * n * nb), (const uint32_t*)d,
                         (const uint32_t*)(d + n * nb), (uint32_t)n, op);
    HIPCHK(hipMemcpyAsync(out, d + 2 * n * nb, n * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_test_curve_op(msm_ctx* ctx, int op, const uint8_t* p, const uint8_t* q, uint8_t* out, uint64_t n) {
  if (!ctx || !p || !q || !out) return fail(ctx, MSM_ERR_ARG, "msm_test_curve_op: null argument");
  if (op < 0 || op > 2) return fail(ctx, MSM_ERR_ARG, "msm_test_curve_op: unknown operator");
  const size_t nb = ctx->is_te() ? 128 : 3 * ctx->coord_bytes();   // extended (X, Y, Z, T) or projective (X, Y, Z)
  try {
    HIPCHK(hipSetDevice(ctx->device));
    ctx->ensure(ctx->misc, n * nb * 3 + 64);
    uint8_t* d = (uint8_t*)ctx->misc.p;
    HIPCHK(hipMemcpyAsync(d, p, n * nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d + n * nb, q, n * nb, hipMemcpyHostToDevice, ctx->stream));
    const dim3 grid((uint32_t)((n + 63) / 64));
    if (ctx->is_te())
      hipLaunchKernelGGL(te::k_te_test_curve_op, grid, dim3(64), 0, ctx->stream, (uint32_t*)(d + 2 * n * nb), (const uint32_t*)d,
                         (const uint32_t*)(d + n * nb), (uint32_t)n, op);
    else
      W_LAUNCH(ctx, k_test_curve_op, grid, dim3(64), 0, ctx->stream, (uint32_t*)(d + 2 * n * nb), (const uint32_t*)d,
                         (const uint32_t*)(d + n * nb), (uint32_t)n, op);
    HIPCHK(hipMemcpyAsync(out, d + 2 * n * nb, n * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_test_bucket_reduce(msm_ctx* ctx, const uint8_t* buckets, int32_t K, uint32_t L, int mode, int c0, uint8_t* partials_out,
                           float* ms_out) {
  if (!ctx || !buckets || !partials_out || K <= 0 || L == 0 || (L & (L - 1)) || (mode != 0 && mode != 1))
    return fail(ctx, MSM_ERR_ARG, "msm_test_bucket_reduce: bad argument");
  if (ctx->is_te()) return fail(ctx, MSM_ERR_ARG, "msm_test_bucket_reduce: Weierstrass curves only");
  int cl = 0;
  while ((1u << cl) < L) cl++;
  if (c0 < 0 || c0 > cl) return fail(ctx, MSM_ERR_ARG, "msm_test_bucket_reduce: c0 must be in [0, log2 L]");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    msm_ctx::Workspace& w = ctx->ws[0];
    hipStream_t s = w.stream;
    const uint64_t nb = (uint64_t)K * L;
    const uint64_t cap = nb + 2 * 257 * 512 + 256;   // plane capacity: idle lanes read (and ignore) past the end
    ScopedDevBuf wire, rows, planes, desc, scr;   // released on every path, a HIPCHK / MsmFail thrown in between included
    const size_t pb = 2 * ctx->coord_bytes();
    const int nw = ctx->nw(), np = nw / 4;
    ctx->ensure(wire, nb * pb);
    ctx->ensure(rows, nb * ROW_WORDS * 4);
    ctx->ensure(planes, cap * pb);
    HIPCHK(hipMemcpyAsync(wire.p, buckets, nb * pb, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(ctx->errflag.p, 0, 4, s));
    HIPCHK(hipMemsetAsync(planes.p, 0, cap * pb, s));
    W_LAUNCH(ctx, k_points_from_wire, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, s, (uint32_t*)rows.p, (const uint32_t*)wire.p,
             nb, 0, (uint32_t*)ctx->errflag.p);
    ROWS_TO_PLANES(ctx, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, s, (uint4*)planes.p, cap, (const uint32_t*)rows.p,
                   (uint32_t)nb);
    std::vector<uint32_t> parts((size_t)K * 36, 0);
    float ms = 0;
    if (mode == 0) {
      // every bucket holds exactly one element of the tree buffer: offsets 0, 1, 2, ...
      std::vector<uint32_t> off(nb + 1);
      for (uint64_t b = 0; b <= nb; b++) off[b] = (uint32_t)b;
      ctx->ensure(desc, (nb + 1) * 4);
      HIPCHK(hipMemcpyAsync(desc.p, off.data(), (nb + 1) * 4, hipMemcpyHostToDevice, s));
      HIPCHK(hipEventRecord(w.ev[3], s));
      reduce_buckets(ctx, w, (const uint4*)planes.p, cap, (const uint32_t*)desc.p, nullptr, L, K, parts.data());
      HIPCHK(hipEventElapsedTime(&ms, w.ev[3], w.ev[4]));
    } else {
      // The rounds of reduceBucketsAffine as (first operand, second operand) element lists; the sum replaces the first.
      // e(k, l) = k L + l - 1 for the 1-based bucket index l of the reference.
      const uint32_t L0 = 1u << c0, D = L / L0;
      std::vector<std::vector<uint32_t>> ga, gb;
      auto e = [&](int k, uint64_t l) { return (uint32_t)((uint64_t)k * L + l - 1); };
      auto round = [&](const std::function<void(int, std::vector<uint32_t>&, std::vector<uint32_t>&)>& fill) {
        std::vector<uint32_t> A, B;
        for (int k = 0; k < K; k++) fill(k, A, B);
        if (!A.empty()) { ga.push_back(std::move(A)); gb.push_back(std::move(B)); }
      };
      // linear part: suffix sums inside every chunk of L0 buckets (:563-588)
      for (uint32_t l = L0 - 1; l >= 1; l--)
        round([&](int k, std::vector<uint32_t>& A, std::vector<uint32_t>& B) {
          for (uint32_t d = 0; d < D; d++) { A.push_back(e(k, (uint64_t)d * L0 + l)); B.push_back(e(k, (uint64_t)d * L0 + l + 1)); }
        });
      // logarithmic part: chunk heads collect the chunks to their right, power-of-two spans (:590-615)
      for (uint64_t L1 = L0, D1 = D >> 1; D1 > 0; L1 <<= 1, D1 >>= 1)
        round([&](int k, std::vector<uint32_t>& A, std::vector<uint32_t>& B) {
          for (uint64_t d = 0; d < D1; d++) { A.push_back(e(k, d * 2 * L1 + 1)); B.push_back(e(k, (d * 2 + 1) * L1 + 1)); }
        });
      // doublings: every head is weighted with the number of buckets it stands for (:616-641)
      if (D > 1)
        for (int j = 0; j < c0; j++)
          round([&](int k, std::vector<uint32_t>& A, std::vector<uint32_t>& B) {
            for (uint32_t d = 1; d < D; d++) { A.push_back(e(k, (uint64_t)d * L0 + 1)); B.push_back(e(k, (uint64_t)d * L0 + 1)); }
          });
      for (uint64_t L1 = 2ull * L0, D1 = D >> 1; D1 > 1; L1 <<= 1, D1 >>= 1)
        round([&](int k, std::vector<uint32_t>& A, std::vector<uint32_t>& B) {
          for (uint64_t d = 1; d < D1; d++) { A.push_back(e(k, d * L1 + 1)); B.push_back(e(k, d * L1 + 1)); }
        });
      // the buckets now fill the triangle: one addition tree over all of them (:643-662)
      for (uint64_t m = 1; m < L; m *= 2)
        round([&](int k, std::vector<uint32_t>& A, std::vector<uint32_t>& B) {
          for (uint64_t l = 1; l < L; l += 2 * m) { A.push_back(e(k, l)); B.push_back(e(k, l + m)); }
        });
      size_t total = 0, biggest = 0;
      for (auto& v : ga) { total += v.size(); biggest = std::max(biggest, v.size()); }
      ctx->ensure(desc, std::max<size_t>(total, 1) * 8);
      std::vector<uint32_t> flat(2 * total);
      {
        size_t o = 0;
        for (size_t r = 0; r < ga.size(); r++) {
          for (size_t i = 0; i < ga[r].size(); i++) { flat[o + i] = (ga[r][i] << 1) | 1u; flat[total + o + i] = gb[r][i]; }
          o += ga[r].size();
        }
      }
      HIPCHK(hipMemcpyAsync(desc.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice, s));
      {
        const RoundGeom g = round_geom(ctx, std::max<uint64_t>(biggest, 1));
        ctx->ensure(scr, (size_t)g.steps * NL * g.T * 4);
      }
      HIPCHK(hipEventRecord(w.ev[3], s));
      size_t o = 0;
      for (size_t r = 0; r < ga.size(); r++) {
        const uint64_t np = ga[r].size();
        const RoundGeom g = round_geom(ctx, np);
        BatchArgs a{};
        a.in = (const uint4*)planes.p;
        a.in_cap = cap;
        a.out = (uint4*)planes.p;
        a.out_cap = cap;
        a.scratch = (uint32_t*)scr.p;
        a.sstride = g.T;
        a.n_out = np;
        a.steps = g.steps;
        a.desc = (const uint32_t*)desc.p + o;
        a.desc_b = (const uint32_t*)desc.p + total + o;
        a.inplace = 1;
        W_LAUNCH_MODE(ctx, k_batch_add, MODE_SEARCH, dim3(g.grid), dim3(256), 0, s, a);
        o += np;
      }
      HIPCHK(hipEventRecord(w.ev[4], s));
      // element e(k, 1) -> partial (X, Y, Z = 1 in device Montgomery form; all-zero = the identity)
      std::vector<uint32_t> el((size_t)K * 24);
      for (int k = 0; k < K; k++)
        for (int cpl = 0; cpl < 2 * np; cpl++)
          HIPCHK(hipMemcpyAsync(&el[(size_t)k * 24 + 4 * cpl], (const uint4*)planes.p + (uint64_t)cpl * cap + (uint64_t)k * L, 16,
                                hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      HIPCHK(hipGetLastError());
      HIPCHK(hipEventElapsedTime(&ms, w.ev[3], w.ev[4]));
      for (int k = 0; k < K; k++) {
        const uint32_t* q = &el[(size_t)k * 24];
        if (q[nw - 1] == INF_WORD) continue;   // identity: the partial stays all-zero
        memcpy(&parts[(size_t)k * 36], q, nw * 4);            // X, Y at words 0 and 12 of the partial (upper words zero)
        memcpy(&parts[(size_t)k * 36 + 12], q + nw, nw * 4);
        const msm_host::Fe6 one_dev = ctx->hc.F.pow2(30 * ctx->nl());   // Z = 1 in the form x and y are in: device Montgomery
        for (int q2 = 0; q2 < 6; q2++) {
          parts[(size_t)k * 36 + 24 + 2 * q2] = (uint32_t)one_dev.v[q2];
          parts[(size_t)k * 36 + 24 + 2 * q2 + 1] = (uint32_t)(one_dev.v[q2] >> 32);
        }
      }
    }
    for (int k = 0; k < K; k++) {
      const uint32_t* q = &parts[(size_t)k * 36];
      bool zero_z = true;
      for (int j = 0; j < 12; j++) zero_z &= q[24 + j] == 0;
      const msm_host::Proj6 P = zero_z ? ctx->hc.zero() : partial_to_host(ctx, q);
      msm_host::Fe6 one = {{1, 0, 0, 0, 0, 0}}, t;
      ctx->hc.F.mul(t, P.X, one); fe6_to_bytes(partials_out + (size_t)k * 144, t);
      ctx->hc.F.mul(t, P.Y, one); fe6_to_bytes(partials_out + (size_t)k * 144 + 48, t);
      ctx->hc.F.mul(t, P.Z, one); fe6_to_bytes(partials_out + (size_t)k * 144 + 96, t);
    }
    if (ms_out) *ms_out = ms;
    for (DevBuf* b : {&wire, &rows, &planes, &desc, &scr}) ctx->release(*b);
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_test_batch_add_mode(msm_ctx* ctx, const uint8_t* g, const uint8_t* h, uint8_t* out, uint64_t n, int mode, uint32_t steps) {
  if (!ctx || !g || !h || !out || n == 0 || steps == 0) return fail(ctx, MSM_ERR_ARG, "msm_test_batch_add_mode: bad argument");
  if (ctx->is_te() || (mode != MODE_REGULAR && mode != MODE_SEARCH))
    return fail(ctx, MSM_ERR_ARG, "msm_test_batch_add_mode: Weierstrass curves, mode 1 (regular) or 2 (search)");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    // element 2e = G_e, 2e + 1 = H_e in plane layout; search mode: an all-zero H_e is passed as "no second operand"
    const uint64_t T = ((n + steps - 1) / steps + 255) / 256 * 256;
    const uint64_t in_cap = 2 * (uint64_t)steps * T;     // idle lanes of the last step read (and ignore) up to here
    const size_t pb = 2 * ctx->coord_bytes();            // wire point = tree node
    DevBuf rows, wire, planes, outb, scr, desc;
    ctx->ensure(wire, 2 * n * pb);
    ctx->ensure(rows, 2 * n * ROW_WORDS * 4);
    ctx->ensure(planes, in_cap * pb);
    ctx->ensure(outb, (uint64_t)steps * T * pb);
    ctx->ensure(scr, (size_t)steps * NL * T * 4);
    std::vector<uint8_t> inter(2 * n * pb);
    std::vector<uint32_t> hd(n);
    for (uint64_t i = 0; i < n; i++) {
      memcpy(&inter[(2 * i) * pb], g + i * pb, pb);
      memcpy(&inter[(2 * i + 1) * pb], h + i * pb, pb);
      bool hz = true;
      for (size_t j = 0; j < pb; j++) hz = hz && h[i * pb + j] == 0;
      hd[i] = (uint32_t)((2 * i) << 1) | (hz ? 0u : 1u);
    }
    HIPCHK(hipMemcpyAsync(wire.p, inter.data(), inter.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->errflag.p, 0, 4, ctx->stream));
    HIPCHK(hipMemsetAsync(planes.p, 0, in_cap * pb, ctx->stream));
    W_LAUNCH(ctx, k_points_from_wire, dim3((uint32_t)((2 * n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)rows.p,
                       (const uint32_t*)wire.p, 2 * n, 0, (uint32_t*)ctx->errflag.p);
    ROWS_TO_PLANES(ctx, dim3((uint32_t)((2 * n + 255) / 256)), dim3(256), 0, ctx->stream, (uint4*)planes.p, in_cap,
                   (const uint32_t*)rows.p, (uint32_t)(2 * n));
    BatchArgs a{};
    a.in = (const uint4*)planes.p;
    a.in_cap = in_cap;
    a.out = (uint4*)outb.p;
    a.out_cap = (uint64_t)steps * T;
    a.scratch = (uint32_t*)scr.p;
    a.sstride = T;
    a.n_out = n;
    a.steps = steps;
    if (mode == MODE_SEARCH) {
      ctx->ensure(desc, n * 4);
      HIPCHK(hipMemcpyAsync(desc.p, hd.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
      a.desc = (const uint32_t*)desc.p;
      W_LAUNCH_MODE(ctx, k_batch_add, MODE_SEARCH, dim3((uint32_t)(T / 256)), dim3(256), 0, ctx->stream, a);
    } else {
      W_LAUNCH_MODE(ctx, k_batch_add, MODE_REGULAR, dim3((uint32_t)(T / 256)), dim3(256), 0, ctx->stream, a);
    }
    std::vector<uint32_t> pl(a.out_cap * pb / 4);
    HIPCHK(hipMemcpyAsync(pl.data(), outb.p, pl.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    for (uint64_t e = 0; e < n; e++) plane_element_to_wire(ctx, pl.data(), a.out_cap, e, out + e * pb);
    for (DevBuf* b : {&rows, &wire, &planes, &outb, &scr, &desc}) ctx->release(*b);
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_generate_points(msm_ctx* ctx, uint64_t n, uint64_t seed, uint8_t* a_out) {
  if (!ctx) return MSM_ERR_ARG;
  if (ctx->children.empty()) return generate_points_one(ctx, n, seed, a_out);
  try {   // the generator is deterministic in (seed, index): every device builds the identical set
    return on_all_devices(ctx, [&](msm_ctx* c) { return generate_points_one(c, n, seed, c == ctx ? a_out : nullptr); });
  } MSM_CATCH_ALL(ctx)
}

int msm_generate_scalars(msm_ctx* ctx, uint64_t n, uint64_t seed, void* dev_dst, uint8_t* host_out) {
  if (!ctx) return MSM_ERR_ARG;
  if (!dev_dst && !host_out) return fail(ctx, MSM_ERR_ARG, "msm_generate_scalars: neither a device nor a host destination");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    return msm_gen::generate_scalars(ctx, n, seed, dev_dst, host_out);
  } MSM_CATCH_ALL(ctx)
}

}  // extern "C"
