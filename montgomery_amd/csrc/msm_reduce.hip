// Bucket reduction of a window group (P_k = sum_l l B_(k,l): reduceBucketsColumnProjective + partition sums,
// src/msm-batched-affine.ts:556-583, :312-319) and the host tail: Horner combination of the window sums, projective ->
// affine (:322-333, src/curve-projective.ts:335-349), msm_combine.
#include "msm_internal.h"

using namespace msm;
using namespace msmi;

namespace msmi {

void words_to_fe6(msm_host::Fe6& r, const uint32_t* w, int nw) {   // nw packed words, zero-extended
  for (int i = 0; i < 6; i++)
    r.v[i] = (2 * i < nw ? (uint64_t)w[2 * i] : 0) | ((2 * i + 1 < nw ? (uint64_t)w[2 * i + 1] : 0) << 32);
}

void fe6_to_bytes(uint8_t* out, const msm_host::Fe6& a) {
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(a.v[i] >> (8 * j));
}

// Window sums travel as 36 words (X, Y, Z: 12 packed words each, device Montgomery form, below 2p).  A projective point is a
// class of triples, so the host neither converts them on the way in nor on the way out: read as host Montgomery values the
// three words of a device point carry one common factor (2^6 for 13 limbs), which is the same point, and so is every sum the
// host forms of such points.  Only affine values need the exact radix (horner_to_affine divides the factor out with Z).
msm_host::Proj6 partial_to_host(const msm_ctx* ctx, const uint32_t* w) {
  const auto& F = ctx->hc.F;
  msm_host::Proj6 P;
  msm_host::Fe6* co[3] = {&P.X, &P.Y, &P.Z};
  for (int j = 0; j < 3; j++) {
    words_to_fe6(*co[j], w + 12 * j);
    while (msm_host::Field6::ge(*co[j], F.p)) F.sub_raw(*co[j], *co[j], F.p);
  }
  return P;
}

// host projective point -> the packed form the pipeline carries (canonical values; see above for the radix)
void host_to_partial(const msm_ctx*, const msm_host::Proj6& P, uint32_t* out36) {
  const msm_host::Fe6* co[3] = {&P.X, &P.Y, &P.Z};
  for (int j = 0; j < 3; j++)
    for (int q = 0; q < 6; q++) {
      out36[12 * j + 2 * q] = (uint32_t)co[j]->v[q];
      out36[12 * j + 2 * q + 1] = (uint32_t)(co[j]->v[q] >> 32);
    }
}

// Bucket reduction of kc windows of L buckets each: P_k = sum_l l * B_(k,l) (reduceBucketsColumnProjective + the partition
// sums, src/msm-batched-affine.ts:556-583, :312-319) -> h_partials_out, kc x 36 (32 on the Edwards path) words.  The bucket
// sums come as projective points from k_bucket_finish (`bucket_proj`) or as the first element of every bucket in the tree
// buffer (`fin`, `off_fin`).  Runs on w.stream, records w.ev[4] behind its last kernel and returns when the sums are on the host.
// stride: bits a window advances by (Plan::c); 0 = log2(L) + 1, the plain plan's c.
void reduce_buckets(msm_ctx* ctx, msm_ctx::Workspace& w, const uint4* fin, uint64_t fin_cap, const uint32_t* off_fin,
                    const uint32_t* bucket_proj, uint32_t L, int kc, uint32_t* h_partials_out, bool merged, int stride) {
  hipStream_t s = w.stream;
  const bool te = ctx->is_te();
  const uint64_t nb = (uint64_t)kc * L;
  const int part_words = te ? 32 : 36;
  // buckets per lane: enough lanes to fill the chip, but never more than 16 buckets deep (2 additions each)
  // (millions of buckets -- the big windows -- go 32 deep: 2^26 at c = 22, with the two policies below, 150.6 -> 149.3 ms)
  uint32_t TC = 2;
  const uint32_t tc_cap = nb >= (1ull << 22) ? 32 : 16;
  while (TC < tc_cap && nb / TC > 65536) TC *= 2;
  MSM_KNOB(TC, "MSM_TC", 1);
  TC = std::min<uint32_t>(TC, L);
  uint32_t nchunks = (L + TC - 1) / TC;
  // bit-sliced weighting (Weierstrass path, enough chunks to matter, TC a power of two)
  uint32_t nbits = 0;
  while ((1u << nbits) < nchunks) nbits++;
  // bit-sliced weighting (enough chunks to matter, TC a power of two)
  // (the host finishes every window with ~2 c additions and doublings: on the Edwards path, whose plain plans have 13 - 28
  // windows, that tail costs more than the weighting chains it replaces -- 0.28 against 0.05 ms at 2^20 -- so it takes the
  // bit-sliced form only for the one or two merged windows of a run on window tables)
  const bool bit_sliced = nchunks >= 64 && (TC & (TC - 1)) == 0 && (!te || kc <= 2);
  const size_t raw_words = te ? (size_t)4 * te::TL : (size_t)3 * NL;   // raw limb words of one point between the reduction kernels
  ctx->ensure(w.columns, (size_t)kc * nchunks * 4 * NL * 4);
  ctx->ensure(w.partials, (size_t)kc * 36 * 4);
  {
    uint32_t threads = nchunks * (uint32_t)kc;
    if (bit_sliced) {
      ctx->ensure(w.rows_sum, (size_t)kc * nchunks * raw_words * 4);
      if (te)
        hipLaunchKernelGGL(te::k_te_bucket_reduce, dim3((threads + 63) / 64), dim3(64), 0, s, (uint32_t*)w.columns.p, (uint32_t*)w.rows_sum.p,
                           fin, fin_cap, off_fin, bucket_proj, L, TC, nchunks, (uint32_t)kc);
      else
        W_LAUNCH(ctx, k_bucket_reduce, dim3((threads + 63) / 64), dim3(64), 0, s, (uint32_t*)w.columns.p,
                           (uint32_t*)w.rows_sum.p, fin, fin_cap, off_fin, bucket_proj, L, TC, nchunks, (uint32_t)kc);
      // first stage: one wave per 512 elements (8 per lane; at 2^20 that is about one wave per SIMD) -- the masked sums
      // have half as many elements as the triangle sum and get half as many blocks; second stage: one wave per
      // (window, bit) over the block sums it finds (the slots a masked sum did not fill are never read)
      const uint32_t nblk = std::max<uint32_t>(2, nchunks / (8 * BT_THREADS));
      ctx->ensure(w.partials, (size_t)kc * (nbits + 1) * 36 * 4);
      // Two-dimensional form (msm_kernels.h, bit_tree_body): row sums A_hi and column sums B_lo of the chunk matrix first --
      // 2 additions per chunk instead of nbits / 2 -- then the nbits masked sums over those 2^(nbits / 2) + ... points.  Measured
      // (profiles/r06_experiments.txt): 2^20 on tables 0.42 -> 0.2x ms for the two launches, 2^26 0.5 -> 0.3x per window group.
      const bool two_d = nbits >= 6 && (1u << nbits) == nchunks;
      if (two_d) {
        const uint32_t M_hi = 1u << (nbits / 2), M_lo = 1u << (nbits - nbits / 2);
        const uint32_t per_kk = M_hi + M_lo + nblk;
        ctx->ensure(w.columns2, (size_t)kc * per_kk * raw_words * 4);
        if (te) {
          hipLaunchKernelGGL(te::k_te_bit_tree, dim3(per_kk, 1, kc), dim3(BT_THREADS), 0, s, (uint32_t*)w.columns2.p,
                             (const uint32_t*)w.rows_sum.p, (const uint32_t*)w.columns.p, nchunks, nbits, 3, 0, nblk);
          hipLaunchKernelGGL(te::k_te_bit_tree, dim3(1, nbits + 1, kc), dim3(BT_THREADS), 0, s, (uint32_t*)w.partials.p,
                             (const uint32_t*)w.columns2.p, (const uint32_t*)nullptr, nchunks, nbits, 4, 1, nblk);
        } else {
          W_LAUNCH(ctx, k_bit_tree, dim3(per_kk, 1, kc), dim3(BT_THREADS), 0, s, (uint32_t*)w.columns2.p,
                             (const uint32_t*)w.rows_sum.p, (const uint32_t*)w.columns.p, nchunks, nbits, 3, 0, nblk);
          W_LAUNCH(ctx, k_bit_tree, dim3(1, nbits + 1, kc), dim3(BT_THREADS), 0, s, (uint32_t*)w.partials.p,
                             (const uint32_t*)w.columns2.p, (const uint32_t*)nullptr, nchunks, nbits, 4, 1, nblk);
        }
      } else {
      ctx->ensure(w.columns2, (size_t)kc * (nbits + 1) * nblk * raw_words * 4);
      if (te) {
        hipLaunchKernelGGL(te::k_te_bit_tree, dim3(nbits * (nblk / 2) + nblk, 1, kc), dim3(BT_THREADS), 0, s, (uint32_t*)w.columns2.p,
                           (const uint32_t*)w.rows_sum.p, (const uint32_t*)w.columns.p, nchunks, nbits, 1, 0, nblk);
        hipLaunchKernelGGL(te::k_te_bit_tree, dim3(1, nbits + 1, kc), dim3(BT_THREADS), 0, s, (uint32_t*)w.partials.p,
                           (const uint32_t*)w.columns2.p, (const uint32_t*)nullptr, nblk, nbits, 2, 1, 1u);
      } else {
        W_LAUNCH(ctx, k_bit_tree, dim3(nbits * (nblk / 2) + nblk, 1, kc), dim3(BT_THREADS), 0, s, (uint32_t*)w.columns2.p,
                           (const uint32_t*)w.rows_sum.p, (const uint32_t*)w.columns.p, nchunks, nbits, 1, 0, nblk);
        W_LAUNCH(ctx, k_bit_tree, dim3(1, nbits + 1, kc), dim3(BT_THREADS), 0, s, (uint32_t*)w.partials.p,
                           (const uint32_t*)w.columns2.p, (const uint32_t*)nullptr, nblk, nbits, 2, 1, 1u);
      }
      }
    } else if (te) {
      hipLaunchKernelGGL(te::k_te_bucket_reduce, dim3((threads + 63) / 64), dim3(64), 0, s, (uint32_t*)w.columns.p, (uint32_t*)nullptr,
                         fin, fin_cap, off_fin, bucket_proj, L, TC, nchunks, (uint32_t)kc);
      hipLaunchKernelGGL(te::k_te_window_sum, dim3(kc), dim3(te::TE_WS_THREADS), 0, s, (uint32_t*)w.partials.p,
                         (const uint32_t*)w.columns.p, nchunks);
    } else {
      W_LAUNCH(ctx, k_bucket_reduce, dim3((threads + 63) / 64), dim3(64), 0, s, (uint32_t*)w.columns.p, (uint32_t*)nullptr,
                         fin, fin_cap, off_fin, bucket_proj, L, TC, nchunks, (uint32_t)kc);
      if (nchunks > 2 * WS_THREADS) {
        // two-stage: blocks of 2 columns per lane, then one block per window over the block sums
        const uint32_t per_block = 2 * WS_THREADS;
        const uint32_t nblk = (nchunks + per_block - 1) / per_block;
        ctx->ensure(w.columns2, (size_t)kc * nblk * 3 * NL * 4);
        W_LAUNCH(ctx, k_column_tree, dim3(nblk, kc), dim3(WS_THREADS), 0, s, (uint32_t*)w.columns2.p,
                           (const uint32_t*)w.columns.p, nchunks, per_block);
        W_LAUNCH(ctx, k_window_sum, dim3(kc), dim3(WS_THREADS), 0, s, (uint32_t*)w.partials.p,
                           (const uint32_t*)w.columns2.p, nblk);
      } else {
        W_LAUNCH(ctx, k_window_sum, dim3(kc), dim3(WS_THREADS), 0, s, (uint32_t*)w.partials.p,
                           (const uint32_t*)w.columns.p, nchunks);
      }
    }
  }
  if (bit_sliced) {
    // read the (nbits + 1) sums per window back and finish P_k = tri + TC * sum_b 2^b S_b on the host
    HIPCHK(hipMemcpyAsync(w.h_part, w.partials.p, (size_t)kc * (nbits + 1) * part_words * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(w.ev[4], s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    uint32_t lt = 0, cbits = 1;
    while ((1u << lt) < TC) lt++;
    while ((1u << (cbits - 1)) < L) cbits++;
    const int adv = stride ? stride : (int)cbits;
    // one implementation for both kinds of point: zero, doubling, addition, a packed sum in, a packed sum out
    auto finish = [&](auto zero, auto dbl, auto add, auto from_part, auto to_part, auto ident_part) {
      if (merged) {
        // The caller only wants S_g = sum_kk 2^(c kk) P_kk of the whole group (a full MSM on this device: the Horner step over
        // the windows follows anyway).  One double-and-add pass over the c kc bit positions then does both jobs -- inside window
        // kk the sum S_b sits at bit log2(TC) + b, the triangle sum at bit 0 -- with c kc doublings instead of (c - 1) kc for
        // the windows plus c (kc - 1) for their combination.  S_g goes into the slot of the group's first window, the identity
        // into the others: sum_k 2^(c k) (slot k) is the same group element as with one P_k per slot.
        // With a folded top window (Plan::fold) a window has one bit position more than it advances by: position `stride` of
        // window kk coincides with position 0 of window kk + 1, so the pass walks GLOBAL bit positions and adds what every
        // window has there.
        auto acc = zero();
        for (int gpos = (kc - 1) * adv + (int)cbits - 1; gpos >= 0; gpos--) {
          acc = dbl(acc);
          for (int kk = kc - 1; kk >= 0; kk--) {
            const int pos = gpos - kk * adv;
            if (pos < 0 || pos >= (int)cbits) continue;
            const uint32_t* base = w.h_part + (size_t)kk * (nbits + 1) * part_words;
            const int b = pos - (int)lt;
            if (b >= 0 && b < (int)nbits) acc = add(acc, from_part(base + (size_t)b * part_words));
            if (pos == 0) acc = add(acc, from_part(base + (size_t)nbits * part_words));
          }
        }
        for (int kk = 1; kk < kc; kk++) ident_part(h_partials_out + (size_t)kk * part_words);
        to_part(acc, h_partials_out);
        return;
      }
      for (int kk = 0; kk < kc; kk++) {
        const uint32_t* base = w.h_part + (size_t)kk * (nbits + 1) * part_words;
        auto acc = zero();
        for (int b = (int)nbits - 1; b >= 0; b--) {
          acc = dbl(acc);
          acc = add(acc, from_part(base + (size_t)b * part_words));
        }
        for (uint32_t t = TC; t > 1; t >>= 1) acc = dbl(acc);   // TC is a power of two on this path
        acc = add(acc, from_part(base + (size_t)nbits * part_words));
        to_part(acc, h_partials_out + (size_t)kk * part_words);
      }
    };
    if (te) {
      const auto& C = ctx->hte;
      finish([&] { return C.zero(); }, [&](const msm_host::Ext6& a) { return C.add(a, a); },
             [&](const msm_host::Ext6& a, const msm_host::Ext6& b) { return C.add(a, b); },
             [&](const uint32_t* p) { return te_partial_to_host(ctx, p); },
             [&](const msm_host::Ext6& a, uint32_t* out) { te_host_to_partial(ctx, a, out); },
             [&](uint32_t* out) { te_host_to_partial(ctx, C.zero(), out); });
    } else {
      const auto& C = ctx->hc;
      finish([&] { return C.zero(); }, [&](const msm_host::Proj6& a) { return C.dbl(a); },
             [&](const msm_host::Proj6& a, const msm_host::Proj6& b) { return C.add(a, b); },
             [&](const uint32_t* p) { return partial_to_host(ctx, p); },
             [&](const msm_host::Proj6& a, uint32_t* out) { host_to_partial(ctx, a, out); },
             [&](uint32_t* out) { memset(out, 0, 36 * 4); });
    }
    return;
  }
  HIPCHK(hipMemcpyAsync(w.h_part, w.partials.p, (size_t)kc * part_words * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipEventRecord(w.ev[4], s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  memcpy(h_partials_out, w.h_part, (size_t)kc * part_words * 4);
}

void plane_element_to_wire(const msm_ctx* ctx, const uint32_t* planes, uint64_t cap, uint64_t e, uint8_t* out_xy) {
  const int nw = ctx->nw(), np = nw / 4;
  const size_t cb = ctx->coord_bytes();
  uint32_t w[24];
  for (int cpl = 0; cpl < 2 * np; cpl++)
    for (int q = 0; q < 4; q++) w[4 * cpl + q] = planes[((uint64_t)cpl * cap + e) * 4 + q];
  memset(out_xy, 0, 2 * cb);
  if (w[nw - 1] == INF_WORD) return;
  msm_host::Fe6 one = {{1, 0, 0, 0, 0, 0}};
  for (int j = 0; j < 2; j++) {
    msm_host::Fe6 t;
    words_to_fe6(t, w + nw * j, nw);
    ctx->hc.F.mul(t, t, ctx->k_dev_to_host);
    ctx->hc.F.mul(t, t, one);
    uint8_t b48[48];
    fe6_to_bytes(b48, t);
    memcpy(out_xy + cb * j, b48, cb);
  }
}

// S = sum_k 2^(ck) P_k (src/msm-batched-affine.ts:322-333)
msm_host::Proj6 horner_points(const msm_host::Curve6& C, const std::vector<msm_host::Proj6>& P, int c) {
  int K = (int)P.size();
  msm_host::Proj6 acc = P[K - 1];
  for (int k = K - 2; k >= 0; k--) {
    for (int j = 0; j < c; j++) acc = C.dbl(acc);
    acc = C.add(acc, P[k]);
  }
  return acc;
}

// projective -> the canonical affine result (src/curve-projective.ts:335-349)
void proj_to_result(const msm_host::Curve6& C, const msm_host::Proj6& acc, msm_result* out) {
  memset(out->x, 0, 48);
  memset(out->y, 0, 48);
  if (C.is_zero(acc)) {
    out->is_infinity = 1;
    return;
  }
  out->is_infinity = 0;
  msm_host::Fe6 zi, x, y, one = {{1, 0, 0, 0, 0, 0}};
  C.F.inv(zi, acc.Z);
  C.F.mul(x, acc.X, zi);
  C.F.mul(y, acc.Y, zi);
  C.F.mul(x, x, one);  // leave Montgomery form
  C.F.mul(y, y, one);
  fe6_to_bytes(out->x, x);
  fe6_to_bytes(out->y, y);
}

void horner_to_affine(const msm_host::Curve6& C, const std::vector<msm_host::Proj6>& P, int c, msm_result* out) {
  proj_to_result(C, horner_points(C, P, c), out);
}

// twisted Edwards tail: S = sum_k 2^(ck) P_k with unified additions (src/msm-basic.ts:142-158), then x = X/Z, y = Y/Z
void te_horner_points(const msm_host::TeCurve6& C, const std::vector<msm_host::Ext6>& P, int c, msm_result* out) {
  const int K = (int)P.size();
  msm_host::Ext6 acc = P[K - 1];
  for (int k = K - 2; k >= 0; k--) {
    for (int j = 0; j < c; j++) acc = C.add(acc, acc);
    acc = C.add(acc, P[k]);
  }
  msm_host::Fe6 zi, x, y, one = {{1, 0, 0, 0, 0, 0}};
  C.F.inv(zi, acc.Z);
  C.F.mul(x, acc.X, zi);
  C.F.mul(y, acc.Y, zi);
  C.F.mul(x, x, one);
  C.F.mul(y, y, one);
  memset(out->x, 0, 48);
  memset(out->y, 0, 48);
  fe6_to_bytes(out->x, x);
  fe6_to_bytes(out->y, y);
  out->is_infinity = 0;
}

// device window sum (X, Y, Z, T: 8 words each, below 2p) -> host extended point: as for the Weierstrass sums no change of
// radix -- (X, Y, Z, T) with T = X Y / Z stays a valid extended point when all four carry one common factor
msm_host::Ext6 te_partial_to_host(const msm_ctx* ctx, const uint32_t* w) {
  const auto& C = ctx->hte;
  msm_host::Ext6 P;
  msm_host::Fe6* dst[4] = {&P.X, &P.Y, &P.Z, &P.T};
  for (int j = 0; j < 4; j++) {
    msm_host::Fe6 t = {{0, 0, 0, 0, 0, 0}};
    for (int i = 0; i < 4; i++) t.v[i] = (uint64_t)w[8 * j + 2 * i] | ((uint64_t)w[8 * j + 2 * i + 1] << 32);
    while (msm_host::Field6::ge(t, C.F.p)) C.F.sub_raw(t, t, C.F.p);
    *dst[j] = t;
  }
  return P;
}

// host extended point -> the window-sum form (X, Y, Z, T: 8 words each)
void te_host_to_partial(const msm_ctx*, const msm_host::Ext6& P, uint32_t* out32) {
  const msm_host::Fe6* co[4] = {&P.X, &P.Y, &P.Z, &P.T};
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++) {
      out32[8 * j + 2 * i] = (uint32_t)co[j]->v[i];
      out32[8 * j + 2 * i + 1] = (uint32_t)(co[j]->v[i] >> 32);
    }
}

void te_horner_to_affine(const msm_ctx* ctx, const std::vector<uint32_t>& words, int K, int c, msm_result* out) {
  std::vector<msm_host::Ext6> P(K);
  for (int k = 0; k < K; k++) P[k] = te_partial_to_host(ctx, &words[(size_t)k * 32]);
  te_horner_points(ctx->hte, P, c, out);
}

// host curve constants without a context (rank 0 of a sharded run may combine without touching a GPU)
const msm_host::Curve6* static_host_curve(int curve) {
  static msm_host::Curve6 hc[4];
  static std::atomic<int> ready[4];
  if (curve < 0 || curve > 3 || curve == MSM_CURVE_ED_ON_BLS12_377) return nullptr;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (!ready[curve].load()) {
    hc[curve].F.init(curve_info(curve).pw);
    ready[curve].store(1);
  }
  return &hc[curve];
}

// twisted Edwards: (X : Y : Z) in, T rebuilt as (X Z : Y Z : Z^2 : X Y), then the unified-addition Horner
int te_combine_impl(const uint8_t* partials, int32_t K, int32_t c, msm_result* out, int32_t G) {
  static msm_host::TeCurve6 C;
  static std::once_flag once;
  std::call_once(once, [] {
    uint32_t pw[12] = {0};
    for (int i = 0; i < 8; i++) pw[i] = Fp253::PW[i];
    C.init(pw, 3021);
  });
  std::vector<msm_host::Ext6> P(K);
  for (int k = 0; k < K; k++) {
    P[k] = C.zero();
    for (int g = 0; g < G; g++) {   // group g's sum of window k
      msm_host::Fe6 t[3];
      for (int j = 0; j < 3; j++) {
        const uint8_t* b = partials + ((size_t)g * K + k) * 144 + 48 * j;
        for (int i = 0; i < 6; i++) {
          uint64_t v = 0;
          for (int q = 0; q < 8; q++) v |= (uint64_t)b[8 * i + q] << (8 * q);
          t[j].v[i] = v;
        }
        if (msm_host::Field6::ge(t[j], C.F.p)) return MSM_ERR_ARG;
        C.F.mul(t[j], t[j], C.F.r2);
      }
      msm_host::Ext6 Q;
      C.F.mul(Q.X, t[0], t[2]);
      C.F.mul(Q.Y, t[1], t[2]);
      C.F.mul(Q.Z, t[2], t[2]);
      C.F.mul(Q.T, t[0], t[1]);
      P[k] = G == 1 ? Q : C.add(P[k], Q);
    }
  }
  memset(out, 0, sizeof(*out));
  te_horner_points(C, P, c, out);
  out->c = c;
  out->K = K;
  return MSM_OK;
}

int combine_impl(msm_ctx* ctx, const msm_host::Curve6& C, const uint8_t* partials, int32_t K, int32_t c, msm_result* out, int32_t G) {
  std::vector<msm_host::Proj6> P(K);
  for (int k = 0; k < K; k++) {
    P[k] = C.zero();
    for (int g = 0; g < G; g++) {   // group g's sum of window k
      msm_host::Fe6 t[3];
      for (int j = 0; j < 3; j++) {
        const uint8_t* b = partials + ((size_t)g * K + k) * 144 + 48 * j;
        for (int i = 0; i < 6; i++) {
          uint64_t v = 0;
          for (int q = 0; q < 8; q++) v |= (uint64_t)b[8 * i + q] << (8 * q);
          t[j].v[i] = v;
        }
        if (msm_host::Field6::ge(t[j], C.F.p)) return fail(ctx, MSM_ERR_ARG, "msm_combine: coordinate >= p");
        C.F.mul(t[j], t[j], C.F.r2);  // to host Montgomery form
      }
      msm_host::Proj6 Q;
      Q.X = t[0]; Q.Y = t[1]; Q.Z = t[2];
      P[k] = G == 1 ? Q : C.add(P[k], Q);
    }
  }
  memset(out, 0, sizeof(*out));
  horner_to_affine(C, P, c, out);
  out->c = c;
  out->K = K;
  return MSM_OK;
}

}  // namespace msmi
