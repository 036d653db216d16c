// HIP kernels of the generic ("msmBasic") MSM path for the twisted Edwards curve Ed-on-BLS12-377
// (a = -1, d = 3021) in extended coordinates (X, Y, Z, T).
//
// Reference: src/msm-basic.ts:45-211 (digit map :72-91, bucket accumulation with mixed add / sub
// :103-123, reduceBucketsChunk :180-211, Horner :142-158) over src/curve-twisted-edwards.ts:84-165
// (unified add-2008-hwcd-3 with k = 2d; subtraction by negating X2 and T2; mixed form when Z2 = 1).
//
// The GPU keeps the reference's arithmetic (unified extended additions, no inversions, identity is
// the ordinary point (0, 1, 1, 0), src/bigint/twisted-edwards.ts:34) but not its loop shape: the
// reference lets every thread scan all N points for its bucket chunk; here the digits are counting-
// sorted exactly like on the Weierstrass path (same k_hist / k_colscan / k_scan / k_scatter_lds) and
// each bucket is summed by the same padded halving tree, whose node operation is the unified
// addition -- no edge cases, so padding with the identity needs no flags at all.
//
// Field: 253-bit prime (the BLS12-377 scalar field), 9 x 30-bit limbs, R = 2^270 >= 2^17 p: every
// sum / difference below feeds a multiplication unreduced, and tree nodes are stored as the raw
// multiplication outputs (< 2p, 8 packed words per coordinate); nothing on this path compares values.
//
// Layouts: point rows 128 B = [x | y | t = xy | k t], Montgomery form; tree buffers: 8 planes of 16 B
// (X, Y, Z, T two planes each); payload = (point index << 1) | negative.
#pragma once
#include "msm_kernels.h"
#include "msm_gen_kernels.h"

namespace msm {
namespace te {

using FT = Fp253;
struct CvEdField { using F = Fp253; };   // for the field-only test kernels of msm_kernels.h (k_test_batch_inverse)
constexpr int TL = FT::NL;   // 9
constexpr int TW = FT::NW;   // 8
constexpr int TE_ROW_WORDS = 32;

struct Ext {
  Fe<FT> X, Y, Z, T;
};

// constant tables are only ever value-used (compile-time constants in device code)
#define TE_CONST(dst, NAME)                                \
  do {                                                     \
    _Pragma("unroll") for (int _l = 0; _l < TL; _l++)(dst).l[_l] = FT::NAME[_l]; \
  } while (0)

MSM_DEV void te_set_identity(Ext& P) {
  fe_set_zero<FT>(P.X);
  fe_set_one<FT>(P.Y);
  fe_set_one<FT>(P.Z);
  fe_set_zero<FT>(P.T);
}

MSM_DEV void load_words8(uint32_t (&w)[TW], const uint32_t* p) {
  const uint4* p4 = reinterpret_cast<const uint4*>(p);
  uint4 a = p4[0], b = p4[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}

MSM_DEV void load_coord_planes(Fe<FT>& r, const uint4* base, uint64_t cap, int coord, uint64_t e) {
  uint32_t w[TW];
  uint4 a = base[(uint64_t)(2 * coord) * cap + e], b = base[(uint64_t)(2 * coord + 1) * cap + e];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  fe_unpack<FT>(r, w);
}

MSM_DEV void store_coord_planes(uint4* base, uint64_t cap, int coord, uint64_t e, const Fe<FT>& v) {
  uint32_t w[TW];
  fe_pack<FT>(w, v);
  base[(uint64_t)(2 * coord) * cap + e] = make_uint4(w[0], w[1], w[2], w[3]);
  base[(uint64_t)(2 * coord + 1) * cap + e] = make_uint4(w[4], w[5], w[6], w[7]);
}

MSM_DEV void load_ext(Ext& P, const uint4* base, uint64_t cap, uint64_t e) {
  load_coord_planes(P.X, base, cap, 0, e);
  load_coord_planes(P.Y, base, cap, 1, e);
  load_coord_planes(P.Z, base, cap, 2, e);
  load_coord_planes(P.T, base, cap, 3, e);
}
MSM_DEV void store_ext(uint4* base, uint64_t cap, uint64_t e, const Ext& P) {
  store_coord_planes(base, cap, 0, e, P.X);
  store_coord_planes(base, cap, 1, e, P.Y);
  store_coord_planes(base, cap, 2, e, P.Z);
  store_coord_planes(base, cap, 3, e, P.T);
}

// tail of add-2008-hwcd-3 from A, B, C, D (all < 2p): 4 multiplications
MSM_DEV void te_finish(Ext& R, const Fe<FT>& A, const Fe<FT>& B, const Fe<FT>& C, const Fe<FT>& D) {
  Fe<FT> E, Fv, G, H;
  fe_sub_2p<FT>(E, B, A);
  fe_sub_2p<FT>(Fv, D, C);
  fe_add<FT>(G, D, C);
  fe_add<FT>(H, B, A);
  fe_mul<FT>(R.X, E, Fv);
  fe_mul<FT>(R.Y, G, H);
  fe_mul<FT>(R.T, E, H);
  fe_mul<FT>(R.Z, Fv, G);
}

// general unified addition, 9M (src/curve-twisted-edwards.ts:84-165); coordinates < 2p in and out
MSM_DEV void te_add(Ext& R, const Ext& P, const Ext& Q) {
  Fe<FT> a, b, A, B, C, D, k;
  fe_sub_2p<FT>(a, P.Y, P.X);
  fe_sub_2p<FT>(b, Q.Y, Q.X);
  fe_mul<FT>(A, a, b);
  fe_add<FT>(a, P.Y, P.X);
  fe_add<FT>(b, Q.Y, Q.X);
  fe_mul<FT>(B, a, b);
  TE_CONST(k, K2DL);
  fe_mul<FT>(C, P.T, Q.T);
  fe_mul<FT>(C, C, k);
  fe_mul<FT>(D, P.Z, Q.Z);
  fe_add<FT>(D, D, D);
  te_finish(R, A, B, C, D);
}

// both operands affine rows (Z = 1) with precomputed k*t: 7M
struct AffRow {
  Fe<FT> x, y, t, kt;
};
MSM_DEV void te_add_rows(Ext& R, const AffRow& P, const AffRow& Q) {
  Fe<FT> a, b, A, B, C, D;
  fe_sub_2p<FT>(a, P.y, P.x);
  fe_sub_2p<FT>(b, Q.y, Q.x);
  fe_mul<FT>(A, a, b);
  fe_add<FT>(a, P.y, P.x);
  fe_add<FT>(b, Q.y, Q.x);
  fe_mul<FT>(B, a, b);
  fe_mul<FT>(C, P.t, Q.kt);
  fe_set_one<FT>(D);
  fe_add<FT>(D, D, D);
  te_finish(R, A, B, C, D);
}

// row -> registers; neg: (x, y, t, kt) -> (-x, y, -t, -kt); absent: identity (0, 1, 0, 0)
MSM_DEV void load_row(AffRow& P, const uint32_t* rows, uint32_t payload) {
  if (payload == SLOT_EMPTY) {
    fe_set_zero<FT>(P.x);
    fe_set_one<FT>(P.y);
    fe_set_zero<FT>(P.t);
    fe_set_zero<FT>(P.kt);
    return;
  }
  const uint32_t* row = rows + (uint64_t)(payload >> 1) * TE_ROW_WORDS;
  uint32_t w[TW];
  load_words8(w, row);      fe_unpack<FT>(P.x, w);
  load_words8(w, row + 8);  fe_unpack<FT>(P.y, w);
  load_words8(w, row + 16); fe_unpack<FT>(P.t, w);
  load_words8(w, row + 24); fe_unpack<FT>(P.kt, w);
  if (payload & 1u) {
    Fe<FT> z;
    fe_set_zero<FT>(z);
    fe_sub_2p<FT>(P.x, z, P.x);    // 2p - x: any representative works, nothing is compared
    fe_sub_2p<FT>(P.t, z, P.t);
    fe_sub_2p<FT>(P.kt, z, P.kt);
  }
}

// ---------------------------------------------------------------------------------------------
// k_te_points_from_wire: N x (x || y) 32-byte little-endian -> rows [x | y | xy | 2d xy]
// ---------------------------------------------------------------------------------------------

MSM_DEV bool words8_ge_p(const uint32_t (&w)[TW]) {
  bool gt = false, lt = false;
#pragma unroll
  for (int j = TW - 1; j >= 0; j--) {
    if (!gt && !lt) {
      if (w[j] > FT::PW[j]) gt = true;
      else if (w[j] < FT::PW[j]) lt = true;
    }
  }
  return !lt;
}

__global__ void __launch_bounds__(256) k_te_points_from_wire(uint32_t* rows, const uint32_t* wire, uint64_t n, int check_curve,
                                                             uint32_t* err)
#ifndef MSM_TE_TU
    ;
#else
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t xw[TW], yw[TW];
  load_words8(xw, wire + i * 16);
  load_words8(yw, wire + i * 16 + 8);
  if (words8_ge_p(xw) || words8_ge_p(yw)) atomicOr(err, 1u);
  Fe<FT> x, y, t, kt, r2, k;
  fe_unpack<FT>(x, xw);
  fe_unpack<FT>(y, yw);
  TE_CONST(r2, R2);
  TE_CONST(k, K2DL);
  fe_mul<FT>(x, x, r2);
  fe_mul<FT>(y, y, r2);
  fe_mul<FT>(t, x, y);
  fe_mul<FT>(kt, t, k);
  if (check_curve) {
    // -x^2 + y^2 = 1 + d x^2 y^2   (src/bigint/twisted-edwards.ts:150-158 with Z = 1)
    Fe<FT> xx, yy, lhs, rhs, dd, one;
    fe_sqr<FT>(xx, x);
    fe_sqr<FT>(yy, y);
    fe_sub_2p<FT>(lhs, yy, xx);
    TE_CONST(dd, DL);
    fe_sqr<FT>(rhs, t);
    fe_mul<FT>(rhs, rhs, dd);
    fe_set_one<FT>(one);
    fe_add<FT>(rhs, rhs, one);
    fe_sub_4p<FT>(lhs, lhs, rhs);
    fe_cond_sub<FT, 4>(lhs);
    fe_reduce_4p<FT>(lhs);
    if (!fe_is_zero_canonical<FT>(lhs)) atomicOr(err, 2u);
  }
  fe_reduce_2p<FT>(x);
  fe_reduce_2p<FT>(y);
  fe_reduce_2p<FT>(t);
  fe_reduce_2p<FT>(kt);
  uint32_t* row = rows + i * TE_ROW_WORDS;
  fe_store<FT>(row, x);
  fe_store<FT>(row + 8, y);
  fe_store<FT>(row + 16, t);
  fe_store<FT>(row + 24, kt);
}
#endif

// ---------------------------------------------------------------------------------------------
// k_te_table_next: window tables of the Edwards path (see k_table_next, msm_kernels.h): row i of table k = 2^(c k) P_i as
// [x | y | xy | 2d xy]; c unified additions P + P (complete on this curve) and one inversion per point
// ---------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) k_te_table_next(uint32_t* rows_out, const uint32_t* rows_in, uint64_t n, int c)
#ifndef MSM_TE_TU
    ;
#else
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* row = rows_in + i * TE_ROW_WORDS;
  uint32_t w[TW];
  Ext P;
  load_words8(w, row);      fe_unpack<FT>(P.X, w);
  load_words8(w, row + 8);  fe_unpack<FT>(P.Y, w);
  load_words8(w, row + 16); fe_unpack<FT>(P.T, w);
  fe_set_one<FT>(P.Z);
#pragma unroll 1
  for (int j = 0; j < c; j++) {
    Ext Q = P;
    te_add(P, Q, Q);
  }
  Fe<FT> zi, x, y, t, kt, k;
  fe_reduce_4p<FT>(P.Z);
  fe_inv<FT>(zi, P.Z);   // Z != 0 for every point of a complete Edwards curve
  fe_mul<FT>(x, P.X, zi);
  fe_mul<FT>(y, P.Y, zi);
  fe_mul<FT>(t, x, y);
  TE_CONST(k, K2DL);
  fe_mul<FT>(kt, t, k);
  fe_reduce_2p<FT>(x);
  fe_reduce_2p<FT>(y);
  fe_reduce_2p<FT>(t);
  fe_reduce_2p<FT>(kt);
  uint32_t* out = rows_out + i * TE_ROW_WORDS;
  fe_store<FT>(out, x);
  fe_store<FT>(out + 8, y);
  fe_store<FT>(out + 16, t);
  fe_store<FT>(out + 24, kt);
}
#endif

// ---------------------------------------------------------------------------------------------
// k_te_digits: signed window digits of full-width scalars (no GLV), src/msm-basic.ts:72-91
// ---------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(1024) k_te_digits(uint32_t* dig, const uint32_t* scalars, uint32_t n, int c, int k_total, int k_lo,
                                                   int k_cnt, int strict, uint32_t* err, uint32_t pps, uint32_t* slice_hist,
                                                   uint32_t hb, uint64_t fbp, uint32_t b_lo, uint32_t b_n, uint32_t bt_lo, uint32_t bt_n)
#ifndef MSM_TE_TU
    ;
#else
{
  // slices and the fused histogram of the coarse bins: as k_digits (msm_kernels.h)
  extern __shared__ uint32_t lds_dig_hist[];
  uint32_t* lds_hist = slice_hist ? lds_dig_hist : nullptr;
  if (lds_hist) {
    for (uint32_t j = threadIdx.x; j < (uint32_t)k_cnt * hb; j += blockDim.x) lds_hist[j] = 0;
    __syncthreads();
  }
  const uint64_t p_end = min((uint64_t)(blockIdx.x + 1) * pps, (uint64_t)n);
  for (uint64_t i64 = (uint64_t)blockIdx.x * pps + threadIdx.x; i64 < p_end; i64 += blockDim.x) {
    const uint32_t i = (uint32_t)i64;
    uint32_t s[8];
    {
      const uint4* p4 = reinterpret_cast<const uint4*>(scalars + (uint64_t)i * 8);
      uint4 a = p4[0], b = p4[1];
      s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    }
    uint32_t q[8];
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = FRED_Q[j];
    if (words8_ge(s, q)) {   // scalars >= q are reduced (2^256 < 56 q), or refused under msm_opts.strict
      if (strict) atomicOr(err, 4u);
      for (int it = 0; it < 64 && words8_ge(s, q); it++) bn_addsub<8, 8>(s, q, true);
    }
    const uint32_t L = 1u << (c - 1);
    uint32_t carry = 0;
    for (int k = 0; k < k_total; k++) {
      uint32_t l = bn_take_bits<8>(s, c) + carry;
      if (l > L) { l = 2 * L - l; carry = 1; } else { carry = 0; }
      int kk = k - k_lo;
      if (kk >= 0 && kk < k_cnt) {
        uint32_t e = l, sgn = carry;
        if (e - 1 - (k == k_total - 1 ? bt_lo : b_lo) >= (k == k_total - 1 ? bt_n : b_n)) { e = 0; sgn = 0; }   // bucket-range shard: see k_digits
        dig[(uint64_t)kk * n + i] = e | (sgn << 31);
        digit_note(lds_hist, hb, fbp, kk, e);
      }
    }
  }
  if (lds_hist) {
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < (uint32_t)k_cnt * hb; j += blockDim.x) {
      const uint32_t kk = j / hb, h = j - kk * hb;
      slice_hist[((uint64_t)kk * gridDim.x + blockIdx.x) * hb + h] = lds_hist[j];
    }
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// k_te_add: one round of the bucket tree, output e = input 2e + input 2e+1 (unified addition)
// ---------------------------------------------------------------------------------------------

template <int MODE>
__global__ void __launch_bounds__(256) k_te_add(BatchArgs a) {
  const uint64_t T = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll 1
  for (uint64_t e = t; e < a.n_out; e += T) {
    Ext R;
    if (MODE == MODE_GATHER) {
      uint2 pp = reinterpret_cast<const uint2*>(a.slots)[e];
      AffRow P, Q;
      load_row(P, a.points, pp.x);
      load_row(Q, a.points, pp.y);
      te_add_rows(R, P, Q);
    } else {
      uint64_t ia, ib;
      bool has_b = true;
      if (MODE == MODE_REGULAR) {
        ia = 2 * e; ib = 2 * e + 1;
      } else {   // MODE_SEARCH: operand descriptors from k_tail_desc
        const uint32_t d = a.desc[e];
        ia = d >> 1; ib = ia + 1;
        has_b = (d & 1u) != 0;
      }
      Ext P, Q;
      load_ext(P, a.in, a.in_cap, ia);
      if (has_b) load_ext(Q, a.in, a.in_cap, ib); else te_set_identity(Q);
      te_add(R, P, Q);
    }
    store_ext(a.out, a.out_cap, e, R);
  }
}
// the three modes are compiled in te_kernels.hip
#ifdef MSM_TE_TU
template __global__ void k_te_add<MODE_GATHER>(BatchArgs);
template __global__ void k_te_add<MODE_REGULAR>(BatchArgs);
template __global__ void k_te_add<MODE_SEARCH>(BatchArgs);
#else
extern template __global__ void k_te_add<MODE_GATHER>(BatchArgs);
extern template __global__ void k_te_add<MODE_REGULAR>(BatchArgs);
extern template __global__ void k_te_add<MODE_SEARCH>(BatchArgs);
#endif


// ---------------------------------------------------------------------------------------------
// k_te_gen_points: synthetic inputs, P_i = sum_j T_j[idx_ij] over 5 basis tables (randomPointsFast,
// src/curve-random.ts:14-92, with the discrete logs known to the host; see msm_gen.h).  Writes wire format.
// ---------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) k_te_gen_points(uint32_t* wire_out, const uint32_t* tables, uint64_t n, uint64_t seed)
#ifndef MSM_TE_TU
    ;
#else
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ext acc;
  te_set_identity(acc);
#pragma unroll 1
  for (int j = 0; j < msm_gen::N_BASIS; j++) {
    const uint32_t t = msm_gen::table_index(seed, i, j);
    AffRow R;
    load_row(R, tables, (uint32_t)((j * msm_gen::TBL + t) << 1));
    Ext Q;
    Q.X = R.x; Q.Y = R.y; Q.T = R.t;
    fe_set_one<FT>(Q.Z);
    te_add(acc, acc, Q);
  }
  Fe<FT> zi, x, y, one;
  fe_inv<FT>(zi, acc.Z);
  fe_mul<FT>(x, acc.X, zi);
  fe_mul<FT>(y, acc.Y, zi);
  fe_set_zero<FT>(one);
  one.l[0] = 1;
  fe_mul<FT>(x, x, one);   // out of Montgomery form
  fe_mul<FT>(y, y, one);
  fe_reduce_4p<FT>(x);
  fe_reduce_4p<FT>(y);
  fe_store<FT>(wire_out + i * 16, x);
  fe_store<FT>(wire_out + i * 16 + 8, y);
}
#endif

// ---------------------------------------------------------------------------------------------
// k_te_bucket_reduce / k_te_window_sum: reduceBucketsChunk (src/msm-basic.ts:180-211) per chunk of TC
// buckets, then the per-window sum of the chunk columns
// ---------------------------------------------------------------------------------------------

MSM_DEV void ext_store_raw(uint32_t* dst, const Ext& P) {
#pragma unroll
  for (int l = 0; l < TL; l++) { dst[l] = P.X.l[l]; dst[TL + l] = P.Y.l[l]; dst[2 * TL + l] = P.Z.l[l]; dst[3 * TL + l] = P.T.l[l]; }
}
MSM_DEV void ext_load_raw(Ext& P, const uint32_t* src) {
#pragma unroll
  for (int l = 0; l < TL; l++) { P.X.l[l] = src[l]; P.Y.l[l] = src[TL + l]; P.Z.l[l] = src[2 * TL + l]; P.T.l[l] = src[3 * TL + l]; }
}

// k_te_bucket_finish: the unified addition needs no inversion, so the tree exists only for parallelism; once every
// bucket is down to a few elements one lane per bucket sums them in sequence (replaces ~5 latency-bound tail rounds).
// perm: buckets in descending order of remaining count (k_finish_perm), so a wave's lanes finish together.
__global__ void __launch_bounds__(256) k_te_bucket_finish(uint32_t* bucket_ext, const uint4* in, uint64_t in_cap, const uint32_t* off,
                                                          uint32_t nb, const uint32_t* perm)
#ifndef MSM_TE_TU
    ;
#else
{
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  if (perm) b = perm[b];
  const uint32_t o0 = off[b], o1 = off[b + 1];
  Ext acc;
  te_set_identity(acc);
#pragma unroll 1
  for (uint32_t o = o0; o < o1; o++) {
    Ext Q;
    load_ext(Q, in, in_cap, o);
    te_add(acc, acc, Q);
  }
  ext_store_raw(bucket_ext + (uint64_t)b * (4 * TL), acc);
}
#endif

// rows != nullptr: bit-sliced mode, as k_bucket_reduce (msm_kernels.h) -- the chunk's plain sum goes to rows, its local
// triangle to columns, both planar ([window][word][chunk]); the weight ch * TC is applied through per-bit sums (k_te_bit_tree)
__global__ void __launch_bounds__(64) k_te_bucket_reduce(uint32_t* columns, uint32_t* rows, const uint4* fin, uint64_t fin_cap,
                                                          const uint32_t* off_fin, const uint32_t* bucket_ext, uint32_t L, uint32_t TC,
                                                          uint32_t nchunks, uint32_t k_cnt)
#ifndef MSM_TE_TU
    ;
#else
{
  uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= nchunks * k_cnt) return;
  uint32_t kk = id / nchunks, ch = id - kk * nchunks;
  uint32_t lstart = ch * TC + 1;
  uint32_t lend = min(lstart + TC - 1, L);
  Ext row, tri;
  te_set_identity(row);
  te_set_identity(tri);
#pragma unroll 1
  for (uint32_t l = lend; l >= lstart; l--) {
    uint64_t b = (uint64_t)kk * L + (l - 1);
    if (bucket_ext) {   // bucket sums from k_te_bucket_finish (the identity for an empty bucket)
      Ext Q;
      ext_load_raw(Q, bucket_ext + b * (4 * TL));
      te_add(row, row, Q);
    } else {
      uint32_t o0 = off_fin[b], o1 = off_fin[b + 1];
      if (o1 > o0) {
        Ext Q;
        load_ext(Q, fin, fin_cap, o0);
        te_add(row, row, Q);
      }
    }
    te_add(tri, tri, row);
  }
  if (rows) {
    uint32_t* rp = rows + (uint64_t)kk * (4 * TL) * nchunks;
    uint32_t* cp = columns + (uint64_t)kk * (4 * TL) * nchunks;
#pragma unroll
    for (int l = 0; l < TL; l++) {
      rp[(uint64_t)l * nchunks + ch] = row.X.l[l]; rp[(uint64_t)(TL + l) * nchunks + ch] = row.Y.l[l];
      rp[(uint64_t)(2 * TL + l) * nchunks + ch] = row.Z.l[l]; rp[(uint64_t)(3 * TL + l) * nchunks + ch] = row.T.l[l];
      cp[(uint64_t)l * nchunks + ch] = tri.X.l[l]; cp[(uint64_t)(TL + l) * nchunks + ch] = tri.Y.l[l];
      cp[(uint64_t)(2 * TL + l) * nchunks + ch] = tri.Z.l[l]; cp[(uint64_t)(3 * TL + l) * nchunks + ch] = tri.T.l[l];
    }
    return;
  }
  uint32_t ls = lstart - 1;
  if (ls) {
#pragma unroll 1
    while (true) {
      if (ls & 1) te_add(tri, tri, row);
      ls >>= 1;
      if (ls == 0) break;
      te_add(row, row, row);   // doubling = unified add, src/curve-twisted-edwards.ts:215-217
    }
  }
  ext_store_raw(columns + (uint64_t)id * (4 * TL), tri);
}
#endif

// the bit tree of msm_kernels.h over extended Edwards points (X, Y, Z, T: 4 x 9 limbs; packed sums: 4 x 8 words)
struct TePT {
  using P = Ext;
  static constexpr int W = 4 * TL;
  static constexpr int PW = 32;
  static MSM_DEV void zero(P& p) { te_set_identity(p); }
  static MSM_DEV void add(P& r, const P& a, const P& b) { te_add(r, a, b); }
  static MSM_DEV uint32_t& word(P& p, int w) { return w < TL ? p.X.l[w] : w < 2 * TL ? p.Y.l[w - TL] : w < 3 * TL ? p.Z.l[w - 2 * TL] : p.T.l[w - 3 * TL]; }
  static MSM_DEV void pack(uint32_t* dst, P& acc) {
    uint32_t w[TW];
    fe_pack<FT>(w, acc.X);
#pragma unroll
    for (int j = 0; j < TW; j++) dst[j] = w[j];
    fe_pack<FT>(w, acc.Y);
#pragma unroll
    for (int j = 0; j < TW; j++) dst[8 + j] = w[j];
    fe_pack<FT>(w, acc.Z);
#pragma unroll
    for (int j = 0; j < TW; j++) dst[16 + j] = w[j];
    fe_pack<FT>(w, acc.T);
#pragma unroll
    for (int j = 0; j < TW; j++) dst[24 + j] = w[j];
  }
};

__global__ void __launch_bounds__(BT_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_te_bit_tree(uint32_t* out, const uint32_t* rows,
                                                         const uint32_t* tris, uint32_t n_in, uint32_t nbits, int masked,
                                                         int pack_out, uint32_t nblk_out)
#ifndef MSM_TE_TU
    ;
#else
{
  __shared__ uint32_t lds[TePT::W * BT_THREADS];
  bit_tree_body<TePT>(out, rows, tris, n_in, nbits, masked, pack_out, nblk_out, lds);
}
#endif

constexpr int TE_WS_THREADS = 256;

// output: 4 x 8 packed words (X, Y, Z, T), Montgomery form, values < 2p
__global__ void __launch_bounds__(TE_WS_THREADS) k_te_window_sum(uint32_t* partials, const uint32_t* columns, uint32_t nchunks)
#ifndef MSM_TE_TU
    ;
#else
{
  __shared__ uint32_t lds[4 * TL * TE_WS_THREADS];
  const uint32_t kk = blockIdx.x, tid = threadIdx.x;
  Ext acc;
  te_set_identity(acc);
#pragma unroll 1
  for (uint32_t j = tid; j < nchunks; j += TE_WS_THREADS) {
    Ext Q;
    ext_load_raw(Q, columns + ((uint64_t)kk * nchunks + j) * (4 * TL));
    te_add(acc, acc, Q);
  }
#pragma unroll 1
  for (uint32_t s = TE_WS_THREADS / 2; s >= 1; s >>= 1) {
    if (tid >= s && tid < 2 * s) {
#pragma unroll
      for (int l = 0; l < TL; l++) {
        lds[(l)*TE_WS_THREADS + tid] = acc.X.l[l];
        lds[(TL + l) * TE_WS_THREADS + tid] = acc.Y.l[l];
        lds[(2 * TL + l) * TE_WS_THREADS + tid] = acc.Z.l[l];
        lds[(3 * TL + l) * TE_WS_THREADS + tid] = acc.T.l[l];
      }
    }
    __syncthreads();
    if (tid < s) {
      Ext Q;
#pragma unroll
      for (int l = 0; l < TL; l++) {
        Q.X.l[l] = lds[(l)*TE_WS_THREADS + tid + s];
        Q.Y.l[l] = lds[(TL + l) * TE_WS_THREADS + tid + s];
        Q.Z.l[l] = lds[(2 * TL + l) * TE_WS_THREADS + tid + s];
        Q.T.l[l] = lds[(3 * TL + l) * TE_WS_THREADS + tid + s];
      }
      te_add(acc, acc, Q);
    }
    __syncthreads();
  }
  if (tid == 0) {
    uint32_t* dst = partials + (uint64_t)kk * 32;
    uint32_t w[TW];
    fe_pack<FT>(w, acc.X);
#pragma unroll
    for (int j = 0; j < TW; j++) dst[j] = w[j];
    fe_pack<FT>(w, acc.Y);
#pragma unroll
    for (int j = 0; j < TW; j++) dst[8 + j] = w[j];
    fe_pack<FT>(w, acc.Z);
#pragma unroll
    for (int j = 0; j < TW; j++) dst[16 + j] = w[j];
    fe_pack<FT>(w, acc.T);
#pragma unroll
    for (int j = 0; j < TW; j++) dst[24 + j] = w[j];
  }
}
#endif

// element-wise base-field operators for parity tests (same op codes as k_test_fp)
__global__ void __launch_bounds__(256) k_te_test_fp(uint32_t* out, const uint32_t* a, const uint32_t* b, uint32_t n, int op)
#ifndef MSM_TE_TU
    ;
#else
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<FT> x, y, r;
  fe_load<FT>(x, a + (uint64_t)i * TW);
  fe_load<FT>(y, b + (uint64_t)i * TW);
  switch (op) {
    case OP_MUL: fe_mul<FT>(r, x, y); break;
    case OP_SQR: fe_sqr<FT>(r, x); break;
    case OP_ADD: fe_add<FT>(r, x, y); break;
    case OP_SUB: fe_sub_p<FT>(r, x, y); break;
    case OP_INV: fe_inv<FT>(r, x); break;
    case OP_INV_FERMAT: fe_inv_fermat<FT>(r, x); break;
    case OP_INV_KALISKI: fe_inv_kaliski<FT>(r, x); break;
    case OP_INV_WORDSLICED: fe_inv_wordsliced<FT>(r, x); break;
    case OP_TO_MONT: {
      Fe<FT> r2;
      TE_CONST(r2, R2);
      fe_mul<FT>(r, x, r2);
      break;
    }
    default: {
      Fe<FT> one;
      fe_set_zero<FT>(one);
      one.l[0] = 1;
      fe_mul<FT>(r, x, one);
      break;
    }
  }
  fe_reduce_4p<FT>(r);
  fe_store<FT>(out + (uint64_t)i * TW, r);
}
#endif

// raw-limb multiplier test on the 253-bit field (see k_test_fp_raw)
__global__ void __launch_bounds__(256) k_te_test_fp_raw(uint32_t* out, const uint32_t* a, const uint32_t* b, uint32_t n, int op)
#ifndef MSM_TE_TU
    ;
#else
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<FT> x, y, r;
#pragma unroll
  for (int l = 0; l < TL; l++) { x.l[l] = a[(uint64_t)i * TL + l]; y.l[l] = b[(uint64_t)i * TL + l]; }
  if (op == OP_SQR) fe_sqr<FT>(r, x);
  else fe_mul<FT>(r, x, y);
#pragma unroll
  for (int l = 0; l < TL; l++) out[(uint64_t)i * TL + l] = r.l[l];
}
#endif

// general unified addition (te_add, 9M) on extended points (X, Y, Z, T: 4 x 8 canonical plain-integer words each):
// the operator of src/curve-twisted-edwards.test.ts:55-158; op 1 = doubling through the same formula (P + P)
__global__ void __launch_bounds__(64) k_te_test_curve_op(uint32_t* out, const uint32_t* pp, const uint32_t* qq, uint32_t n, int op)
#ifndef MSM_TE_TU
    ;
#else
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<FT> r2, one;
  TE_CONST(r2, R2);
  fe_set_zero<FT>(one);
  one.l[0] = 1;
  Ext P, Q, R;
  Fe<FT>* co[8] = {&P.X, &P.Y, &P.Z, &P.T, &Q.X, &Q.Y, &Q.Z, &Q.T};
  for (int j = 0; j < 8; j++) {
    fe_load<FT>(*co[j], (j < 4 ? pp : qq) + (uint64_t)i * 32 + (j % 4) * 8);
    fe_mul<FT>(*co[j], *co[j], r2);
  }
  if (op == 1) te_add(R, P, P);
  else te_add(R, P, Q);
  Fe<FT>* ro[4] = {&R.X, &R.Y, &R.Z, &R.T};
  for (int j = 0; j < 4; j++) {
    fe_mul<FT>(*ro[j], *ro[j], one);
    fe_reduce_4p<FT>(*ro[j]);
    fe_store<FT>(out + (uint64_t)i * 32 + j * 8, *ro[j]);
  }
}
#endif

}  // namespace te
}  // namespace msm
