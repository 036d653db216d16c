// HIP kernels of the batched-affine Pippenger MSM for short Weierstrass curves y^2 = x^3 + b with a
// GLV endomorphism (BLS12-377 G1, BLS12-381 G1) on gfx950.
//
// Phase map (reference: src/msm-batched-affine.ts:69-340):
//   k_points_from_wire   pointsFromBytes + toMontgomery + endomorphism      src/parallel.ts:97-116, src/wasm/curve.ts:90-103
//   k_digits             decompose + signed slices + bucket histogram       :350-421 (scalars), :175-203
//   k_scan               integrateBucketCounts                              :423-447
//   k_scatter            sortPoints (indices, not 116-byte points)          :456-502
//   k_batch_add          bucket accumulation tree, one inversion per lane   :243-282, src/curve-affine.ts:376-522
//   k_bucket_reduce      normalizeBucketsStorage + reduceBucketsColumnProjective :504-583
//   k_window_sum         partition sums                                     :312-319
// The Horner combination of the K partition sums (:322-333) runs on the host (msm_api.hip).
//
// Data layout in HBM
//   point rows     : N x 64 words (256 B): two 128-byte lines [x | y | pad] and [beta*x | y | pad], canonical
//                    Montgomery (R = 2^390).  A gather of either GLV half touches exactly one 128-byte
//                    line (x alone = its first 64-byte sector): random 64-byte sectors are what bounds
//                    round 1, so no window may straddle sectors it does not need.
//   digits         : Kg x 2N words, magnitude | sign << 31, entry j = 2*point + half
//   slots          : bucket-sorted entry payloads ((j << 1) | neg), every bucket padded with
//                    SLOT_EMPTY to a multiple of G = 2^g so that g tree rounds need no index math
//   tree buffers   : "plane" layout, 16-byte piece c of element e at uint4 index c*cap + e
//                    (planes 0-2 = x, 3-5 = y): consecutive lanes read consecutive 16-byte pieces
//   prefix scratch : dword planes [(step*13 + limb)*T + thread]
// The identity is encoded with x = all-ones words (never canonical), y = 0.
#pragma once
#include "curve.h"
#include "glv.h"
#include "packed.h"

namespace msm {

// curve configurations: every curve-dependent kernel is a template over one of these and takes its limb and word counts from
// the field (the reference sizes limbs per field too, src/parallel.ts:53-57, src/field-msm.ts:20-56): 13 x 30-bit limbs in
// registers and 12 x 32-bit words (3 pieces of 16 bytes) in memory for the 377- / 381-bit primes, 9 limbs and 8 words (2 pieces)
// for Pallas.  Point rows are 256 bytes for all of them (two 128-byte lines), the curve-independent sort / scan kernels are shared.
struct CvBls377 { using F = Fp377; using G = GlvBls377; };   // src/concrete/bls12-377.params.ts
struct CvBls381 { using F = Fp381; using G = GlvBls381; };   // src/concrete/bls12-381.params.ts
struct CvPallas { using F = FpPallas; using G = GlvPallas; }; // src/concrete/pasta.params.ts (255-bit p)
constexpr int NL = 13;   // the wide layout; kernels use F::NL / F::NW
constexpr int NW = 12;
static_assert(Fp377::NL == NL && Fp377::NW == NW && Fp381::NL == NL && Fp381::NW == NW, "the 12-word layout");
constexpr int PART_WORDS = 36;   // a window sum on its way to the host: X, Y, Z at word 0, 12, 24 (8-word fields: upper words zero)
constexpr int ROW_WORDS = 64;   // 256 B per point
constexpr int ROW_HALF = 32;    // word offset of the endomorphism image (second 128-byte line)
constexpr int ROW_Y = 12;       // word offset of y inside a line of the 12-word fields (in general: F::NW)
constexpr uint32_t SLOT_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t INF_WORD = 0xFFFFFFFFu;

// How window kk's bucket index (l - 1) is cut for the LDS-staged passes of the sort (sort_kernels.h):
// | ab coarse bits | mb mid bits | fb fine bits |.  Two-pass radix split (c <= 16): mb = 0, fb = 7.  Bin split (c > 16): mb = 0,
// the cut is made on the window's EFFECTIVE bits (a short top window fills only the low end of its bucket range).
struct WinSplit {
  uint8_t ab[16], mb[16], fb[16];
};

}  // namespace msm
#include "batch_add.h"   // k_batch_add: the accumulation tree round
namespace msm {

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------

template <int W>
MSM_DEV void load_words12(uint32_t (&w)[W], const uint32_t* p) {
  const uint4* p4 = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int j = 0; j < W / 4; j++) {
    uint4 v = p4[j];
    w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
  }
}

// one coordinate (W / 4 pieces) of element e; x starts at plane 0, y at plane W / 4
template <int W>
MSM_DEV void load_planes3(uint32_t (&w)[W], const uint4* base, uint64_t cap, int first_plane, uint64_t e) {
#pragma unroll
  for (int j = 0; j < W / 4; j++) {
    uint4 v = base[(uint64_t)(first_plane + j) * cap + e];
    w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
  }
}

template <int W>
MSM_DEV void store_planes3(uint4* base, uint64_t cap, int first_plane, uint64_t e, const uint32_t (&w)[W]) {
#pragma unroll
  for (int j = 0; j < W / 4; j++)
    base[(uint64_t)(first_plane + j) * cap + e] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}

template <class F>
MSM_DEV bool words_ge_p(const uint32_t (&w)[F::NW]) {  // w >= p ?
  constexpr int NW = F::NW;
  bool gt = false, lt = false;
#pragma unroll
  for (int j = NW - 1; j >= 0; j--) {
    if (!gt && !lt) {
      if (w[j] > F::PW[j]) gt = true;
      else if (w[j] < F::PW[j]) lt = true;
    }
  }
  return !lt;
}

template <class F>
MSM_DEV bool fe_equal(const Fe<F>& a, const Fe<F>& b) {
  constexpr int NL = F::NL;
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) o |= a.l[i] ^ b.l[i];
  return o == 0;
}

template <class F>
MSM_DEV void store_row(uint32_t* row, const Fe<F>& x, const Fe<F>& y, const Fe<F>& bx) {   // y follows x: word F::NW of a line
  fe_store<F>(row, x);
  fe_store<F>(row + F::NW, y);
  fe_store<F>(row + ROW_HALF, bx);
  fe_store<F>(row + ROW_HALF + F::NW, y);
}

template <int NP>   // NP = pieces per coordinate
MSM_DEV void store_row_identity(uint32_t* row) {
  uint4* r4 = reinterpret_cast<uint4*>(row);
  const uint4 ones = make_uint4(INF_WORD, INF_WORD, INF_WORD, INF_WORD), zeros = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int j = 0; j < NP; j++) {
      r4[h * (ROW_HALF / 4) + j] = ones;
      r4[h * (ROW_HALF / 4) + NP + j] = zeros;
    }
}

// ---------------------------------------------------------------------------------------------
// k_points_from_wire: N x (x || y), 48-byte little-endian canonical integers -> point rows
// ---------------------------------------------------------------------------------------------

template <class CV>
__global__ void __launch_bounds__(256) k_points_from_wire(uint32_t* rows, const uint32_t* wire, uint64_t n,
                                                          int check_curve, uint32_t* err) {
  using F = typename CV::F;
  constexpr int NL = F::NL, NW = F::NW;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t xw[NW], yw[NW];
  load_words12(xw, wire + i * (2 * NW));
  load_words12(yw, wire + i * (2 * NW) + NW);
  uint32_t* row = rows + i * ROW_WORDS;
  uint32_t any = 0;
#pragma unroll
  for (int j = 0; j < NW; j++) any |= xw[j] | yw[j];
  if (any == 0) {  // (0, 0) is not on y^2 = x^3 + 1: used as the wire encoding of the identity
    store_row_identity<NW / 4>(row);
    return;
  }
  if (words_ge_p<F>(xw) || words_ge_p<F>(yw)) atomicOr(err, 1u);
  Fe<F> x, y, r2, beta, bx;
  fe_unpack<F>(x, xw);
  fe_unpack<F>(y, yw);
#pragma unroll
  for (int l = 0; l < NL; l++) { r2.l[l] = F::R2[l]; beta.l[l] = F::BETAL[l]; }
  fe_mul<F>(x, x, r2);
  fe_reduce_2p<F>(x);
  fe_mul<F>(y, y, r2);
  fe_reduce_2p<F>(y);
  fe_mul<F>(bx, x, beta);
  fe_reduce_2p<F>(bx);
  if (check_curve) {
    Fe<F> lhs, rhs, bb;
#pragma unroll
    for (int l = 0; l < NL; l++) bb.l[l] = F::BL[l];
    fe_sqr<F>(lhs, y);
    fe_sqr<F>(rhs, x);
    fe_mul<F>(rhs, rhs, x);
    fe_add<F>(rhs, rhs, bb);          // < 3p
    fe_sub_4p<F>(lhs, lhs, rhs);      // < 6p: reduce by 4p first
    fe_cond_sub<F, 4>(lhs);
    if (!fe_is_zero_mod_p<F>(lhs)) atomicOr(err, 2u);
  }
  store_row(row, x, y, bx);
}

// ---------------------------------------------------------------------------------------------
// k_table_next: window tables.  Row i of table k holds 2^(c k) P_i (with its beta x line), so that the digit of window k
// addresses a point that already carries the window's weight: all K windows of an MSM then share ONE set of buckets
// (sum_k 2^(c k) sum_i d_ik P_i = sum_(i,k) d_ik T_k[i]) -- K times fewer buckets to finish and reduce, no Horner step.
// The reference keeps no such tables (its memory is 4 GiB of wasm heap); 288 GB of HBM hold six of them for 2^26 points.
// One launch makes table k from table k - 1: c projective doublings (dbl-1998-cmo-2, src/curve-projective.ts:202-253) and
// one inversion per point.  Built once per point set and plan, like the beta x of k_points_from_wire.
// ---------------------------------------------------------------------------------------------

template <class CV>
__global__ void __launch_bounds__(256) k_table_next(uint32_t* rows_out, const uint32_t* rows_in, uint64_t n, int c) {
  using F = typename CV::F;
  constexpr int NL = F::NL, NW = F::NW;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t xw[NW], yw[NW];
  load_words12(xw, rows_in + i * ROW_WORDS);
  load_words12(yw, rows_in + i * ROW_WORDS + NW);
  uint32_t* out = rows_out + i * ROW_WORDS;
  if (xw[NW - 1] == INF_WORD) {
    store_row_identity<NW / 4>(out);
    return;
  }
  Proj<F> P;
  fe_unpack<F>(P.X, xw);
  fe_unpack<F>(P.Y, yw);
  fe_set_one<F>(P.Z);
#pragma unroll 1
  for (int j = 0; j < c; j++) proj_double<F>(P, P);
  if (proj_is_zero<F>(P)) {   // a point of even order (outside the prime-order subgroup) doubled to the identity
    store_row_identity<NW / 4>(out);
    return;
  }
  Fe<F> zi, x, y, bx, beta;
  fe_reduce_4p<F>(P.Z);
  fe_inv<F>(zi, P.Z);
  fe_mul<F>(x, P.X, zi);
  fe_reduce_2p<F>(x);
  fe_mul<F>(y, P.Y, zi);
  fe_reduce_2p<F>(y);
#pragma unroll
  for (int l = 0; l < NL; l++) beta.l[l] = F::BETAL[l];
  fe_mul<F>(bx, x, beta);
  fe_reduce_2p<F>(bx);
  store_row(out, x, y, bx);
}

// ---------------------------------------------------------------------------------------------
// k_digits: scalars (N x 32 B LE) -> signed window digits of both GLV halves
// ---------------------------------------------------------------------------------------------

MSM_DEV bool words8_ge(const uint32_t (&a)[8], const uint32_t* q) {
  bool gt = false, lt = false;
#pragma unroll
  for (int j = 7; j >= 0; j--) {
    if (!gt && !lt) {
      if (a[j] > q[j]) gt = true;
      else if (a[j] < q[j]) lt = true;
    }
  }
  return !lt;
}

// windows [k_lo, k_lo + k_cnt) of K_total are emitted (window groups / multi-GPU window shards)
// glv: bit 0 = endomorphism split, bit 1 = folded top window (Plan::fold, msm_api.hip): window K_total - 1 is c + 1 bits wide
// and keeps its value as it is -- at most 2^c, it cannot carry out.
// Slices: block b owns the points [b * pps, (b + 1) * pps).  slice_hist != nullptr (the bin split of big windows): the block also
// counts, per window, the coarse bins ((l - 1) >> ws.fb[kk]) of the digits it writes -- in the LDS, `hb` counters per window -- and
// leaves them in slice_hist[(kk * gridDim.x + b) * hb + bin]: the digits are in registers here, a separate histogram pass would
// read all of them again.
// b_lo, b_n: the bucket-range shard of a multi-GPU run -- only digits whose bucket index l - 1 lies in [b_lo, b_lo + b_n) are
// kept, the others become "no entry" (0) and are never sorted (the whole range: 0, 0xFFFFFFFF); bt_lo, bt_n: the same for the
// TOP window, whose digits cover another range than the recoded ones (twice as many buckets when folded, fewer when short).
// fbp: the fine bits of the group's windows, four bits each (WinSplit::fb packed: a register shift instead of a load from the
// kernel arguments per digit)
MSM_DEV void digit_note(uint32_t* lds_hist, uint32_t hb, uint64_t fbp, int kk, uint32_t l) {
  if (lds_hist && l) atomicAdd(&lds_hist[(uint32_t)kk * hb + ((l - 1) >> ((uint32_t)(fbp >> (4 * kk)) & 15u))], 1u);
}
inline uint64_t pack_fine_bits(const WinSplit& ws) {
  uint64_t v = 0;
  for (int kk = 0; kk < 16; kk++) v |= (uint64_t)(ws.fb[kk] & 15u) << (4 * kk);
  return v;
}

template <class CV>
__global__ void __launch_bounds__(1024) k_digits(uint32_t* dig, const uint32_t* scalars, uint32_t n, int c, int k_total,
                                                int k_lo, int k_cnt, int glv_flags, int strict, uint32_t* err, uint32_t pps,
                                                uint32_t* slice_hist, uint32_t hb, uint64_t fbp, uint32_t b_lo, uint32_t b_n, uint32_t bt_lo,
                                                uint32_t bt_n) {
  extern __shared__ uint32_t lds_dig_hist[];
  uint32_t* lds_hist = slice_hist ? lds_dig_hist : nullptr;
  if (lds_hist) {
    for (uint32_t j = threadIdx.x; j < (uint32_t)k_cnt * hb; j += blockDim.x) lds_hist[j] = 0;
    __syncthreads();
  }
  const int glv = glv_flags & 1;
  const bool fold = glv_flags & 2;
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
  for (int j = 0; j < 8; j++) q[j] = CV::G::Q[j];
  // inputs are specified < q (src/curve-random.ts:151-194).  Larger values are reduced mod q -- the group element is the
  // same -- unless the caller asked for strict checking (msm_opts.strict), in which case the call fails with MSM_ERR_SCALAR
  if (words8_ge(s, q)) {
    if (strict) atomicOr(err, 4u);
    for (int it = 0; it < 16 && words8_ge(s, q); it++) bn_addsub<8, 8>(s, q, true);
  }

  const uint32_t L = 1u << (c - 1);
  const uint64_t two_n = 2ull * n;
  if (!glv) {
    // msmBasic / msmProjective window structure (src/msm-basic.ts:72-91): signed digits of the whole scalar on the
    // first entry of the point; the endomorphism entry stays empty (digit 0 is never sorted)
    uint32_t carry = 0;
    for (int k = 0; k < k_total; k++) {
      const bool top = fold && k == k_total - 1;
      uint32_t l = bn_take_bits<8>(s, top ? c + 1 : c) + carry;
      if (!top && l > L) { l = 2 * L - l; carry = 1; } else { carry = 0; }
      if (top && l > 2 * L) { atomicOr(err, 8u); l = 2 * L; }   // (see below: a folded top window must stay within its buckets)
      uint32_t sgn = carry;
      if (l - 1 - (k == k_total - 1 ? bt_lo : b_lo) >= (k == k_total - 1 ? bt_n : b_n)) { l = 0; sgn = 0; }
      int kk = k - k_lo;
      if (kk >= 0 && kk < k_cnt) {
        *reinterpret_cast<uint2*>(dig + (uint64_t)kk * two_n + 2ull * i) = make_uint2(l | (sgn << 31), 0u);
        digit_note(lds_hist, hb, fbp, kk, l);
      }
    }
    continue;
  }
  GlvHalf h[2];
  glv_decompose<typename CV::G>(h[0], h[1], s);
  // both halves of a window leave as ONE 8-byte store (entries 2 i and 2 i + 1 are neighbours): 4-byte stores a dword apart
  // ran the kernel at half the streaming rate
  uint32_t carry0 = 0, carry1 = 0;
  for (int k = 0; k < k_total; k++) {
    const bool top = fold && k == k_total - 1;
    uint32_t l0 = bn_take_bits<4>(h[0].mag, top ? c + 1 : c) + carry0;
    uint32_t l1 = bn_take_bits<4>(h[1].mag, top ? c + 1 : c) + carry1;
    if (!top && l0 > L) { l0 = 2 * L - l0; carry0 = 1; } else { carry0 = 0; }
    if (!top && l1 > L) { l1 = 2 * L - l1; carry1 = 1; } else { carry1 = 0; }
    // the folded top window is not recoded: its value stays within its 2^c buckets only while the GLV halves stay below
    // 2^glv_max_bits -- a half that ever exceeded the bound would index past the window's buckets, so it is flagged (the call
    // fails with MSM_ERR_INTERNAL) and clamped instead
    if (top && (l0 > 2 * L || l1 > 2 * L)) { atomicOr(err, 8u); l0 = min(l0, 2 * L); l1 = min(l1, 2 * L); }
    const int kk = k - k_lo;
    if (kk >= 0 && kk < k_cnt) {
      uint32_t neg0 = carry0 ^ (h[0].neg ? 1u : 0u), neg1 = carry1 ^ (h[1].neg ? 1u : 0u);
      uint32_t e0 = l0, e1 = l1;
      const uint32_t f_lo = k == k_total - 1 ? bt_lo : b_lo, f_n = k == k_total - 1 ? bt_n : b_n;
      if (e0 - 1 - f_lo >= f_n) { e0 = 0; neg0 = 0; }
      if (e1 - 1 - f_lo >= f_n) { e1 = 0; neg1 = 0; }
      *reinterpret_cast<uint2*>(dig + (uint64_t)kk * two_n + 2ull * i) = make_uint2(e0 | (neg0 << 31), e1 | (neg1 << 31));
      digit_note(lds_hist, hb, fbp, kk, e0);
      digit_note(lds_hist, hb, fbp, kk, e1);
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

// ---------------------------------------------------------------------------------------------
// projective points in raw limb form (3 x 13 words) between the reduction kernels
// ---------------------------------------------------------------------------------------------

template <class F>
MSM_DEV void proj_store(uint32_t* dst, const Proj<F>& P) {
  constexpr int NL = F::NL;
#pragma unroll
  for (int l = 0; l < NL; l++) { dst[l] = P.X.l[l]; dst[NL + l] = P.Y.l[l]; dst[2 * NL + l] = P.Z.l[l]; }
}
template <class F>
MSM_DEV void proj_load(Proj<F>& P, const uint32_t* src) {
  constexpr int NL = F::NL;
#pragma unroll
  for (int l = 0; l < NL; l++) { P.X.l[l] = src[l]; P.Y.l[l] = src[NL + l]; P.Z.l[l] = src[2 * NL + l]; }
}

// "planar" form for arrays that consecutive lanes walk together: word w of element j at base[w * stride + j]
template <class F>
MSM_DEV void proj_store_planar(uint32_t* base, uint64_t stride, uint64_t j, const Proj<F>& P) {
  constexpr int NL = F::NL;
#pragma unroll
  for (int l = 0; l < NL; l++) {
    base[(uint64_t)l * stride + j] = P.X.l[l];
    base[(uint64_t)(NL + l) * stride + j] = P.Y.l[l];
    base[(uint64_t)(2 * NL + l) * stride + j] = P.Z.l[l];
  }
}
template <class F>
MSM_DEV void proj_load_planar(Proj<F>& P, const uint32_t* base, uint64_t stride, uint64_t j) {
  constexpr int NL = F::NL;
#pragma unroll
  for (int l = 0; l < NL; l++) {
    P.X.l[l] = base[(uint64_t)l * stride + j];
    P.Y.l[l] = base[(uint64_t)(NL + l) * stride + j];
    P.Z.l[l] = base[(uint64_t)(2 * NL + l) * stride + j];
  }
}

// ---------------------------------------------------------------------------------------------
// k_bucket_finish: once every bucket is down to a handful of elements the remaining tree rounds are pure
// latency (one shared inversion per launch for a few additions per lane).  This kernel ends the accumulation
// in one launch instead: one lane per bucket sums its remaining affine elements with mixed projective
// additions -- no inversion at all -- and leaves the bucket sum in projective form for k_bucket_reduce.
// ---------------------------------------------------------------------------------------------

template <class CV>
__global__ void __launch_bounds__(256) k_bucket_finish(uint32_t* bucket_proj, const uint4* in, uint64_t in_cap,
                                                       const uint32_t* off, uint32_t nb, const uint32_t* perm) {
  using F = typename CV::F;
  constexpr int NL = F::NL, NW = F::NW, NP = F::NW / 4;
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  if (perm) b = perm[b];   // buckets ordered by remaining count (k_finish_perm): equal trip counts inside a wave
  const uint32_t o0 = off[b], o1 = off[b + 1];
  Proj<F> acc;
  proj_set_zero<F>(acc);
  // the next element's words are in flight while the current one is added (lanes read 16-byte pieces a bucket
  // length apart: nothing coalesces, so the latency has to be covered by arithmetic)
  uint32_t nx[NW], ny[NW];
  if (o0 < o1) {
    load_planes3(nx, in, in_cap, 0, o0);
    load_planes3(ny, in, in_cap, NP, o0);
  }
#pragma unroll 1
  for (uint32_t o = o0; o < o1; o++) {
    Proj<F> Q;
    const bool qinf = nx[NW - 1] == INF_WORD;
    fe_unpack<F>(Q.X, nx);
    fe_unpack<F>(Q.Y, ny);
    if (o + 1 < o1) {
      load_planes3(nx, in, in_cap, 0, o + 1);
      load_planes3(ny, in, in_cap, NP, o + 1);
    }
    proj_add_mixed<F>(acc, acc, Q, qinf);
  }
  proj_store(bucket_proj + (uint64_t)b * (3 * NL), acc);
}

// rows != nullptr: "bit-sliced" mode -- the chunk's plain sum is written to rows[id] and its local triangle to
// columns[id]; the weight (lstart - 1) = ch * TC is applied later through per-bit sums (k_bit_tree) instead of a
// double-and-add chain in every lane.
// bucket_proj != nullptr: bucket sums come from k_bucket_finish (projective, one per bucket) instead of the tree buffer
template <class CV>
__global__ void __launch_bounds__(64) k_bucket_reduce(uint32_t* columns, uint32_t* rows, const uint4* fin, uint64_t fin_cap,
                                                       const uint32_t* off_fin, const uint32_t* bucket_proj, uint32_t L,
                                                       uint32_t TC, uint32_t nchunks, uint32_t k_cnt) {
  using F = typename CV::F;
  constexpr int NL = F::NL, NW = F::NW, NP = F::NW / 4;
  uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= nchunks * k_cnt) return;
  uint32_t kk = id / nchunks, ch = id - kk * nchunks;
  uint32_t lstart = ch * TC + 1;                       // bucket indices l are 1-based
  uint32_t lend = min(lstart + TC - 1, L);
  Proj<F> row, tri;
  proj_set_zero<F>(row);
  proj_set_zero<F>(tri);
#pragma unroll 1
  for (uint32_t l = lend; l >= lstart; l--) {
    uint64_t b = (uint64_t)kk * L + (l - 1);
    Proj<F> Q;
    if (bucket_proj) {
      proj_load(Q, bucket_proj + b * (3 * NL));
      proj_add<F>(row, row, Q);
    } else {
      // every element the tree left in this bucket goes straight into the running row sum (mixed additions): no separate
      // pass that first sums each bucket on its own, and one full addition per bucket less
      const uint32_t o0 = off_fin[b], o1 = off_fin[b + 1];
      uint32_t nx[NW], ny[NW];
      if (o0 < o1) {
        load_planes3(nx, fin, fin_cap, 0, o0);
        load_planes3(ny, fin, fin_cap, NP, o0);
      }
#pragma unroll 1
      for (uint32_t o = o0; o < o1; o++) {
        const bool qinf = nx[NW - 1] == INF_WORD;
        fe_unpack<F>(Q.X, nx);
        fe_unpack<F>(Q.Y, ny);
        if (o + 1 < o1) {   // the next element's words are in flight during the addition
          load_planes3(nx, fin, fin_cap, 0, o + 1);
          load_planes3(ny, fin, fin_cap, NP, o + 1);
        }
        proj_add_mixed<F>(row, row, Q, qinf);
      }
    }
    proj_add<F>(tri, tri, row);
  }
  if (rows) {   // per window a [39][nchunks] matrix each: lanes (= consecutive chunks) write and k_bit_tree reads coalesced
    proj_store_planar(rows + (uint64_t)kk * (3 * NL) * nchunks, nchunks, ch, row);
    proj_store_planar(columns + (uint64_t)kk * (3 * NL) * nchunks, nchunks, ch, tri);
    return;
  }
  uint32_t ls = lstart - 1;
  if (ls) {
#pragma unroll 1
    while (true) {
      if (ls & 1) proj_add<F>(tri, tri, row);
      ls >>= 1;
      if (ls == 0) break;
      proj_double<F>(row, row);
    }
  }
  proj_store(columns + (uint64_t)id * (3 * NL), tri);
}

// ---------------------------------------------------------------------------------------------
// k_window_sum: P_k = sum of the window's columns; one workgroup per window
// output: 3 x 12 packed canonical Montgomery words (X, Y, Z) per window
// ---------------------------------------------------------------------------------------------

constexpr int WS_THREADS = 256;

template <class CV>
__global__ void __launch_bounds__(WS_THREADS) k_window_sum(uint32_t* partials, const uint32_t* columns, uint32_t nchunks) {
  using F = typename CV::F;
  constexpr int NL = F::NL, NW = F::NW, NP = F::NW / 4;
  __shared__ uint32_t lds[3 * NL * WS_THREADS];
  const uint32_t kk = blockIdx.x, tid = threadIdx.x;
  Proj<F> acc;
  proj_set_zero<F>(acc);
#pragma unroll 1
  for (uint32_t j = tid; j < nchunks; j += WS_THREADS) {
    Proj<F> Q;
    proj_load(Q, columns + ((uint64_t)kk * nchunks + j) * (3 * NL));
    proj_add<F>(acc, acc, Q);
  }
#pragma unroll 1
  for (uint32_t s = WS_THREADS / 2; s >= 1; s >>= 1) {
    if (tid >= s && tid < 2 * s) {
#pragma unroll
      for (int l = 0; l < NL; l++) {
        lds[(l)*WS_THREADS + tid] = acc.X.l[l];
        lds[(NL + l) * WS_THREADS + tid] = acc.Y.l[l];
        lds[(2 * NL + l) * WS_THREADS + tid] = acc.Z.l[l];
      }
    }
    __syncthreads();
    if (tid < s) {
      Proj<F> Q;
#pragma unroll
      for (int l = 0; l < NL; l++) {
        Q.X.l[l] = lds[(l)*WS_THREADS + tid + s];
        Q.Y.l[l] = lds[(NL + l) * WS_THREADS + tid + s];
        Q.Z.l[l] = lds[(2 * NL + l) * WS_THREADS + tid + s];
      }
      proj_add<F>(acc, acc, Q);
    }
    __syncthreads();
  }
  if (tid == 0) {
    fe_reduce_2p<F>(acc.X);
    fe_reduce_2p<F>(acc.Y);
    fe_reduce_2p<F>(acc.Z);
    uint32_t* dst = partials + (uint64_t)kk * PART_WORDS;
    uint32_t w[NW];
#pragma unroll
    for (int j = 0; j < PART_WORDS; j++) dst[j] = 0;
    fe_pack<F>(w, acc.X);
#pragma unroll
    for (int j = 0; j < NW; j++) dst[j] = w[j];
    fe_pack<F>(w, acc.Y);
#pragma unroll
    for (int j = 0; j < NW; j++) dst[12 + j] = w[j];
    fe_pack<F>(w, acc.Z);
#pragma unroll
    for (int j = 0; j < NW; j++) dst[24 + j] = w[j];
  }
}

// first stage of the window sum when a window has many chunk columns: block (b, kk) tree-sums columns
// [b * per_block, (b + 1) * per_block) of window kk into one raw projective point, so that no lane adds more
// than a couple of columns serially (the sum of 8192 columns drops from 40 dependent additions to ~18)
template <class CV>
__global__ void __launch_bounds__(WS_THREADS) k_column_tree(uint32_t* out, const uint32_t* columns, uint32_t nchunks,
                                                            uint32_t per_block) {
  using F = typename CV::F;
  constexpr int NL = F::NL, NW = F::NW, NP = F::NW / 4;
  __shared__ uint32_t lds[3 * NL * WS_THREADS];
  const uint32_t b = blockIdx.x, kk = blockIdx.y, tid = threadIdx.x, nblk = gridDim.x;
  const uint32_t beg = b * per_block, end = min(beg + per_block, nchunks);
  Proj<F> acc;
  proj_set_zero<F>(acc);
#pragma unroll 1
  for (uint32_t j = beg + tid; j < end; j += WS_THREADS) {
    Proj<F> Q;
    proj_load(Q, columns + ((uint64_t)kk * nchunks + j) * (3 * NL));
    proj_add<F>(acc, acc, Q);
  }
#pragma unroll 1
  for (uint32_t s = WS_THREADS / 2; s >= 1; s >>= 1) {
    if (tid >= s && tid < 2 * s) {
#pragma unroll
      for (int l = 0; l < NL; l++) {
        lds[(l)*WS_THREADS + tid] = acc.X.l[l];
        lds[(NL + l) * WS_THREADS + tid] = acc.Y.l[l];
        lds[(2 * NL + l) * WS_THREADS + tid] = acc.Z.l[l];
      }
    }
    __syncthreads();
    if (tid < s) {
      Proj<F> Q;
#pragma unroll
      for (int l = 0; l < NL; l++) {
        Q.X.l[l] = lds[(l)*WS_THREADS + tid + s];
        Q.Y.l[l] = lds[(NL + l) * WS_THREADS + tid + s];
        Q.Z.l[l] = lds[(2 * NL + l) * WS_THREADS + tid + s];
      }
      proj_add<F>(acc, acc, Q);
    }
    __syncthreads();
  }
  if (tid == 0) proj_store(out + ((uint64_t)kk * nblk + b) * (3 * NL), acc);
}

// Bit-sliced weighting of the chunk sums:  sum_ch (ch * TC) * row_ch = TC * sum_b 2^b * S_b  with
// S_b = sum of row_ch over the chunks whose index has bit b set.  Wave (blk, y, kk) sums, for window kk,
//   y <  nbits : its share of the n_in / 2 rows whose chunk index has bit y set (enumerated directly),
//   y == nbits : its share of the n_in local triangles (columns), unmasked.
// masked = 0 is the second stage: plain sums of the first stage's [kk][y][n_in] block results.
// The K * (nbits + 1) results go to the host, which applies the 2^b weights with nbits doublings per window --
// one chain per window instead of one per lane (the reference applies the same weight by double-and-add per
// chunk, src/msm-batched-affine.ts:574-580).
//
// One wave per block: every lane adds a few elements serially, then the lane sums are folded through LDS in
// log2 steps.  A projective addition is ~7000 dependent VALU instructions (15 us for a lone wave), so what
// matters is the number of additions in sequence, not their count: ~8 + 6 here, then log2(n_in) in the second stage.
constexpr int BT_THREADS = 64;

// amdgpu_waves_per_eu(1, 1): one wave per SIMD.  The work is a chain of dependent additions per wave; a second wave on
// the same SIMD halves the speed of both, and the dispatcher fills a SIMD before it moves to the next one (measured:
// 40 us per addition in sequence with two resident waves against 19 us alone).
// What a kind of point must offer the bit tree: raw words (limb w of the point as it sits in registers), the group addition,
// and the packed window sum the host reads.
template <class F>
struct ProjPT {
  using P = Proj<F>;
  static constexpr int W = 3 * F::NL;     // raw words per point between the reduction kernels
  static constexpr int PW = PART_WORDS;   // words of a packed window sum
  static MSM_DEV void zero(P& p) { proj_set_zero<F>(p); }
  static MSM_DEV void add(P& r, const P& a, const P& b) { proj_add<F>(r, a, b); }
  static MSM_DEV uint32_t& word(P& p, int w) { return w < F::NL ? p.X.l[w] : w < 2 * F::NL ? p.Y.l[w - F::NL] : p.Z.l[w - 2 * F::NL]; }
  static MSM_DEV void pack(uint32_t* dst, P& acc) {
    constexpr int NW = F::NW;
    fe_reduce_2p<F>(acc.X);
    fe_reduce_2p<F>(acc.Y);
    fe_reduce_2p<F>(acc.Z);
    uint32_t w[NW];
#pragma unroll
    for (int j = 0; j < PART_WORDS; j++) dst[j] = 0;
    fe_pack<F>(w, acc.X);
#pragma unroll
    for (int j = 0; j < NW; j++) dst[j] = w[j];
    fe_pack<F>(w, acc.Y);
#pragma unroll
    for (int j = 0; j < NW; j++) dst[12 + j] = w[j];
    fe_pack<F>(w, acc.Z);
#pragma unroll
    for (int j = 0; j < NW; j++) dst[24 + j] = w[j];
  }
};

// the body of k_bit_tree / k_te_bit_tree (see above); one wave per block
template <class PT>
MSM_DEV void bit_tree_body(uint32_t* out, const uint32_t* rows, const uint32_t* tris, uint32_t n_in, uint32_t nbits, int masked,
                           int pack_out, uint32_t nblk_out, uint32_t* lds) {
  using P = typename PT::P;
  constexpr int W = PT::W;
  // First stage (masked): grid.x enumerates, per window, nbits masked sums of nblk_out / 2 blocks each and then the
  // triangle sum of nblk_out blocks -- the masked sums have half as many elements, so every wave of the launch has
  // the same number of additions in sequence.  Second stage (masked == 2): grid (1, nbits + 1, kc).
  uint32_t blk = blockIdx.x, y = blockIdx.y, nblk = gridDim.x;
  const uint32_t kk = blockIdx.z, tid = threadIdx.x;
  if (masked >= 3) {
    // Two-dimensional form (round 6).  The masked sums above add every row to half of the nbits sums: nchunks * nbits / 2
    // additions.  With the chunk index cut as ch = hi * M_lo + lo,  sum_ch ch * row_ch = M_lo * sum_hi hi * A_hi + sum_lo lo * B_lo
    // with A_hi = sum_lo row (a run of M_lo rows) and B_lo = sum_hi row (a column of M_hi rows): 2 * nchunks additions, and the
    // bit-sliced weighting then runs over the M_hi + M_lo sums only -- S_b = masked sum over B for the low bits, over A for the
    // high ones: the same nbits + 1 results, so the host tail is unchanged.
    //   masked == 3: grid (M_hi + M_lo + nblk_out, 1, kc): block -> A_hi | B_lo | a share of the local triangles; raw points to
    //                out[kk][M_hi + M_lo + nblk_out]
    //   masked == 4: grid (1, nbits + 1, kc) over those (`rows` = the first stage's output): y < h0: bit y of lo over B;
    //                y < nbits: bit y - h0 of hi over A; y == nbits: the nblk_out triangle shares
    const uint32_t h1 = nbits / 2, h0 = nbits - h1, M_lo = 1u << h0, M_hi = 1u << h1, per_kk = M_hi + M_lo + nblk_out;
    uint32_t count, kind;   // kind 0: A run, 1: B column, 2: triangle share (first stage); 3: masked over stage-1 slots, 4: plain over slots
    uint32_t a0 = 0, a1 = 0;
    if (masked == 3) {
      if (blk < M_hi) { kind = 0; count = M_lo; a0 = blk * M_lo; }
      else if (blk < M_hi + M_lo) { kind = 1; count = M_hi; a0 = blk - M_hi; }
      else {
        kind = 2;
        const uint32_t t = blk - M_hi - M_lo, span = (n_in + nblk_out - 1) / nblk_out;
        a0 = min(t * span, n_in);
        count = min(a0 + span, n_in) - a0;
      }
    } else {
      if (y < h0) { kind = 3; count = M_lo >> 1; a0 = M_hi; a1 = y; }
      else if (y < nbits) { kind = 3; count = M_hi >> 1; a0 = 0; a1 = y - h0; }
      else { kind = 4; count = nblk_out; a0 = M_hi + M_lo; }
    }
    uint32_t m = 1;
    while (m < count && m < BT_THREADS) m <<= 1;
    const uint32_t gl = tid & (m - 1);
    P acc;
    PT::zero(acc);
#pragma unroll 1
    for (uint32_t i = gl; i < count; i += m) {
      P Q;
      if (kind <= 2) {
        const uint32_t j = kind == 1 ? i * M_lo + a0 : a0 + i;
        const uint32_t* pl = (kind == 2 ? tris : rows) + (uint64_t)kk * W * n_in;
#pragma unroll
        for (int w = 0; w < W; w++) PT::word(Q, w) = pl[(uint64_t)w * n_in + j];
      } else {
        const uint32_t j = kind == 3 ? ((((i >> a1) << 1) | 1u) << a1) | (i & ((1u << a1) - 1u)) : i;
        const uint32_t* pp = rows + ((uint64_t)kk * per_kk + a0 + j) * W;
#pragma unroll
        for (int w = 0; w < W; w++) PT::word(Q, w) = pp[w];
      }
      PT::add(acc, acc, Q);
    }
#pragma unroll 1
    for (uint32_t s = m >> 1; s >= 1; s >>= 1) {
#pragma unroll
      for (int w = 0; w < W; w++) lds[w * BT_THREADS + tid] = PT::word(acc, w);
      __syncthreads();
      const uint32_t partner = (tid & ~(m - 1)) | ((gl + s) & (m - 1));
      P Q;
#pragma unroll
      for (int w = 0; w < W; w++) PT::word(Q, w) = lds[w * BT_THREADS + partner];
      __syncthreads();
      PT::add(acc, acc, Q);
    }
    if (tid == 0) {
      if (masked == 3) {
        uint32_t* o = out + ((uint64_t)kk * per_kk + blk) * W;
#pragma unroll
        for (int w = 0; w < W; w++) o[w] = PT::word(acc, w);
      } else {
        PT::pack(out + ((uint64_t)kk * (nbits + 1) + y) * PT::PW, acc);
      }
    }
    return;
  }
  if (masked == 1) {
    const uint32_t half = nblk_out >> 1;
    if (blk < nbits * half) { y = blk / half; blk -= y * half; nblk = half; }
    else { blk -= nbits * half; y = nbits; nblk = nblk_out; }
  }
  // masked == 1: first stage.  masked == 2: second stage over the first stage's block sums -- a masked sum filled only the
  // first half of its nblk_out slots (the rest is never written, never read: all-zero words are the identity of a projective
  // point but not of an extended Edwards one).  masked == 0: plain sums of n_in points.
  const bool sel = masked == 1 && y < nbits;         // n_in = 2^nbits on the masked stage
  const uint32_t count = (sel || (masked == 2 && y < nbits)) ? n_in >> 1 : n_in;
  if (masked == 2) masked = 0;
  const uint32_t span = (count + nblk - 1) / nblk;
  const uint32_t beg = min(blk * span, count), end = min(beg + span, count);
  // first stage: planar arrays [kk][W][n_in] written by the bucket-reduce kernel; second stage: [kk][y][n_in] points of W words
  const uint64_t base = ((uint64_t)kk * (nbits + 1) + y) * n_in;
  const uint32_t* src = (masked && y == nbits) ? tris : rows;
  // Every lane takes part in every addition: a wave whose EXEC mask is down to a few lanes runs this arithmetic
  // up to 3.5x slower per instruction when the chip is busy (tools/ubench_exec.hip), so the usual halving tree --
  // 32, 16, ... 1 active lanes -- is the worst shape.  Instead the lanes form groups of m = min(64, 2^ceil(log2 n))
  // lanes, each group sums all n elements of the block (64 / m groups do the same work redundantly), and the fold is
  // a cyclic all-reduce inside the group: after log2 m steps every lane holds the total.
  const uint32_t n_el = end - beg;
  uint32_t m = 1;
  while (m < n_el && m < BT_THREADS) m <<= 1;
  const uint32_t gl = tid & (m - 1);
  P acc;
  PT::zero(acc);
#pragma unroll 1
  for (uint32_t i = beg + gl; i < end; i += m) {
    // i-th index with bit y set: insert a one at bit position y
    const uint32_t j = sel ? ((((i >> y) << 1) | 1u) << y) | (i & ((1u << y) - 1u)) : i;
    P Q;
    if (masked) {
      const uint32_t* pl = src + (uint64_t)kk * W * n_in;
#pragma unroll
      for (int w = 0; w < W; w++) PT::word(Q, w) = pl[(uint64_t)w * n_in + j];
    } else {
      const uint32_t* pp = src + (base + j) * W;
#pragma unroll
      for (int w = 0; w < W; w++) PT::word(Q, w) = pp[w];
    }
    PT::add(acc, acc, Q);
  }
#pragma unroll 1
  for (uint32_t s = m >> 1; s >= 1; s >>= 1) {
#pragma unroll
    for (int w = 0; w < W; w++) lds[w * BT_THREADS + tid] = PT::word(acc, w);
    __syncthreads();
    const uint32_t partner = (tid & ~(m - 1)) | ((gl + s) & (m - 1));
    P Q;
#pragma unroll
    for (int w = 0; w < W; w++) PT::word(Q, w) = lds[w * BT_THREADS + partner];
    __syncthreads();
    PT::add(acc, acc, Q);
  }
  if (tid == 0) {
    const uint64_t o = ((uint64_t)kk * (nbits + 1) + y) * nblk_out + blk;
    if (!pack_out) {
#pragma unroll
      for (int w = 0; w < W; w++) out[o * W + w] = PT::word(acc, w);
    } else {
      PT::pack(out + o * PT::PW, acc);
    }
  }
}

template <class CV>
__global__ void __launch_bounds__(BT_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_bit_tree(uint32_t* out, const uint32_t* rows, const uint32_t* tris,
                                                         uint32_t n_in, uint32_t nbits, int masked, int pack_out,
                                                         uint32_t nblk_out) {
  __shared__ uint32_t lds[ProjPT<typename CV::F>::W * BT_THREADS];
  bit_tree_body<ProjPT<typename CV::F>>(out, rows, tris, n_in, nbits, masked, pack_out, nblk_out, lds);
}

// ---------------------------------------------------------------------------------------------
// test kernels: element-wise field / curve / GLV operations for parity tests against the oracle
// (the fine-grained operator table of src/field-msm.ts:86-123 as a GPU debug surface)
// ---------------------------------------------------------------------------------------------

enum : int { OP_MUL = 0, OP_SQR = 1, OP_ADD = 2, OP_SUB = 3, OP_INV = 4, OP_TO_MONT = 5, OP_FROM_MONT = 6,
             OP_INV_FERMAT = 7, OP_INV_KALISKI = 8, OP_INV_WORDSLICED = 9 };

// a, b, out: n x 12 packed words; values are canonical Montgomery form unless the op says otherwise
template <class CV>
__global__ void __launch_bounds__(256) k_test_fp(uint32_t* out, const uint32_t* a, const uint32_t* b, uint32_t n, int op) {
  using F = typename CV::F;
  constexpr int NL = F::NL, NW = F::NW, NP = F::NW / 4;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<F> x, y, r;
  fe_load<F>(x, a + (uint64_t)i * NW);
  fe_load<F>(y, b + (uint64_t)i * NW);
  switch (op) {
    case OP_MUL: fe_mul<F>(r, x, y); break;
    case OP_SQR: fe_sqr<F>(r, x); break;
    case OP_ADD: fe_add<F>(r, x, y); break;
    case OP_SUB: fe_sub_p<F>(r, x, y); break;
    case OP_INV: fe_inv<F>(r, x); break;
    case OP_INV_FERMAT: fe_inv_fermat<F>(r, x); break;     // the inversion variants of SURVEY section 8(f)-3
    case OP_INV_KALISKI: fe_inv_kaliski<F>(r, x); break;
    case OP_INV_WORDSLICED: fe_inv_wordsliced<F>(r, x); break;   // src/inverse/faster-inverse-wasm.ts:133-343
    case OP_TO_MONT: {
      Fe<F> r2;
#pragma unroll
      for (int l = 0; l < NL; l++) r2.l[l] = F::R2[l];
      fe_mul<F>(r, x, r2);
      break;
    }
    default: {
      Fe<F> one;
      fe_set_zero<F>(one);
      one.l[0] = 1;
      fe_mul<F>(r, x, one);
      break;
    }
  }
  fe_reduce_4p<F>(r);
  fe_store<F>(out + (uint64_t)i * NW, r);
}

// `batchInverse` (src/wasm/inverse.ts:220-271, JS twin src/curve-affine.ts:692-727): Montgomery's trick over a
// contiguous array; lane t inverts elements [t * per_lane, (t + 1) * per_lane) with ONE fe_inv.
// Operands and results are canonical Montgomery-form words; zeros are not allowed (as in the reference).
template <class CV>
__global__ void __launch_bounds__(256) k_test_batch_inverse(uint32_t* out, const uint32_t* xs, uint32_t n, uint32_t per_lane) {
  using F = typename CV::F;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t beg = (uint64_t)t * per_lane, end = min(beg + per_lane, (uint64_t)n);
  if (beg >= end) return;
  Fe<F> acc, x;
  fe_set_one<F>(acc);
  for (uint64_t i = beg; i < end; i++) {            // out[i] temporarily holds the prefix product before x_i
    fe_reduce_2p<F>(acc);
    fe_store<F>(out + i * F::NW, acc);
    fe_load<F>(x, xs + i * F::NW);
    fe_mul<F>(acc, acc, x);
  }
  Fe<F> inv, pre, r;
  fe_inv<F>(inv, acc);
  for (uint64_t i = end; i-- > beg;) {
    fe_load<F>(pre, out + i * F::NW);
    fe_load<F>(x, xs + i * F::NW);
    fe_mul<F>(r, inv, pre);
    fe_mul<F>(inv, inv, x);
    fe_reduce_2p<F>(r);
    fe_store<F>(out + i * F::NW, r);
  }
}

// Raw-limb multiplier test: operands are NL 30-bit limbs exactly as given -- unreduced sums, all-ones limbs -- and the
// result limbs are stored as they leave fe_mul / fe_sqr (no reduction): the worst-case side of src/field.test.ts:27-155.
template <class CV>
__global__ void __launch_bounds__(256) k_test_fp_raw(uint32_t* out, const uint32_t* a, const uint32_t* b, uint32_t n, int op) {
  using F = typename CV::F;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<F> x, y, r;
#pragma unroll
  for (int l = 0; l < F::NL; l++) { x.l[l] = a[(uint64_t)i * F::NL + l]; y.l[l] = b[(uint64_t)i * F::NL + l]; }
  if (op == OP_SQR) fe_sqr<F>(r, x);
  else fe_mul<F>(r, x, y);
#pragma unroll
  for (int l = 0; l < F::NL; l++) out[(uint64_t)i * F::NL + l] = r.l[l];
}

// Curve operators on projective points (X, Y, Z: 3 x 12 canonical plain-integer words each, any representative):
// op 0 = proj_add (general, every edge case), 1 = proj_double, 2 = proj_add_mixed (Q affine: Q.Z ignored, Q == (0, 0) is
// the identity).  Operator-level counterpart of src/curve-projective.test.ts:77-208.
enum : int { CURVE_OP_ADD = 0, CURVE_OP_DOUBLE = 1, CURVE_OP_ADD_MIXED = 2 };
template <class CV>
__global__ void __launch_bounds__(64) k_test_curve_op(uint32_t* out, const uint32_t* pp, const uint32_t* qq, uint32_t n, int op) {
  using F = typename CV::F;
  constexpr int NL = F::NL, NW = F::NW, NP = F::NW / 4;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<F> r2, one;
#pragma unroll
  for (int l = 0; l < NL; l++) r2.l[l] = F::R2[l];
  fe_set_zero<F>(one);
  one.l[0] = 1;
  Proj<F> P, Q, R;
  Fe<F>* co[6] = {&P.X, &P.Y, &P.Z, &Q.X, &Q.Y, &Q.Z};
  bool q_zero_xy = true;
  for (int j = 0; j < 6; j++) {
    const uint32_t* src = (j < 3 ? pp : qq) + (uint64_t)i * (3 * NW) + (j % 3) * NW;
    fe_load<F>(*co[j], src);
    if (j == 3 || j == 4) q_zero_xy = q_zero_xy && fe_is_zero_canonical<F>(*co[j]);
    fe_mul<F>(*co[j], *co[j], r2);      // to Montgomery form, < 1.5 p
  }
  if (op == CURVE_OP_DOUBLE) proj_double<F>(R, P);
  else if (op == CURVE_OP_ADD_MIXED) proj_add_mixed<F>(R, P, Q, q_zero_xy);
  else proj_add<F>(R, P, Q);
  Fe<F>* ro[3] = {&R.X, &R.Y, &R.Z};
  for (int j = 0; j < 3; j++) {
    fe_mul<F>(*ro[j], *ro[j], one);     // leave Montgomery form
    fe_reduce_4p<F>(*ro[j]);
    fe_store<F>(out + (uint64_t)i * (3 * NW) + j * NW, *ro[j]);
  }
}

// out: n x 10 words: |s0| (4), |s1| (4), neg0, neg1
template <class CV>
__global__ void __launch_bounds__(256) k_test_glv(uint32_t* out, const uint32_t* scalars, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
#pragma unroll
  for (int j = 0; j < 8; j++) s[j] = scalars[(uint64_t)i * 8 + j];
  GlvHalf h0, h1;
  glv_decompose<typename CV::G>(h0, h1, s);
  uint32_t* o = out + (uint64_t)i * 10;
#pragma unroll
  for (int j = 0; j < 4; j++) { o[j] = h0.mag[j]; o[4 + j] = h1.mag[j]; }
  o[8] = h0.neg; o[9] = h1.neg;
}

}  // namespace msm
