// Montgomery field arithmetic for gfx950, one field element per lane.
//
// Replaces the reference's generated-WebAssembly field backend:
//   multiply / square      src/wasm/multiply-montgomery.ts:58-215
//   add / sub / reduce     src/wasm/field-arithmetic.ts:32-166
//   inverse                src/wasm/inverse.ts:191-218 (here: division steps, fe_inv; Fermat / Kaliski / word-sliced kept as cross-checks)
//   packed-bytes codecs    src/wasm/field-helpers.ts:211-301
//
// Register form ("Fe"): NL limbs of 30 bits, radix R = 2^(30*NL) (2^390 for the 377-bit prime,
// 2^270 for the 253-bit prime).  30-bit limbs let a whole 13-term column of 60-bit products sit in
// a 64-bit v_mad_u64_u32 accumulator without carry handling -- gfx950 issues v_mad_u64_u32 at
// nearly the plain-VALU rate (measured: tools/ubench_int.hip), so instruction count is what
// matters and carry-flag chains for 32-bit limbs would double it.
// Memory form: NW packed 32-bit little-endian words, Montgomery representation, CANONICAL
// (value in [0, p)) so equality is a plain word compare.
//
// Value-range discipline (the reference keeps values < 2p with R = 2^406): fe_mul / fe_sqr accept any operands with
// normalised limbs and return a value < p + a b / R, i.e. < 1.5 p whenever a b < R p / 2 -- that is a b < 2^12 p^2 for
// the 377-bit prime (R = 2^13 p), 2^14 p^2 for the 253-bit prime, and 2^8 p^2 for BLS12-381 (R = 2^9 p), which still
// covers every product the kernels form (operands below 4.5 p and 2 p).  So sums / differences of a few elements may be
// multiplied without reduction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "constants_gen.h"

namespace msm {

constexpr int LB = 30;
constexpr uint32_t LMASK = (1u << LB) - 1;

// __host__ as well: the same templates are unit-tested on the CPU (tests/csrc/field_host.hip)
#define MSM_DEV __host__ __device__ __forceinline__

template <class C>
struct Fe {
  uint32_t l[C::NL];
};

// ---------------------------------------------------------------- pack / unpack

// low 32 bits of (hi:lo) >> sh, 0 < sh < 32 (v_alignbit_b32 on the device)
MSM_DEV uint32_t fe_funnel_r(uint32_t lo, uint32_t hi, int sh) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
}

// c + a * K for the small constants K = 1 and 4 as ONE v_mad_u64_u32.  Written as plain C the compiler turns them
// into a zero-extension (v_mov) plus a 64-bit shift-add; the multiplier's carry steps use them 25 times.
template <int K>
MSM_DEV uint64_t fe_mad_const(uint32_t a, uint64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t cy;   // carry-out operand of the VOP3b encoding, unused; accumulate in place: no extra registers
  if (K == 1) asm("v_mad_u64_u32 %0, %1, %2, 1, %0" : "+v"(c), "=s"(cy) : "v"(a));
  else asm("v_mad_u64_u32 %0, %1, %2, 4, %0" : "+v"(c), "=s"(cy) : "v"(a));
  return c;
#else
  return c + (uint64_t)a * (uint32_t)K;
#endif
}

// words (32-bit packed, little endian) -> 30-bit limbs
template <class C>
MSM_DEV void fe_unpack(Fe<C>& r, const uint32_t (&w)[C::NW]) {
#pragma unroll
  for (int i = 0; i < C::NL; i++) {
    const int bit = LB * i;
    const int wi = bit / 32, sh = bit % 32;
    uint32_t lo = wi < C::NW ? w[wi] : 0u;
    uint32_t hi = (wi + 1) < C::NW ? w[wi + 1] : 0u;
    uint32_t v = sh == 0 ? lo : fe_funnel_r(lo, hi, sh);
    r.l[i] = v & LMASK;
  }
}

// 30-bit limbs (normalized, value < 2^(32*NW)) -> packed words
template <class C>
MSM_DEV void fe_pack(uint32_t (&w)[C::NW], const Fe<C>& a) {
#pragma unroll
  for (int j = 0; j < C::NW; j++) {
    const int bit = 32 * j;
    const int li = bit / LB, sh = bit % LB;  // word j starts inside limb li at bit sh
    uint32_t v = a.l[li] >> sh;
    int have = LB - sh;
    if (li + 1 < C::NL) {
      v |= a.l[li + 1] << have;
      have += LB;
      if (have < 32 && li + 2 < C::NL) v |= a.l[li + 2] << have;
    }
    w[j] = v;
  }
}

// ---------------------------------------------------------------- linear ops on limbs

// carry-normalize limbs that may have grown to < 2^32 (unsigned)
template <class C>
MSM_DEV void fe_norm(Fe<C>& a) {
#pragma unroll
  for (int i = 0; i + 1 < C::NL; i++) {
    a.l[i + 1] += a.l[i] >> LB;
    a.l[i] &= LMASK;
  }
}

// r = a + b (no modular reduction; value bound adds)
template <class C>
MSM_DEV void fe_add(Fe<C>& r, const Fe<C>& a, const Fe<C>& b) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < C::NL; i++) {
    uint32_t t = a.l[i] + b.l[i] + c;
    if (i + 1 < C::NL) {
      c = t >> LB;
      r.l[i] = t & LMASK;
    } else {
      r.l[i] = t;
    }
  }
}

// limb i of K*p for K in {1, 2, 4} (compile-time table pick; tables are only ever value-used
// so they stay compile-time constants in device code)
template <class C, int K>
MSM_DEV constexpr uint32_t fe_kp_limb(int i) {
  static_assert(K == 1 || K == 2 || K == 4, "multiples available: p, 2p, 4p");
  return K == 1 ? C::P[i] : (K == 2 ? C::P2[i] : C::P4[i]);
}

// r = a - b + K*p, requires b < K*p so the result is >= 0.
template <class C, int K>
MSM_DEV void fe_sub_k(Fe<C>& r, const Fe<C>& a, const Fe<C>& b) {
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < C::NL; i++) {
    int32_t t = (int32_t)(a.l[i] + fe_kp_limb<C, K>(i)) - (int32_t)b.l[i] + c;
    if (i + 1 < C::NL) {
      c = t >> LB;  // arithmetic
      r.l[i] = (uint32_t)t & LMASK;
    } else {
      r.l[i] = (uint32_t)t;
    }
  }
}
template <class C>
MSM_DEV void fe_sub_p(Fe<C>& r, const Fe<C>& a, const Fe<C>& b) { fe_sub_k<C, 1>(r, a, b); }   // b < p
template <class C>
MSM_DEV void fe_sub_2p(Fe<C>& r, const Fe<C>& a, const Fe<C>& b) { fe_sub_k<C, 2>(r, a, b); }  // b < 2p
template <class C>
MSM_DEV void fe_sub_4p(Fe<C>& r, const Fe<C>& a, const Fe<C>& b) { fe_sub_k<C, 4>(r, a, b); }  // b < 4p

// a -= K*p if a >= K*p
template <class C, int K>
MSM_DEV void fe_cond_sub(Fe<C>& a) {
  uint32_t t[C::NL];
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < C::NL; i++) {
    int32_t v = (int32_t)a.l[i] - (int32_t)fe_kp_limb<C, K>(i) + c;
    if (i + 1 < C::NL) {
      c = v >> LB;
      t[i] = (uint32_t)v & LMASK;
    } else {
      c = v >> 31;  // sign of the top limb = sign of the whole difference
      t[i] = (uint32_t)v;
    }
  }
  bool neg = c != 0;
#pragma unroll
  for (int i = 0; i < C::NL; i++) a.l[i] = neg ? a.l[i] : t[i];
}

// canonical form for a value < 2p / < 4p
template <class C>
MSM_DEV void fe_reduce_2p(Fe<C>& a) { fe_cond_sub<C, 1>(a); }
template <class C>
MSM_DEV void fe_reduce_4p(Fe<C>& a) {
  fe_cond_sub<C, 2>(a);
  fe_cond_sub<C, 1>(a);
}

template <class C>
MSM_DEV bool fe_is_zero_canonical(const Fe<C>& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < C::NL; i++) o |= a.l[i];
  return o == 0;
}

template <class C>
MSM_DEV void fe_set_one(Fe<C>& r) {  // Montgomery form of 1
#pragma unroll
  for (int i = 0; i < C::NL; i++) r.l[i] = C::ONE[i];
}
template <class C>
MSM_DEV void fe_set_zero(Fe<C>& r) {
#pragma unroll
  for (int i = 0; i < C::NL; i++) r.l[i] = 0;
}

template <class C>
MSM_DEV void fe_select(Fe<C>& r, bool c, const Fe<C>& a, const Fe<C>& b) {  // r = c ? a : b
#pragma unroll
  for (int i = 0; i < C::NL; i++) r.l[i] = c ? a.l[i] : b.l[i];
}

// ---------------------------------------------------------------- Montgomery product

// Can every 64-bit column accumulator of the interleaved product (SQR: of the interleaved square) hold its worst
// case -- all limbs 2^30 - 1, every quotient digit 2^30 - 1 -- when the only overflow guard is a sweep after row
// `guard_row` (-1: none)?  Evaluated at compile time per field: BLS12-377, its scalar field and Pallas need no guard
// at all (their column sums peak at 2^63.99), BLS12-381 needs one.
template <class C, bool SQR>
constexpr bool fe_acc_fits(int guard_row) {
  constexpr int N = C::NL;
  typedef unsigned __int128 u128;
  const u128 M = LMASK, LIM = (u128)1 << 64;
  u128 w[N] = {};
  for (int i = 0; i < N; i++) {
    for (int j = SQR ? i : 0; j < N; j++) w[j] += (SQR && j > i) ? M * (2 * M) : M * M;
    for (int j = 0; j < N; j++)
      if (w[j] >= LIM) return false;
    if (w[0] + M * C::P[0] >= LIM) return false;
    const u128 carry = ((w[0] + M * C::P[0]) >> LB) + 1;
    for (int j = 1; j < N; j++) w[j] += M * C::P[j];
    w[1] += carry;
    for (int j = 0; j < N; j++)
      if (w[j] >= LIM) return false;
    for (int j = 0; j + 1 < N; j++) w[j] = w[j + 1];
    w[N - 1] = 0;
    if (i == guard_row)
      for (int j = 0; j + 1 < N; j++) { w[j + 1] += (w[j] >> 32) * 4; w[j] &= 0xffffffffu; }
  }
  return true;
}
// row after which the guard sweep runs: -1 = no guard needed, -2 = no single sweep suffices
template <class C, bool SQR>
constexpr int fe_guard_row() {
  if (fe_acc_fits<C, SQR>(-1)) return -1;
  for (int d = 0; d < C::NL; d++) {
    if (C::NL / 2 - d >= 0 && fe_acc_fits<C, SQR>(C::NL / 2 - d)) return C::NL / 2 - d;
    if (C::NL / 2 + d < C::NL && fe_acc_fits<C, SQR>(C::NL / 2 + d)) return C::NL / 2 + d;
  }
  return -2;
}

// one reduction row of the interleaved product: add m * p with m chosen so that column 0 vanishes, push its carry
// into column 1.  p == 1 mod 2^30 (BLS12-377 base and scalar fields, Pallas): m = -t0 mod 2^30 needs no multiply
// and m * P[0] = m is a MAD by 1.
template <class C>
MSM_DEV void fe_reduce_row(uint64_t (&t)[C::NL]) {
  constexpr int N = C::NL;
  constexpr bool UNIT = (C::MU == LMASK) && (C::P[0] == 1);
  uint64_t carry;
  uint32_t m;
  if (UNIT) {
    m = (0u - (uint32_t)t[0]) & LMASK;
    carry = fe_mad_const<1>(m, t[0]) >> LB;   // t[0] + m is a multiple of 2^30
  } else {
    m = ((uint32_t)t[0] * C::MU) & LMASK;
    carry = (t[0] + (uint64_t)m * C::P[0]) >> LB;
  }
#pragma unroll
  for (int j = 1; j < N; j++) t[j] += (uint64_t)m * C::P[j];
  t[1] += carry;
#pragma unroll
  for (int j = 0; j + 1 < N; j++) t[j] = t[j + 1];
  t[N - 1] = 0;
}

// overflow guard, not a normalisation: move each accumulator's high word up one column (x 2^32 = 4 x 2^30), one MAD +
// one clear per column; the low words stay as they are (< 2^32)
template <int N>
MSM_DEV void fe_guard_sweep(uint64_t (&t)[N]) {
#pragma unroll
  for (int j = 0; j + 1 < N; j++) {
    t[j + 1] = fe_mad_const<4>((uint32_t)(t[j] >> 32), t[j + 1]);
    t[j] = (uint32_t)t[j];
  }
}

template <int N>
MSM_DEV void fe_carry_out(uint32_t (&r)[N], uint64_t (&t)[N]) {
#pragma unroll
  for (int j = 0; j + 1 < N; j++) {
    t[j + 1] += t[j] >> LB;
    r[j] = (uint32_t)t[j] & LMASK;
  }
  r[N - 1] = (uint32_t)t[N - 1];
}

// Interleaved (CIOS-style) Montgomery multiplication on 30-bit limbs with 64-bit column accumulators.  Row i adds
// a_i * b and m_i * p, then shifts one limb.  Operands: normalised limbs (< 2^30), a * b < 2^12 p^2; result < p + a b / R.
// A field that rides a wider limb layout than it needs (Pallas: 255 bits in the 13-limb layout of the 381-bit curves) has
// C::NLA < C::NL active limbs: every value the kernels form is below 2^(30 NLA), so the partial products of the limbs above
// are zeros and are not computed (81 instead of 169 multiply-adds; the NL reduction rows stay, R is 2^(30 NL)).
template <class C>
MSM_DEV void fe_mul(Fe<C>& r, const Fe<C>& a, const Fe<C>& b) {
  constexpr int N = C::NL, NA = C::NLA;
  constexpr int GUARD = fe_guard_row<C, false>();
  static_assert(GUARD != -2, "no single guard sweep keeps the accumulators below 2^64");
  uint64_t t[N];
#pragma unroll
  for (int j = 0; j < N; j++) t[j] = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    if (i < NA) {
#pragma unroll
      for (int j = 0; j < NA; j++) t[j] += (uint64_t)a.l[i] * b.l[j];
    }
    fe_reduce_row<C>(t);
    if (i == GUARD) fe_guard_sweep<N>(t);
  }
  fe_carry_out<N>(r.l, t);
}

// Interleaved squaring: row i adds a_i^2 at column 2i and 2 a_i a_j (j > i) at column i + j -- positions i .. N-1 of
// the shifted window -- so 13 accumulators suffice (the separate 2N-column square needed 26) and column i is complete
// when reduction row i retires it: every term a_k a_(i-k) comes from a row k <= i / 2.
template <class C>
MSM_DEV void fe_sqr(Fe<C>& r, const Fe<C>& a) {
  constexpr int N = C::NL, NA = C::NLA;
  constexpr int GUARD = fe_guard_row<C, true>();
  static_assert(GUARD != -2, "no single guard sweep keeps the accumulators below 2^64");
  uint32_t a2[N];
#pragma unroll
  for (int j = 0; j < N; j++) a2[j] = a.l[j] << 1;  // < 2^31
  uint64_t t[N];
#pragma unroll
  for (int j = 0; j < N; j++) t[j] = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    if (i < NA) {
      t[i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
      for (int j = i + 1; j < NA; j++) t[j] += (uint64_t)a.l[i] * a2[j];
    }
    fe_reduce_row<C>(t);
    if (i == GUARD) fe_guard_sweep<N>(t);
  }
  fe_carry_out<N>(r.l, t);
}

// ---------------------------------------------------------------- inversion

// a^(p-2) by plain MSB-first square-and-multiply (uniform control flow: the exponent is a
// compile-time constant, so no lane diverges).  Input any value < 2p, Montgomery form; output
// Montgomery form of the inverse, < p + p/2.  a == 0 gives 0.  Kept as the cross-check of fe_inv.
template <class C>
MSM_DEV void fe_inv_fermat(Fe<C>& r, const Fe<C>& a) {
  Fe<C> acc;
  fe_set_one<C>(acc);
#pragma unroll 1
  for (int bit = C::BITS - 1; bit >= 0; bit--) {
    fe_sqr<C>(acc, acc);
    if ((C::PM2W[bit / 32] >> (bit % 32)) & 1u) fe_mul<C>(acc, acc, a);
  }
  r = acc;
}

// wave-uniform "does any lane still have work" (device) / plain predicate (host unit tests)
MSM_DEV bool fe_any_lane(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __any((int)p) != 0;
#else
  return p;
#endif
}

// Kaliski's almost-inverse, the reference's algorithm (src/wasm/inverse.ts:136-218: `almostInverse`, then a
// shift by the missing power of two) -- kept as the third variant for cross-checks and for the comparison in
// a binary gcd with four data-dependent cases per step, which a 64-lane wave executes as all four.
//   u = p, v = a, r = 0, s = 1;  while v > 0: (u even) u /= 2, s *= 2 | (v even) v /= 2, r *= 2 |
//   (u > v) u = (u - v) / 2, r += s, s *= 2 | (else) v = (v - u) / 2, s += r, r *= 2;  k steps in total
// leaves r = -a^-1 2^k mod p.  The input is a R (Montgomery form), so doubling p - r up to the exponent 2 log2 R
// gives a^-1 R^-1 R^2 = a^-1 R directly: no table of powers of two, no final multiplication.
template <class C>
MSM_DEV void fe_inv_kaliski(Fe<C>& out, const Fe<C>& a_in) {
  constexpr int N = C::NL;
  Fe<C> a = a_in;
  fe_reduce_4p<C>(a);   // canonical [0, p)
  uint32_t u[N], v[N], r[N], s[N];
#pragma unroll
  for (int i = 0; i < N; i++) { u[i] = C::P[i]; v[i] = a.l[i]; r[i] = 0; s[i] = 0; }
  s[0] = 1;
  auto shr1 = [](uint32_t (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = (x[i] >> 1) | ((i + 1 < N ? (x[i + 1] & 1u) : 0u) << (LB - 1));
  };
  auto shl1 = [](uint32_t (&x)[N]) {
#pragma unroll
    for (int i = N - 1; i >= 0; i--) x[i] = ((x[i] << 1) & LMASK) | (i ? (x[i - 1] >> (LB - 1)) : 0u);
  };
  auto add = [](uint32_t (&x)[N], const uint32_t (&y)[N]) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { uint32_t t = x[i] + y[i] + c; x[i] = t & LMASK; c = t >> LB; }
  };
  auto sub = [](uint32_t (&x)[N], const uint32_t (&y)[N]) {   // x -= y, returns the borrow
    uint32_t b = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { uint32_t t = x[i] - y[i] - b; x[i] = t & LMASK; b = (t >> 31) & 1u; }
    return b;
  };
  auto gt = [](const uint32_t (&x)[N], const uint32_t (&y)[N]) {
    bool g = false, l = false;
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
      if (!g && !l) { if (x[i] > y[i]) g = true; else if (x[i] < y[i]) l = true; }
    }
    return g;
  };
  int k = 0;
#pragma unroll 1
  for (int it = 0; it < 2 * LB * N; it++) {
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < N; i++) nz |= v[i];
    if (!fe_any_lane(nz != 0)) break;
    if (nz == 0) continue;
    if (!(u[0] & 1u)) { shr1(u); shl1(s); }
    else if (!(v[0] & 1u)) { shr1(v); shl1(r); }
    else if (gt(u, v)) { sub(u, v); shr1(u); add(r, s); shl1(s); }
    else { sub(v, u); shr1(v); add(s, r); shl1(r); }
    k++;
  }
  uint32_t pp[N];
#pragma unroll
  for (int i = 0; i < N; i++) pp[i] = C::P[i];
  // r < 2p: r mod p, then p - r (0 stays 0: a == 0 gives 0 like the other variants)
  {
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = r[i];
    if (!sub(t, pp)) {
#pragma unroll
      for (int i = 0; i < N; i++) r[i] = t[i];
    }
    uint32_t nzr = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { t[i] = pp[i]; nzr |= r[i]; }
    sub(t, r);
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = nzr ? t[i] : 0u;
  }
  // up to the exponent 2 log2 R
#pragma unroll 1
  for (int e = 0; e < 2 * LB * N; e++) {
    if (!fe_any_lane(k + e < 2 * LB * N)) break;
    if (k + e >= 2 * LB * N) continue;
    shl1(r);
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = r[i];
    if (!sub(t, pp)) {
#pragma unroll
      for (int i = 0; i < N; i++) r[i] = t[i];
    }
  }
#pragma unroll
  for (int i = 0; i < N; i++) out.l[i] = r[i];
}

// Word-sliced almost-inverse: the reference's experimental `src/inverse/` (faster-inverse-wasm.ts:133-343, prototype
// faster-inverse.ts:78-177).  Kaliski's binary gcd again, but w = 30 steps at a time are decided on the LOW words of u, v
// (parity) and on 63-bit approximations of their HIGH ends (the comparison u >= v), recording the steps in a 2x2 matrix
//   u' = (u f0 - v g0) / 2^w,  v' = (v g1 - u f1) / 2^w,   r' = r f0 + s g0,  s' = r f1 + s g1
// that is then applied to the full-width values with 64-bit multiply-adds.  A comparison decided on the approximations
// can be wrong; u or v then comes out negative and is negated together with its matrix row ("sign flip").
// Ends with u = 0, v = 1 and a s = 2^k (mod p) after k = 30 x (outer iterations) steps; the input is a R (Montgomery
// form), so doubling s up to the exponent 2 log2 R gives a^-1 R, as in fe_inv_kaliski.  Fourth inversion variant of
// SURVEY section 8(f)-3, kept for cross-checks and timing (tools/ubench_inv.hip); the MSM uses fe_inv.
template <int N>
MSM_DEV void fe_signed_negate(int32_t (&v)[N], int32_t mask);   // below, with the division-step inverse

template <int N>
MSM_DEV int fe_ws_bitlen(const int32_t (&x)[N]) {
  int len = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const uint32_t w = (uint32_t)x[i];
    int hb = 0;
    for (uint32_t t = w; t; t >>= 1) hb++;   // host-friendly; the device compiler turns it into a find-first-set
    if (w) len = LB * i + hb;
  }
  return len;
}
// bits [shift, shift + 63) of a non-negative N-limb value
template <int N>
MSM_DEV int64_t fe_ws_hi63(const int32_t (&x)[N], int shift) {
  uint64_t r = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const int pos = LB * i - shift;   // where limb i lands
    const uint64_t w = (uint32_t)x[i];
    if (pos >= 64 || pos <= -LB) continue;
    r |= pos >= 0 ? (w << pos) : (w >> (-pos));
  }
  return (int64_t)(r & 0x7FFFFFFFFFFFFFFFull);
}
// x <- (x a - y b) / 2^30 (exact), y <- (y d - x c) / 2^30: limbs 0..N-2 normalised, top limb signed
template <int N>
MSM_DEV void fe_ws_update_uv(int32_t (&x)[N], int32_t (&y)[N], int32_t a, int32_t b, int32_t c, int32_t d) {
  int64_t cx = 0, cy = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    const int64_t tx = (int64_t)x[j] * a - (int64_t)y[j] * b + cx;
    const int64_t ty = (int64_t)y[j] * d - (int64_t)x[j] * c + cy;
    if (j > 0) { x[j - 1] = (int32_t)((uint32_t)tx & LMASK); y[j - 1] = (int32_t)((uint32_t)ty & LMASK); }
    cx = tx >> LB;
    cy = ty >> LB;
  }
  x[N - 1] = (int32_t)cx;
  y[N - 1] = (int32_t)cy;
}
// x <- x a + y b, y <- x c + y d over M limbs (no shift), top limb signed
template <int M>
MSM_DEV void fe_ws_update_rs(int32_t (&x)[M], int32_t (&y)[M], int32_t a, int32_t b, int32_t c, int32_t d) {
  int64_t cx = 0, cy = 0;
#pragma unroll
  for (int j = 0; j < M; j++) {
    const int64_t tx = (int64_t)x[j] * a + (int64_t)y[j] * b + cx;
    const int64_t ty = (int64_t)x[j] * c + (int64_t)y[j] * d + cy;
    if (j + 1 < M) {
      x[j] = (int32_t)((uint32_t)tx & LMASK); y[j] = (int32_t)((uint32_t)ty & LMASK);
      cx = tx >> LB; cy = ty >> LB;
    } else {
      x[j] = (int32_t)tx; y[j] = (int32_t)ty;
    }
  }
}

template <class C>
MSM_DEV void fe_inv_wordsliced(Fe<C>& out, const Fe<C>& a_in) {
  constexpr int N = C::NL;
  Fe<C> a = a_in;
  fe_reduce_4p<C>(a);   // canonical [0, p)
  int32_t u[N], v[N], r[N + 1], s[N + 1];
#pragma unroll
  for (int i = 0; i < N; i++) { u[i] = (int32_t)C::P[i]; v[i] = (int32_t)a.l[i]; r[i] = 0; s[i] = 0; }
  r[N] = 0; s[N] = 0;
  s[0] = 1;
  int k = 0;
#pragma unroll 1
  for (int it = 0; it < 2 * N; it++) {   // the reference's bound: forLoop1(i, 0, 2 n)
    uint32_t unz = 0, vnz = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { unz |= (uint32_t)u[i]; vnz |= (uint32_t)v[i]; }
    const bool live = unz != 0 && vnz != 0;   // a == 0: v = 0 from the start, nothing to invert (result 0)
    if (!fe_any_lane(live)) break;
    if (!live) continue;
    const int ulen = fe_ws_bitlen<N>(u), vlen = fe_ws_bitlen<N>(v);
    int shift = (ulen > vlen ? ulen : vlen) - 63;
    if (shift < 0) shift = 0;
    int64_t uhi = fe_ws_hi63<N>(u, shift), vhi = fe_ws_hi63<N>(v, shift);
    int32_t ulo = u[0], vlo = v[0];
    int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll 2
    for (int j = 0; j < LB; j++) {
      const bool ue = (ulo & 1) == 0, ve = (vlo & 1) == 0;
      const bool sub_u = !ue && !ve && vhi <= uhi;       // u <- (u - v) / 2
      const bool sub_v = !ue && !ve && !(vhi <= uhi);    // v <- (v - u) / 2
      const bool side_u = ue || sub_u;                   // this step halves u (else v)
      if (side_u) {
        uhi = (uhi - (sub_u ? vhi : 0)) >> 1;
        ulo = (ulo - (sub_u ? vlo : 0)) >> 1;
        f0 += sub_u ? f1 : 0; g0 += sub_u ? g1 : 0;
        f1 <<= 1; g1 <<= 1;
      } else {
        vhi = (vhi - (sub_v ? uhi : 0)) >> 1;
        vlo = (vlo - (sub_v ? ulo : 0)) >> 1;
        f1 += sub_v ? f0 : 0; g1 += sub_v ? g0 : 0;
        f0 <<= 1; g0 <<= 1;
      }
    }
    k += LB;
    fe_ws_update_uv<N>(u, v, f0, g0, f1, g1);
    // a comparison decided on the 63-bit approximations can be off: a negative result is negated with its matrix row
    if (u[N - 1] < 0) { fe_signed_negate<N>(u, -1); f0 = -f0; g0 = -g0; }
    if (v[N - 1] < 0) { fe_signed_negate<N>(v, -1); f1 = -f1; g1 = -g1; }
    fe_ws_update_rs<N + 1>(r, s, f0, g0, f1, g1);
  }
  // v = 1 (or the input was 0 mod p) and a s = +-2^k.  The last batch kept halving u = 0, i.e. doubling s: `makeOdd`
  // (faster-inverse-wasm.ts:47-104, faster-inverse.ts:168-176) strips those factors of two again (at most one limb's
  // worth) and takes them off k; what remains is below p in magnitude.
  const int32_t sneg = s[N] >> 31;            // a sign flip may leave s negative: work on |s|, negate at the end
  fe_signed_negate<N + 1>(s, sneg);
  {
    const uint32_t lo = (uint32_t)s[0];
    int tz = 0;
    for (uint32_t t = lo; tz < LB && !(t & 1u); t >>= 1) tz++;   // lo == 0: a whole limb
#pragma unroll
    for (int i = 0; i <= N; i++) {
      const uint32_t cur = (uint32_t)s[i], nxt = i < N ? (uint32_t)s[i + 1] : 0u;
      s[i] = tz == LB ? (int32_t)nxt : (int32_t)(((cur >> tz) | (nxt << (LB - tz))) & (i < N ? LMASK : 0xFFFFFFFFu));
    }
    k -= tz;
  }
  int32_t pp[N + 1];
#pragma unroll
  for (int i = 0; i < N; i++) pp[i] = (int32_t)C::P[i];
  pp[N] = 0;
  auto sub_p_if_ge = [&]() {
    int32_t t[N + 1], c = 0;
#pragma unroll
    for (int i = 0; i <= N; i++) {
      int32_t d = s[i] - pp[i] + c;
      if (i < N) { c = d >> LB; t[i] = d & (int32_t)LMASK; } else { t[i] = d; }
    }
    const bool ge = t[N] >= 0;
#pragma unroll
    for (int i = 0; i <= N; i++) s[i] = ge ? t[i] : s[i];
  };
  sub_p_if_ge();
  if (sneg) {   // -|s| mod p = p - |s| (|s| != 0: it is invertible)
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i <= N; i++) {
      int32_t d = pp[i] - s[i] + c;
      if (i < N) { c = d >> LB; s[i] = d & (int32_t)LMASK; } else { s[i] = d; }
    }
  }
  uint32_t snz = 0;
#pragma unroll
  for (int i = 0; i < N; i++) snz |= (uint32_t)u[i];
  const bool zero_in = snz != 0;           // u never reached 0: the input was 0 mod p -> 0, like the other variants
#pragma unroll 1
  for (int e = 0; e < 2 * LB * N; e++) {
    if (!fe_any_lane(k + e < 2 * LB * N)) break;
    if (k + e >= 2 * LB * N) continue;
    int32_t c = 0;   // s <- 2 s
#pragma unroll
    for (int i = 0; i <= N; i++) {
      int32_t t = (s[i] << 1) + c;
      if (i < N) { c = t >> LB; s[i] = t & (int32_t)LMASK; } else { s[i] = t; }
    }
    sub_p_if_ge();
  }
#pragma unroll
  for (int i = 0; i < N; i++) out.l[i] = zero_in ? 0u : (uint32_t)s[i];
}

// Division-step inverse (Bernstein-Yang "safegcd", half-delta variant) on signed 30-bit limbs.
// Plays the role of the reference's Kaliski almost-inverse (src/wasm/inverse.ts:136-218): a binary
// gcd whose per-step decisions only look at the low bits -- but branch-free, so the 64 lanes of a
// wave never diverge, and batched: 30 division steps are run on the low words only and summarised
// in a 2x2 matrix that is then applied to the full-width (f, g) and (d, e) with 64-bit MADs.
// About 27 batches x ~700 VALU ops, i.e. ~40 field multiplications instead of ~570 for Fermat.
struct DivstepMatrix {
  int32_t u, v, q, r;
};

MSM_DEV int32_t fe_divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, DivstepMatrix& t) {
  // invariant after i steps: 2^i f_i = u f_0 + v g_0 (f row pre-scaled), 2^i g_i = q f_0 + r g_0
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll
  for (int i = 0; i < 30; i++) {
    uint32_t c1 = (uint32_t)(zeta >> 31);   // delta > 0
    uint32_t c2 = 0u - (g & 1u);            // g odd
    uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // conditionally negated f row
    g += x & c2; q += y & c2; r += z & c2;
    c1 &= c2;                               // swap case
    zeta = (zeta ^ (int32_t)c1) - 1;
    f += g & c1; u += q & c1; v += r & c1;
    g >>= 1; u <<= 1; v <<= 1;
  }
  t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
  return zeta;
}

// (f, g) <- (u f + v g, q f + r g) / 2^30, exact; limbs 0..N-2 in [0, 2^30), top limb signed
template <int N>
MSM_DEV void fe_update_fg(int32_t (&f)[N], int32_t (&g)[N], const DivstepMatrix& t) {
  int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0];
  int64_t cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
  cf >>= LB; cg >>= LB;
#pragma unroll
  for (int i = 1; i < N; i++) {
    cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
    cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
    f[i - 1] = (int32_t)((uint32_t)cf & LMASK);
    g[i - 1] = (int32_t)((uint32_t)cg & LMASK);
    cf >>= LB; cg >>= LB;
  }
  f[N - 1] = (int32_t)cf;
  g[N - 1] = (int32_t)cg;
}

// (d, e) <- (u d + v e, q d + r e) / 2^30 mod p, kept in (-2p, p)
template <class C>
MSM_DEV void fe_update_de(int32_t (&d)[C::NL], int32_t (&e)[C::NL], const DivstepMatrix& t) {
  constexpr int N = C::NL;
  const int32_t sd = d[N - 1] >> 31, se = e[N - 1] >> 31;
  int32_t md = (t.u & sd) + (t.v & se);
  int32_t me = (t.q & sd) + (t.r & se);
  int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0];
  int64_t ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
  // make the low 30 bits vanish: md, me += -(p^-1 (c + p m)) mod 2^30  (p^-1 mod 2^30 = 1 for the BLS12-377 fields)
  md -= (int32_t)((C::PINV30 * (uint32_t)cd + (uint32_t)md) & LMASK);
  me -= (int32_t)((C::PINV30 * (uint32_t)ce + (uint32_t)me) & LMASK);
  cd += (int64_t)C::P[0] * md;
  ce += (int64_t)C::P[0] * me;
  cd >>= LB; ce >>= LB;
#pragma unroll
  for (int i = 1; i < N; i++) {
    cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + (int64_t)C::P[i] * md;
    ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + (int64_t)C::P[i] * me;
    d[i - 1] = (int32_t)((uint32_t)cd & LMASK);
    e[i - 1] = (int32_t)((uint32_t)ce & LMASK);
    cd >>= LB; ce >>= LB;
  }
  d[N - 1] = (int32_t)cd;
  e[N - 1] = (int32_t)ce;
}

// v += mask & p, then carry-normalise (limbs 0..N-2 into [0, 2^30), top limb signed)
template <class C>
MSM_DEV void fe_signed_add_p(int32_t (&v)[C::NL], int32_t mask) {
  constexpr int N = C::NL;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    int32_t t = v[i] + ((int32_t)C::P[i] & mask) + c;
    if (i + 1 < N) { c = t >> LB; v[i] = t & (int32_t)LMASK; } else { v[i] = t; }
  }
}
template <int N>
MSM_DEV void fe_signed_negate(int32_t (&v)[N], int32_t mask) {  // v = mask ? -v : v
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    int32_t t = ((v[i] ^ mask) - mask) + c;
    if (i + 1 < N) { c = t >> LB; v[i] = t & (int32_t)LMASK; } else { v[i] = t; }
  }
}

// r = a^-1 in Montgomery form (a in Montgomery form, any value < 2p; a == 0 mod p gives 0).
template <class C>
MSM_DEV void fe_inv(Fe<C>& r, const Fe<C>& a) {
  constexpr int N = C::NL;
  int32_t f[N], g[N], d[N], e[N];
#pragma unroll
  for (int i = 0; i < N; i++) { f[i] = (int32_t)C::P[i]; g[i] = (int32_t)a.l[i]; d[i] = 0; e[i] = 0; }
  e[0] = 1;
  int32_t zeta = -1;
  // (49 * 390 + 57) / 17 = 1127 division steps bound the original variant for 390-bit inputs; the
  // half-delta variant used here needs fewer.  Lanes leave early, as a wave, once every g is 0.
#pragma unroll 1
  for (int it = 0; it < 38; it++) {
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < N; i++) nz |= (uint32_t)g[i];
    if (!fe_any_lane(nz != 0)) break;
    DivstepMatrix t;
    zeta = fe_divsteps_30(zeta, (uint32_t)f[0], (uint32_t)g[0], t);
    fe_update_de<C>(d, e, t);
    fe_update_fg<N>(f, g, t);
  }
  // f = +-1 (or +-p if a == 0 mod p); d = f * a^-1 in (-2p, p)
  fe_signed_add_p<C>(d, d[N - 1] >> 31);      // (-p, p)
  fe_signed_negate<N>(d, f[N - 1] >> 31);     // times the sign of f
  fe_signed_add_p<C>(d, d[N - 1] >> 31);      // [0, p)
  Fe<C> x, r3;
#pragma unroll
  for (int i = 0; i < N; i++) { x.l[i] = (uint32_t)d[i]; r3.l[i] = C::R3[i]; }
  // d = (a R)^-1 as a plain integer; times R^3 / R gives a^-1 R
  fe_mul<C>(r, x, r3);
}

// ---------------------------------------------------------------- memory helpers

// canonical packed element <-> registers, AoS (contiguous NW words, 16-byte aligned)
template <class C>
MSM_DEV void fe_load(Fe<C>& r, const uint32_t* p) {
  uint32_t w[C::NW];
  const uint4* p4 = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int j = 0; j < C::NW / 4; j++) {
    uint4 v = p4[j];
    w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
  }
  fe_unpack<C>(r, w);
}

template <class C>
MSM_DEV void fe_store(uint32_t* p, const Fe<C>& a) {
  uint32_t w[C::NW];
  fe_pack<C>(w, a);
  uint4* p4 = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int j = 0; j < C::NW / 4; j++) p4[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}

}  // namespace msm
