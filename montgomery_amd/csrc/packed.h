// Linear field operations on the PACKED form (12 x 32-bit words, canonical [0, p)) with hardware
// carry chains (v_sub_co / v_subb_co / v_addc_co through an SGPR-pair carry).
//
// The 30-bit limb form of field.h is what multiplications want; additions, subtractions and the
// final canonical reduction are 3-4 VALU ops per limb there (no carry flag between 30-bit limbs),
// against 1 op per 32-bit word here.  k_batch_add therefore keeps coordinates packed, does
// x2 - x1, y2 - y1, m^2 - x1 - x2, x1 - x3, m t - y1 and the reductions on words, and only unpacks
// the three values that feed multiplications (reference counterparts: subtract / subtractPositive /
// reduce of src/wasm/field-arithmetic.ts:32-166).
//
// gfx9 allows one SGPR-or-literal read per VALU instruction and the carry-in is one, so constants
// enter the chains through VGPRs (v_and / v_mov with a literal first).
#pragma once
#include "field.h"

namespace msm {

template <int W>
struct PkW {
  uint32_t w[W];
};
using Pk = PkW<12>;    // the 377- / 381-bit fields
using Pk8 = PkW<8>;    // 255-bit fields (Pallas): 8 words

// r = a - b over 384 bits; returns all-ones if the subtraction borrowed (a < b), else 0
__device__ __forceinline__ uint32_t pk_sub(Pk& r, const Pk& a, const Pk& b) {
  uint64_t c;
  uint32_t m;
  asm volatile(
      "v_sub_co_u32 %0, %6, %7, %13\n\t"
      "v_subb_co_u32 %1, %6, %8, %14, %6\n\t"
      "v_subb_co_u32 %2, %6, %9, %15, %6\n\t"
      "v_subb_co_u32 %3, %6, %10, %16, %6\n\t"
      "v_subb_co_u32 %4, %6, %11, %17, %6\n\t"
      "v_subb_co_u32 %5, %6, %12, %18, %6"
      : "=&v"(r.w[0]), "=&v"(r.w[1]), "=&v"(r.w[2]), "=&v"(r.w[3]), "=&v"(r.w[4]), "=&v"(r.w[5]), "=&s"(c)
      : "v"(a.w[0]), "v"(a.w[1]), "v"(a.w[2]), "v"(a.w[3]), "v"(a.w[4]), "v"(a.w[5]), "v"(b.w[0]), "v"(b.w[1]), "v"(b.w[2]),
        "v"(b.w[3]), "v"(b.w[4]), "v"(b.w[5]));
  asm volatile(
      "v_subb_co_u32 %0, %7, %8, %14, %7\n\t"
      "v_subb_co_u32 %1, %7, %9, %15, %7\n\t"
      "v_subb_co_u32 %2, %7, %10, %16, %7\n\t"
      "v_subb_co_u32 %3, %7, %11, %17, %7\n\t"
      "v_subb_co_u32 %4, %7, %12, %18, %7\n\t"
      "v_subb_co_u32 %5, %7, %13, %19, %7\n\t"
      "v_subb_co_u32 %6, %7, 0, 0, %7"
      : "=&v"(r.w[6]), "=&v"(r.w[7]), "=&v"(r.w[8]), "=&v"(r.w[9]), "=&v"(r.w[10]), "=&v"(r.w[11]), "=&v"(m), "+s"(c)
      : "v"(a.w[6]), "v"(a.w[7]), "v"(a.w[8]), "v"(a.w[9]), "v"(a.w[10]), "v"(a.w[11]), "v"(b.w[6]), "v"(b.w[7]), "v"(b.w[8]),
        "v"(b.w[9]), "v"(b.w[10]), "v"(b.w[11]));
  return m;
}

// r = a + b over 384 bits (carry out dropped)
__device__ __forceinline__ void pk_add(Pk& r, const Pk& a, const Pk& b) {
  uint64_t c;
  asm volatile(
      "v_add_co_u32 %0, %6, %7, %13\n\t"
      "v_addc_co_u32 %1, %6, %8, %14, %6\n\t"
      "v_addc_co_u32 %2, %6, %9, %15, %6\n\t"
      "v_addc_co_u32 %3, %6, %10, %16, %6\n\t"
      "v_addc_co_u32 %4, %6, %11, %17, %6\n\t"
      "v_addc_co_u32 %5, %6, %12, %18, %6"
      : "=&v"(r.w[0]), "=&v"(r.w[1]), "=&v"(r.w[2]), "=&v"(r.w[3]), "=&v"(r.w[4]), "=&v"(r.w[5]), "=&s"(c)
      : "v"(a.w[0]), "v"(a.w[1]), "v"(a.w[2]), "v"(a.w[3]), "v"(a.w[4]), "v"(a.w[5]), "v"(b.w[0]), "v"(b.w[1]), "v"(b.w[2]),
        "v"(b.w[3]), "v"(b.w[4]), "v"(b.w[5]));
  asm volatile(
      "v_addc_co_u32 %0, %6, %7, %13, %6\n\t"
      "v_addc_co_u32 %1, %6, %8, %14, %6\n\t"
      "v_addc_co_u32 %2, %6, %9, %15, %6\n\t"
      "v_addc_co_u32 %3, %6, %10, %16, %6\n\t"
      "v_addc_co_u32 %4, %6, %11, %17, %6\n\t"
      "v_addc_co_u32 %5, %6, %12, %18, %6"
      : "=&v"(r.w[6]), "=&v"(r.w[7]), "=&v"(r.w[8]), "=&v"(r.w[9]), "=&v"(r.w[10]), "=&v"(r.w[11]), "+s"(c)
      : "v"(a.w[6]), "v"(a.w[7]), "v"(a.w[8]), "v"(a.w[9]), "v"(a.w[10]), "v"(a.w[11]), "v"(b.w[6]), "v"(b.w[7]), "v"(b.w[8]),
        "v"(b.w[9]), "v"(b.w[10]), "v"(b.w[11]));
}

// the same chains over 256 bits
__device__ __forceinline__ uint32_t pk_sub(Pk8& r, const Pk8& a, const Pk8& b) {
  uint64_t c;
  uint32_t m;
  asm volatile(
      "v_sub_co_u32 %0, %9, %10, %18\n\t"
      "v_subb_co_u32 %1, %9, %11, %19, %9\n\t"
      "v_subb_co_u32 %2, %9, %12, %20, %9\n\t"
      "v_subb_co_u32 %3, %9, %13, %21, %9\n\t"
      "v_subb_co_u32 %4, %9, %14, %22, %9\n\t"
      "v_subb_co_u32 %5, %9, %15, %23, %9\n\t"
      "v_subb_co_u32 %6, %9, %16, %24, %9\n\t"
      "v_subb_co_u32 %7, %9, %17, %25, %9\n\t"
      "v_subb_co_u32 %8, %9, 0, 0, %9"
      : "=&v"(r.w[0]), "=&v"(r.w[1]), "=&v"(r.w[2]), "=&v"(r.w[3]), "=&v"(r.w[4]), "=&v"(r.w[5]), "=&v"(r.w[6]), "=&v"(r.w[7]), "=&v"(m),
        "=&s"(c)
      : "v"(a.w[0]), "v"(a.w[1]), "v"(a.w[2]), "v"(a.w[3]), "v"(a.w[4]), "v"(a.w[5]), "v"(a.w[6]), "v"(a.w[7]), "v"(b.w[0]), "v"(b.w[1]),
        "v"(b.w[2]), "v"(b.w[3]), "v"(b.w[4]), "v"(b.w[5]), "v"(b.w[6]), "v"(b.w[7]));
  return m;
}
__device__ __forceinline__ void pk_add(Pk8& r, const Pk8& a, const Pk8& b) {
  uint64_t c;
  asm volatile(
      "v_add_co_u32 %0, %8, %9, %17\n\t"
      "v_addc_co_u32 %1, %8, %10, %18, %8\n\t"
      "v_addc_co_u32 %2, %8, %11, %19, %8\n\t"
      "v_addc_co_u32 %3, %8, %12, %20, %8\n\t"
      "v_addc_co_u32 %4, %8, %13, %21, %8\n\t"
      "v_addc_co_u32 %5, %8, %14, %22, %8\n\t"
      "v_addc_co_u32 %6, %8, %15, %23, %8\n\t"
      "v_addc_co_u32 %7, %8, %16, %24, %8"
      : "=&v"(r.w[0]), "=&v"(r.w[1]), "=&v"(r.w[2]), "=&v"(r.w[3]), "=&v"(r.w[4]), "=&v"(r.w[5]), "=&v"(r.w[6]), "=&v"(r.w[7]), "=&s"(c)
      : "v"(a.w[0]), "v"(a.w[1]), "v"(a.w[2]), "v"(a.w[3]), "v"(a.w[4]), "v"(a.w[5]), "v"(a.w[6]), "v"(a.w[7]), "v"(b.w[0]), "v"(b.w[1]),
        "v"(b.w[2]), "v"(b.w[3]), "v"(b.w[4]), "v"(b.w[5]), "v"(b.w[6]), "v"(b.w[7]));
}

template <class C>
__device__ __forceinline__ void pk_set_p_masked(PkW<C::NW>& t, uint32_t mask) {
#pragma unroll
  for (int i = 0; i < C::NW; i++) t.w[i] = C::PW[i] & mask;
}

// r = (a - b) mod p for a in [0, p + eps), b in [0, p): result in [0, p + eps)
template <class C>
__device__ __forceinline__ void pk_sub_mod(PkW<C::NW>& r, const PkW<C::NW>& a, const PkW<C::NW>& b) {
  PkW<C::NW> d, t;
  uint32_t borrow = pk_sub(d, a, b);
  pk_set_p_masked<C>(t, borrow);
  pk_add(r, d, t);
}

// r -= p if r >= p  (r < 2p)
template <class C>
__device__ __forceinline__ void pk_cond_sub_p(PkW<C::NW>& r) {
  PkW<C::NW> t, d;
  pk_set_p_masked<C>(t, 0xFFFFFFFFu);
  uint32_t borrow = pk_sub(d, r, t);
  const bool keep = borrow != 0;
#pragma unroll
  for (int i = 0; i < C::NW; i++) r.w[i] = keep ? r.w[i] : d.w[i];
}

template <int W>
__device__ __forceinline__ bool pk_is_zero(const PkW<W>& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < W; i++) o |= a.w[i];
  return o == 0;
}

template <int W>
__device__ __forceinline__ bool pk_equal(const PkW<W>& a, const PkW<W>& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < W; i++) o |= a.w[i] ^ b.w[i];
  return o == 0;
}

template <class C>
__device__ __forceinline__ void pk_unpack(Fe<C>& r, const PkW<C::NW>& a) { fe_unpack<C>(r, a.w); }
template <class C>
__device__ __forceinline__ void pk_pack(PkW<C::NW>& r, const Fe<C>& a) { fe_pack<C>(r.w, a); }

}  // namespace msm
