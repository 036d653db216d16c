// GLV scalar decomposition and signed-window slicing on device, one scalar per lane.
//
//   decompose          src/wasm/glv.ts:68-169 (`glvGeneral.decompose`), constants :35-63
//   multiplyMsb        src/wasm/glv.ts:187-214 (round-half-up of the top half of a product)
//   extractBitSlice    src/wasm/field-helpers.ts:307-358
//   signed recoding    src/msm-batched-affine.ts:183-193
//
// Arithmetic is on little-endian 32-bit words with 64-bit accumulation; these routines run once
// per scalar (a few hundred VALU ops) and are nowhere near the hot loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "constants_gen.h"

namespace msm {

#ifndef MSM_DEV
#define MSM_DEV __host__ __device__ __forceinline__
#endif

// r[0..NA+NB) = a * b
template <int NA, int NB>
MSM_DEV void bn_mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
#pragma unroll
  for (int i = 0; i < NA + NB; i++) r[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      c += (uint64_t)a[i] * b[j] + r[i + j];
      r[i + j] = (uint32_t)c;
      c >>= 32;
    }
    r[i + NB] = (uint32_t)c;
  }
}

// two's complement r += sign * a over N words (a has NA <= N words, zero extended)
template <int N, int NA>
MSM_DEV void bn_addsub(uint32_t* r, const uint32_t* a, bool subtract) {
  uint64_t c = subtract ? 1 : 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint32_t ai = i < NA ? a[i] : 0u;
    if (subtract) ai = ~ai;
    c += (uint64_t)r[i] + ai;
    r[i] = (uint32_t)c;
    c >>= 32;
  }
}

template <int N>
MSM_DEV void bn_negate(uint32_t* r) {
  uint64_t c = 1;
#pragma unroll
  for (int i = 0; i < N; i++) {
    c += (uint64_t)(~r[i]);
    r[i] = (uint32_t)c;
    c >>= 32;
  }
}

// (x >> shift) truncated to NR words, x has NX words
template <int NR, int NX>
MSM_DEV void bn_shr(uint32_t* r, const uint32_t* x, int shift) {
  const int ws = shift / 32, bs = shift % 32;
#pragma unroll
  for (int i = 0; i < NR; i++) {
    uint32_t lo = (i + ws) < NX ? x[i + ws] : 0u;
    uint32_t hi = (i + ws + 1) < NX ? x[i + ws + 1] : 0u;
    r[i] = bs == 0 ? lo : ((lo >> bs) | (hi << (32 - bs)));
  }
}

struct GlvHalf {
  uint32_t mag[4];  // |s_j| < 2^126
  bool neg;
};

// s (8 words, < q) -> s0 + s1 * lambda, as sign + magnitude each
template <class G>
MSM_DEV void glv_decompose(GlvHalf& h0, GlvHalf& h1, const uint32_t (&s)[8]) {
  // s_hi = s >> k  (k = 116): 253 - 116 = 137 bits -> 5 words
  uint32_t shi[5];
  bn_shr<5, 8>(shi, s, G::K_SHIFT);
  // X_j = round(|m_j| * s_hi / 2^m), m = 145: product has 10 words
  uint32_t prod[10], X0[5], X1[5];
  uint32_t cM0[5], cM1[5], cV00[5], cV01[5], cV10[5], cV11[5];
#pragma unroll
  for (int i = 0; i < 5; i++) {
    cM0[i] = G::M0[i]; cM1[i] = G::M1[i];
    cV00[i] = G::V00[i]; cV01[i] = G::V01[i]; cV10[i] = G::V10[i]; cV11[i] = G::V11[i];
  }
  {
    bn_mul<5, 5>(prod, shi, cM0);
    bn_shr<5, 10>(X0, prod, G::M_SHIFT);
    uint32_t rb = (prod[(G::M_SHIFT - 1) / 32] >> ((G::M_SHIFT - 1) % 32)) & 1u;
    uint32_t one[1] = {rb};
    bn_addsub<5, 1>(X0, one, false);
    bn_mul<5, 5>(prod, shi, cM1);
    bn_shr<5, 10>(X1, prod, G::M_SHIFT);
    rb = (prod[(G::M_SHIFT - 1) / 32] >> ((G::M_SHIFT - 1) % 32)) & 1u;
    one[0] = rb;
    bn_addsub<5, 1>(X1, one, false);
  }
  // x_j = sign(m_j) * X_j;  s0 = s + v00 x0 + v01 x1;  s1 = v10 x0 + v11 x1   (two's complement, 10 words)
  uint32_t acc[10], term[10];
#pragma unroll
  for (int i = 0; i < 10; i++) acc[i] = i < 8 ? s[i] : 0u;
  bn_mul<5, 5>(term, X0, cV00);
  bn_addsub<10, 10>(acc, term, (G::M0_NEG ^ G::V00_NEG) != 0);
  bn_mul<5, 5>(term, X1, cV01);
  bn_addsub<10, 10>(acc, term, (G::M1_NEG ^ G::V01_NEG) != 0);
  h0.neg = (acc[9] >> 31) != 0;
  if (h0.neg) bn_negate<10>(acc);
#pragma unroll
  for (int i = 0; i < 4; i++) h0.mag[i] = acc[i];

#pragma unroll
  for (int i = 0; i < 10; i++) acc[i] = 0u;
  bn_mul<5, 5>(term, X0, cV10);
  bn_addsub<10, 10>(acc, term, (G::M0_NEG ^ G::V10_NEG) != 0);
  bn_mul<5, 5>(term, X1, cV11);
  bn_addsub<10, 10>(acc, term, (G::M1_NEG ^ G::V11_NEG) != 0);
  h1.neg = (acc[9] >> 31) != 0;
  if (h1.neg) bn_negate<10>(acc);
#pragma unroll
  for (int i = 0; i < 4; i++) h1.mag[i] = acc[i];
}

// bits [start, start + len) of a little-endian word array (len <= 31)
template <int NX>
MSM_DEV uint32_t bn_bits(const uint32_t* x, int start, int len) {
  int wi = start >> 5, bs = start & 31;
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int i = 0; i < NX; i++) {
    if (i == wi) lo = x[i];
    if (i == wi + 1) hi = x[i];
  }
  uint64_t v = ((uint64_t)hi << 32) | lo;
  return (uint32_t)(v >> bs) & ((1u << len) - 1u);
}

// The windows of a scalar one after the other: take the low `len` bits (len <= 31) and shift the words down by them -- one
// funnel shift per word and window, where bn_bits at a running position selects its two words with a compare and a
// conditional move per word (the digit loop was three quarters of k_digits' instructions)
template <int NX>
MSM_DEV uint32_t bn_take_bits(uint32_t (&x)[NX], int len) {
  const uint32_t v = x[0] & ((1u << len) - 1u);
#pragma unroll
  for (int i = 0; i + 1 < NX; i++) x[i] = (x[i] >> len) | (x[i + 1] << (32 - len));   // (len >= 1: window sizes start at 2)
  x[NX - 1] >>= len;
  return v;
}

}  // namespace msm
