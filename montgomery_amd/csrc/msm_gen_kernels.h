// Device side of the synthetic input generators (see msm_gen.h): counter-based splitmix64 stream, masked
// rejection sampling of scalars (src/curve-random.ts:151-194), table-sum points (src/curve-random.ts:14-92).
#pragma once
#include "msm_kernels.h"

namespace msm_gen {

constexpr int N_BASIS = 5;
constexpr int TBL_BITS = 13;   // 5 tables of 2^13 multiples: the reference's defaults, 65 bits of entropy per point (src/curve-random.ts:14-92)
constexpr int TBL = 1 << TBL_BITS;

__host__ __device__ inline uint64_t mix64(uint64_t seed, uint64_t ctr) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// ---- scalars ---------------------------------------------------------------------------------

__host__ __device__ inline bool ge_q(const uint64_t* s, const uint64_t* q) {
  for (int i = 3; i >= 0; i--) {
    if (s[i] > q[i]) return true;
    if (s[i] < q[i]) return false;
  }
  return true;
}

__host__ __device__ inline void draw_scalar(uint64_t* s, uint64_t seed, uint64_t i, const uint64_t* q, int q_bits) {
  for (uint64_t attempt = 0; attempt < 256; attempt++) {
    for (int j = 0; j < 4; j++) s[j] = mix64(seed ^ 0x5ca1ab1e00000000ull, (i * 256 + attempt) * 4 + j);
    s[3] &= (1ull << (q_bits - 192)) - 1;  // mask to the bit length of q, then reject
    if (!ge_q(s, q)) return;
  }
  s[0] = 1; s[1] = s[2] = s[3] = 0;
}

struct Q256 {
  uint64_t v[4];
  int bits;
};

// ---- points ----------------------------------------------------------------------------------

// index of point i into table j: bits [13 j, 13 j + 13) of a 128-bit draw (two words of the stream)
__host__ __device__ inline uint32_t table_index(uint64_t seed, uint64_t i, int j) {
  const uint64_t lo = mix64(seed ^ 0x90117500000000ull, 2 * i), hi = mix64(seed ^ 0x90117500000000ull, 2 * i + 1);
  const int sh = TBL_BITS * j;
  const uint64_t v = sh < 64 ? (lo >> sh) | (sh && sh + TBL_BITS > 64 ? hi << (64 - sh) : 0) : hi >> (sh - 64);
  return (uint32_t)v & (TBL - 1);
}

// tables: N_BASIS * TBL point rows (x, y used); rows_out: n point rows
template <class CV>
__global__ void __launch_bounds__(256) k_gen_points(uint32_t* rows_out, const uint32_t* tables, uint64_t n, uint64_t seed) {
  using namespace msm;
  using F = typename CV::F;
  constexpr int NL = F::NL;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Proj<F> acc;
  proj_set_zero<F>(acc);
#pragma unroll 1
  for (int j = 0; j < N_BASIS; j++) {
    uint32_t t = table_index(seed, i, j);
    const uint32_t* row = tables + ((uint64_t)j * TBL + t) * ROW_WORDS;
    Proj<F> Q;
    fe_load<F>(Q.X, row);
    fe_load<F>(Q.Y, row + F::NW);
    proj_add_mixed<F>(acc, acc, Q, false);
  }
  uint32_t* out = rows_out + i * ROW_WORDS;
  if (proj_is_zero<F>(acc)) {
    store_row_identity<F::NW / 4>(out);
    return;
  }
  Fe<F> zi, x, y, bx, beta;
  fe_inv<F>(zi, acc.Z);
  fe_mul<F>(x, acc.X, zi);
  fe_mul<F>(y, acc.Y, zi);
  fe_reduce_2p<F>(x);
  fe_reduce_2p<F>(y);
#pragma unroll
  for (int l = 0; l < NL; l++) beta.l[l] = F::BETAL[l];
  fe_mul<F>(bx, x, beta);
  fe_reduce_2p<F>(bx);
  store_row(out, x, y, bx);
}

}  // namespace msm_gen
