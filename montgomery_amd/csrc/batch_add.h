// k_batch_add: one round of the bucket accumulation tree -- batched-affine pair additions with one shared field
// inversion per lane (Montgomery's trick).  Reference: batchAddNew / batchAddUnsafeNew, src/curve-affine.ts:376-522
// (denominators and prefix products :484-505, shared inversion src/wasm/inverse.ts:220-271, back-substitution
// :506-516), addAffine src/wasm/curve.ts:32-58, Affine.double src/curve-affine.ts:90-109; the round structure is
// src/msm-batched-affine.ts:243-282.
//
//   output element e = input element 2e + input element 2e+1   (affine; every edge case handled)
//   lane t owns the pairs e = t + i*T, i < steps, and shares ONE fe_inv among them:
//     forward sweep : pre_i = den_0 ... den_(i-1) to the scratch, acc *= den_i           (1 M per pair)
//     backward sweep: d = inv * pre_i (= 1 / den_i), inv *= den_i, m = num * d, x3 = m^2 - x1 - x2,
//                     y3 = m (x1 - x3) - y1                                              (4 M + 1 S per pair)
//
// Shape of the code (round 2).  What the measurements say: the instruction stream alone and the
// memory traffic alone each take ~83 % of the kernel's time; the multiplier is bound by the v_mad_u64_u32 pipe and gains
// nothing from more than two resident waves (tools/ubench_mul2.hip); the chip holds 1.8-2.0 GHz under this load.  Hence
//   * two waves per SIMD, everything in registers: no register double buffer of the next pair; the next pair's x and
//     prefix product are requested three multiplications ahead into registers that are dead by then, its y after the
//     last use of y1;
//   * all lanes run the common case (two finite points, different x) straight-line; identity operands, P + P and
//     P - P are handled in wave-uniform side branches; "second operand missing" and "both missing" -- ordinary behind
//     a bucket's last element -- are selects on live registers, only "first operand is the identity" reloads;
//   * linear operations stay on packed 32-bit words with carry chains and lazily reduced results
//     (a - b + p instead of a conditional add), multiplication results are only reduced when the rare excess
//     over p actually occurs (a product is < p (1 + 2^-11));
//   * addresses are a uniform base (SALU) plus a per-lane 32-bit offset.
#pragma once
#include "curve.h"
#include "packed.h"

namespace msm {

enum : int { MODE_GATHER = 0, MODE_REGULAR = 1, MODE_SEARCH = 2 };

struct BatchArgs {
  const uint32_t* points;   // MODE_GATHER: point rows
  const uint32_t* slots;    // MODE_GATHER: payload slots, pair e = slots[2e], slots[2e+1]
  const uint4* in;          // MODE_REGULAR / MODE_SEARCH: input planes
  uint64_t in_cap;
  uint4* out;               // output planes
  uint64_t out_cap;
  uint32_t* scratch;        // prefix products: per step NL / 4 uint4 planes + 1 dword plane of T lanes each (52 B per pair at 13 limbs)
  uint64_t n_out;           // number of output elements
  uint32_t steps;
  const uint32_t* desc;     // MODE_SEARCH: operand descriptors from k_tail_desc
  uint64_t sstride;         // lanes per scratch plane (>= T; see ba_store_pre)
  const uint32_t* dest;     // MODE_GATHER, pairs in the sort's tile order (k_bin_pairs): pair e writes element dest[e]
  uint32_t* out_rows;       // MODE_GATHER: if set, results leave as 64-byte element records instead of planes (element k: its x record
                            // at byte 64 k, its y record out_y_off bytes further; 8-word fields: one record [x | y]); a round with
                            // slots == nullptr reads such records back through `points` (pair e = elements 2e, 2e + 1)
  uint64_t y_off;           // MODE_GATHER: bytes from the x of an operand to its y (4 NW inside a row of the point table)
  uint64_t out_y_off;       // out_rows, 12-word fields: bytes from the x record of a result to its y record
  const uint32_t* desc_b;   // MODE_SEARCH, optional: element index of the second operand of pair e (default: first operand + 1)
  uint32_t inplace;         // MODE_SEARCH: the sum replaces the FIRST operand (out == in): the in-place batched additions of the
                            // all-affine bucket reduction, src/msm-batched-affine-single-thread.ts:522-667
};

constexpr int BA_THREADS = 256;
// Phase fence: hipcc's scheduler otherwise interleaves independent multiplications of one step (inv * den with num * d,
// the unpacking of the next operands with the running product), which doubles the live accumulators and costs a
// resident wave.  Nothing crosses it.
#define BA_FENCE() __builtin_amdgcn_sched_barrier(0)
static_assert(BA_THREADS == 256, "the shared inversion is written for four waves per workgroup");
// resident waves per SIMD the register allocation is held to: 256 VGPRs per lane, everything stays in registers (three waves --
// 168 VGPRs and three coordinates parked in the LDS per step -- and more were measured slower: DESIGN.md section 5)
constexpr int BA_WAVES = 2;

// ---------------------------------------------------------------------------------------------------------------
// packed-word helpers.  gfx9 allows ONE scalar or literal operand per VALU instruction and the carry-in of a chain is
// one, so the modulus cannot ride the chains as a literal: it enters through a plain v_and with the borrow mask
// (pk_set_p_masked, packed.h).
// ---------------------------------------------------------------------------------------------------------------

// y -> p - y where `flip` (gather mode: the sign bit of the payload), else unchanged:  (y ^ m) + ((p + 1) & m).
// A zero y becomes p (congruent); such a value is reduced before it is stored or compared.
template <class F>
__device__ __forceinline__ void pk_cond_neg(PkW<F::NW>& y, bool flip) {
  const uint32_t m = flip ? 0xFFFFFFFFu : 0u;
  PkW<F::NW> t;
#pragma unroll
  for (int j = 0; j < F::NW; j++) {
    y.w[j] ^= m;
    t.w[j] = (j == 0 ? F::PW[0] + 1u : F::PW[j]) & m;   // p is odd: p + 1 changes word 0 only
  }
  pk_add(y, y, t);
}

// the (rare) excess of a Montgomery product over p: r < p (1 + 2^-11) -> [0, p).  One compare on the top word decides
// for the whole wave; the subtraction itself runs for a wave in ~30 (2^-11 x 64 lanes).
template <class F>
__device__ __forceinline__ void pk_reduce_product(PkW<F::NW>& r) {
  if (__any(r.w[F::NW - 1] >= F::PW[F::NW - 1])) pk_cond_sub_p<F>(r);
}

// ---------------------------------------------------------------------------------------------------------------
// memory access: planes ("piece" c of element e at uint4 index c * cap + e) and point rows
// ---------------------------------------------------------------------------------------------------------------

// one coordinate = W / 4 pieces of 16 bytes (3 for the 12-word fields, 2 for the 8-word ones), `stride` bytes apart, starting
// at byte `off` (32-bit, per lane) from the uniform `base`
template <int W>
__device__ __forceinline__ void ba_load3(PkW<W>& w, const char* base, uint32_t off, uint64_t stride) {
#pragma unroll
  for (int j = 0; j < W / 4; j++) {
    const uint4 v = *reinterpret_cast<const uint4*>(base + (uint64_t)j * stride + off);
    w.w[4 * j] = v.x; w.w[4 * j + 1] = v.y; w.w[4 * j + 2] = v.z; w.w[4 * j + 3] = v.w;
  }
}
template <int W>
__device__ __forceinline__ void ba_load3_wide(PkW<W>& w, const char* p) {   // per-lane 64-bit address, consecutive pieces
#pragma unroll
  for (int j = 0; j < W / 4; j++) {
    const uint4 v = reinterpret_cast<const uint4*>(p)[j];
    w.w[4 * j] = v.x; w.w[4 * j + 1] = v.y; w.w[4 * j + 2] = v.z; w.w[4 * j + 3] = v.w;
  }
}
template <int W>
__device__ __forceinline__ void ba_store3(char* base, uint32_t off, uint64_t stride, const PkW<W>& w) {
#pragma unroll
  for (int j = 0; j < W / 4; j++)
    *reinterpret_cast<uint4*>(base + (uint64_t)j * stride + off) = make_uint4(w.w[4 * j], w.w[4 * j + 1], w.w[4 * j + 2], w.w[4 * j + 3]);
}

// where the two operands of one pair live
template <int MODE>
struct PairLoc;
template <>
struct PairLoc<MODE_REGULAR> {   // elements 2e, 2e + 1: uniform base of the step + 32 t bytes
  const char* base;
  __device__ __forceinline__ bool absentA() const { return false; }
  __device__ __forceinline__ bool absentB() const { return false; }
};
template <>
struct PairLoc<MODE_SEARCH> {    // elements idx, idx + 1 (or desc_b's) from the descriptors
  uint64_t a, b;                 // byte offsets of the two operands inside a plane (b = a when the second is absent)
  bool b_absent, skip;
  __device__ __forceinline__ bool absentA() const { return skip; }
  __device__ __forceinline__ bool absentB() const { return b_absent; }
};
template <>
struct PairLoc<MODE_GATHER> {    // two point rows named by the slot payloads
  const char* pa;                // 128-byte line [x | y | pad] of the operand (row 0 when absent)
  const char* pb;
  uint32_t flags;                // 1: A absent, 2: B absent, 4: negate A, 8: negate B
  __device__ __forceinline__ bool absentA() const { return flags & 1u; }
  __device__ __forceinline__ bool absentB() const { return flags & 2u; }
};

template <int MODE>
__device__ __forceinline__ void ba_locate(PairLoc<MODE>& L, const BatchArgs& a, uint32_t i, uint64_t T, uint32_t t, bool active);

template <>
__device__ __forceinline__ void ba_locate<MODE_REGULAR>(PairLoc<MODE_REGULAR>& L, const BatchArgs& a, uint32_t i, uint64_t T, uint32_t,
                                                         bool) {
  L.base = reinterpret_cast<const char*>(a.in + 2ull * i * T);
}
template <>
__device__ __forceinline__ void ba_locate<MODE_SEARCH>(PairLoc<MODE_SEARCH>& L, const BatchArgs& a, uint32_t i, uint64_t T, uint32_t t,
                                                        bool active) {
  const uint32_t d = active ? a.desc[(uint64_t)i * T + t] : 0u;
  L.a = (uint64_t)(d >> 1) * 16;
  L.b_absent = (d & 1u) == 0;
  L.b = L.b_absent ? L.a : (a.desc_b && active) ? (uint64_t)a.desc_b[(uint64_t)i * T + t] * 16 : L.a + 16;
  L.skip = !active;
}
template <>
__device__ __forceinline__ void ba_locate<MODE_GATHER>(PairLoc<MODE_GATHER>& L, const BatchArgs& a, uint32_t i, uint64_t T, uint32_t t,
                                                        bool active) {
  uint2 pp = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
  if (active) {
    if (a.slots) pp = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(a.slots) + 8ull * i * T + 8u * t);
    else {   // element records written by the round before: element k is "entry" k of a table of 64-byte records
      const uint32_t e4 = (uint32_t)(((uint64_t)i * T + t) << 2);
      pp = make_uint2(e4, e4 + 2u);
    }
  }
  const uint32_t rs = a.slots ? 128u : 64u;   // uniform: bytes per operand (a row of the point table / one record)
  const bool aa = pp.x == 0xFFFFFFFFu, bb = pp.y == 0xFFFFFFFFu;
  // absent operands read row 0 (valid memory) and are then ignored: no divergent loads
  const uint64_t oa = aa ? 0 : (uint64_t)(pp.x >> 1) * rs;
  const uint64_t ob = bb ? 0 : (uint64_t)(pp.y >> 1) * rs;
  L.pa = reinterpret_cast<const char*>(a.points) + oa;
  L.pb = reinterpret_cast<const char*>(a.points) + ob;
  L.flags = (aa ? 1u : 0u) | (bb ? 2u : 0u) | ((pp.x & 1u) ? 4u : 0u) | ((pp.y & 1u) ? 8u : 0u);
}

template <class F, int MODE>
__device__ __forceinline__ void ba_load_x(PkW<F::NW>& x1, PkW<F::NW>& x2, const PairLoc<MODE>& L, const BatchArgs& a, uint32_t t) {
  constexpr int NP = F::NW / 4;   // 16-byte pieces per coordinate
  if constexpr (MODE == MODE_GATHER) {
    ba_load3_wide(x1, L.pa);
    ba_load3_wide(x2, L.pb);
  } else if constexpr (MODE == MODE_REGULAR) {
    ba_load3(x1, L.base, 32u * t, a.in_cap * 16);
    ba_load3(x2, L.base + 16, 32u * t, a.in_cap * 16);
  } else {
    const char* p = reinterpret_cast<const char*>(a.in) + L.a;
    const char* q = reinterpret_cast<const char*>(a.in) + L.b;
    const uint64_t s = a.in_cap * 16;
#pragma unroll
    for (int j = 0; j < NP; j++) {
      const uint4 v = *reinterpret_cast<const uint4*>(p + j * s), u = *reinterpret_cast<const uint4*>(q + j * s);
      x1.w[4 * j] = v.x; x1.w[4 * j + 1] = v.y; x1.w[4 * j + 2] = v.z; x1.w[4 * j + 3] = v.w;
      x2.w[4 * j] = u.x; x2.w[4 * j + 1] = u.y; x2.w[4 * j + 2] = u.z; x2.w[4 * j + 3] = u.w;
    }
  }
}

// y coordinates; gather mode applies the sign bit of the payload: y -> p - y (a zero y becomes p: congruent, and
// only ever stored after ba_canonical_y)
template <class F, int MODE>
__device__ __forceinline__ void ba_load_y(PkW<F::NW>& y1, PkW<F::NW>& y2, const PairLoc<MODE>& L, const BatchArgs& a, uint32_t t) {
  constexpr int NP = F::NW / 4;
  if constexpr (MODE == MODE_GATHER) {
    ba_load3_wide(y1, L.pa + a.y_off);
    ba_load3_wide(y2, L.pb + a.y_off);
    pk_cond_neg<F>(y1, L.flags & 4u);
    pk_cond_neg<F>(y2, L.flags & 8u);
  } else if constexpr (MODE == MODE_REGULAR) {
    const char* b = L.base + NP * a.in_cap * 16;
    ba_load3(y1, b, 32u * t, a.in_cap * 16);
    ba_load3(y2, b + 16, 32u * t, a.in_cap * 16);
  } else {
    const uint64_t s = a.in_cap * 16;
    const char* p = reinterpret_cast<const char*>(a.in) + L.a + NP * s;
    const char* q = reinterpret_cast<const char*>(a.in) + L.b + NP * s;
#pragma unroll
    for (int j = 0; j < NP; j++) {
      const uint4 v = *reinterpret_cast<const uint4*>(p + j * s), u = *reinterpret_cast<const uint4*>(q + j * s);
      y1.w[4 * j] = v.x; y1.w[4 * j + 1] = v.y; y1.w[4 * j + 2] = v.z; y1.w[4 * j + 3] = v.w;
      y2.w[4 * j] = u.x; y2.w[4 * j + 1] = u.y; y2.w[4 * j + 2] = u.z; y2.w[4 * j + 3] = u.w;
    }
  }
}

// The second operand of a pair once more (x and y, sign applied): only for the patch of "first operand is the identity",
// which trailing padding never produces -- it needs a cancellation P - P earlier in the tree.
template <class F, int MODE>
__device__ __forceinline__ void ba_load_b(PkW<F::NW>& x2, PkW<F::NW>& y2, const PairLoc<MODE>& L, const BatchArgs& a, uint32_t t) {
  constexpr int NP = F::NW / 4;
  if constexpr (MODE == MODE_GATHER) {
    ba_load3_wide(x2, L.pb);
    ba_load3_wide(y2, L.pb + a.y_off);
    pk_cond_neg<F>(y2, L.flags & 8u);
    pk_cond_sub_p<F>(y2);
  } else if constexpr (MODE == MODE_REGULAR) {
    ba_load3(x2, L.base + 16, 32u * t, a.in_cap * 16);
    ba_load3(y2, L.base + NP * a.in_cap * 16 + 16, 32u * t, a.in_cap * 16);
  } else {
    const uint64_t s = a.in_cap * 16;
    const char* p = reinterpret_cast<const char*>(a.in) + L.b;
#pragma unroll
    for (int j = 0; j < NP; j++) {
      const uint4 v = *reinterpret_cast<const uint4*>(p + j * s), u = *reinterpret_cast<const uint4*>(p + (j + NP) * s);
      x2.w[4 * j] = v.x; x2.w[4 * j + 1] = v.y; x2.w[4 * j + 2] = v.z; x2.w[4 * j + 3] = v.w;
      y2.w[4 * j] = u.x; y2.w[4 * j + 1] = u.y; y2.w[4 * j + 2] = u.z; y2.w[4 * j + 3] = u.w;
    }
  }
}

// prefix product of one (step, lane): N limbs (13 or 9) as N / 4 uint4 + 1 dword, each in its own plane of T lanes
template <int N>
__device__ __forceinline__ void ba_store_pre(const BatchArgs& a, uint32_t i, uint64_t, uint32_t t, const uint32_t (&l)[N]) {
  static_assert(N % 4 == 1, "limb count = whole 16-byte pieces + one dword");
  const uint64_t T = a.sstride;
  char* sb = reinterpret_cast<char*>(a.scratch) + (uint64_t)i * T * (4 * N);
#pragma unroll
  for (int j = 0; j < N / 4; j++)
    *reinterpret_cast<uint4*>(sb + (uint64_t)j * T * 16 + 16u * t) = make_uint4(l[4 * j], l[4 * j + 1], l[4 * j + 2], l[4 * j + 3]);
  *reinterpret_cast<uint32_t*>(sb + (uint64_t)(N / 4) * 16 * T + 4u * t) = l[N - 1];
}
template <int N>
__device__ __forceinline__ void ba_load_pre(uint32_t (&l)[N], const BatchArgs& a, uint32_t i, uint64_t, uint32_t t) {
  const uint64_t T = a.sstride;
  const char* sb = reinterpret_cast<const char*>(a.scratch) + (uint64_t)i * T * (4 * N);
#pragma unroll
  for (int j = 0; j < N / 4; j++) {
    const uint4 v = *reinterpret_cast<const uint4*>(sb + (uint64_t)j * T * 16 + 16u * t);
    l[4 * j] = v.x; l[4 * j + 1] = v.y; l[4 * j + 2] = v.z; l[4 * j + 3] = v.w;
  }
  l[N - 1] = *reinterpret_cast<const uint32_t*>(sb + (uint64_t)(N / 4) * 16 * T + 4u * t);
}

// ---------------------------------------------------------------------------------------------------------------
// the denominator of one pair and what kind of pair it is
// ---------------------------------------------------------------------------------------------------------------

// bit 0: output is the identity, bit 1: output is operand A, bit 2: output is operand B, bit 3: doubling, bit 4: lane idle
enum : uint32_t { BA_ZERO = 1, BA_COPY_A = 2, BA_COPY_B = 4, BA_DOUBLE = 8, BA_SKIP = 16 };

// Common case first: den = x2 - x1 + p for every lane, straight-line.  Lanes with an identity operand, an idle lane or
// equal x get den = 1 (P + P: den = 2 y) in a branch that a wave takes only if one of its lanes needs it.
// Forward and backward sweep call this with the same inputs, so both see the same den.  HAVE_Y: the caller already
// holds the pair's y coordinates (backward sweep); otherwise they are loaded in the rare equal-x case only.
template <class F, int MODE, bool HAVE_Y>
__device__ __forceinline__ uint32_t ba_denominator(Fe<F>& den, const PkW<F::NW>& x1, const PkW<F::NW>& x2, const PkW<F::NW>* y1p,
                                                    const PkW<F::NW>* y2p, const PairLoc<MODE>& L, const BatchArgs& a, uint32_t t,
                                                    bool active) {
  using Pk = PkW<F::NW>;
  constexpr int NW = F::NW;
  bool same_x_out;
  Pk dx;
  {
    Pk pm;
    const uint32_t borrow = pk_sub(dx, x2, x1);
    const bool z = pk_is_zero(dx);           // x1, x2 canonical (or the identity's all-ones): equal iff the words are
    pk_set_p_masked<F>(pm, borrow);
    pk_add(dx, dx, pm);                      // (x2 - x1) mod p
    pk_unpack<F>(den, dx);
    same_x_out = z;
  }
  const bool inf1 = L.absentA() || x1.w[NW - 1] == INF_WORD, inf2 = L.absentB() || x2.w[NW - 1] == INF_WORD;
  const bool same_x = same_x_out;
  uint32_t kind = 0;
  if (__any(!active || inf1 || inf2 || same_x)) {
    if (!active) kind = BA_SKIP | BA_ZERO;
    else if (inf1 && inf2) kind = BA_ZERO;
    else if (inf2) kind = BA_COPY_A;
    else if (inf1) kind = BA_COPY_B;
    else if (same_x) {
      Pk y1, y2;
      if (HAVE_Y) { y1 = *y1p; y2 = *y2p; }
      else ba_load_y<F, MODE>(y1, y2, L, a, t);
      pk_cond_sub_p<F>(y1);   // gather mode: a negated zero is p
      pk_cond_sub_p<F>(y2);
      kind = (pk_equal(y1, y2) && !pk_is_zero(y1)) ? BA_DOUBLE : BA_ZERO;
      if (kind == BA_DOUBLE) {
        Fe<F> yl;
        pk_unpack<F>(yl, y1);
        fe_add<F>(den, yl, yl);
      }
    }
    if (kind & ~BA_DOUBLE) fe_set_one<F>(den);
  }
  return kind;
}

// ---------------------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------------------

template <class CV, int MODE>
__global__ void __launch_bounds__(BA_THREADS, BA_WAVES) k_batch_add(BatchArgs a) {
  using F = typename CV::F;
  using Pk = PkW<F::NW>;
  constexpr int NW = F::NW, NP = F::NW / 4;   // packed words and 16-byte pieces per coordinate
  const uint64_t T = (uint64_t)gridDim.x * BA_THREADS;
  const uint32_t t = blockIdx.x * BA_THREADS + threadIdx.x;
  const uint32_t steps = a.steps;
  // lane t owns the pairs e = t + i * T; all lanes walk all `steps` steps (a uniform loop: the step's bases stay in
  // SGPRs), lanes past the end of the round idle through it with den = 1 and no stores
  auto is_active = [&](uint32_t i) { return (uint64_t)i * T + t < a.n_out; };

  Fe<F> acc;
  fe_set_one<F>(acc);

  // ---- forward sweep: prefix products of the denominators ---------------------------------------
  {
    PairLoc<MODE> L;
    Pk x1, x2;
    ba_locate<MODE>(L, a, 0, T, t, is_active(0));
    ba_load_x<F, MODE>(x1, x2, L, a, t);
#pragma unroll 1
    for (uint32_t i = 0; i < steps; i++) {
      Fe<F> den;
      ba_denominator<F, MODE, false>(den, x1, x2, nullptr, nullptr, L, a, t, is_active(i));
      if (i + 1 < steps) {   // the next pair's x: its registers are free now, the multiplication covers the latency
        ba_locate<MODE>(L, a, i + 1, T, t, is_active(i + 1));
        ba_load_x<F, MODE>(x1, x2, L, a, t);
      }
      ba_store_pre(a, i, T, t, acc.l);
      BA_FENCE();
      fe_mul<F>(acc, acc, den);
      BA_FENCE();
    }
  }

  Fe<F> inv;
  {
    // One inversion per WORKGROUP and lane column instead of one per wave.  An inversion costs ~13 pair additions' worth of
    // instructions and every lane paid one per launch: a quarter of the tree at 2^20, where a lane's chain is 28 - 56 pairs,
    // 3.6 % of the multiply-adds at 2^26.  The four waves publish their running products in the LDS; every wave multiplies the
    // other three of its lane column (2 products), wave 0 also its own and inverts the product of all four, and every wave
    // gets its inverse with one more product: three multiplications of latency around the one inversion, during which the
    // other three waves leave their issue slots to the other workgroup of the CU.  Measured (profiles/r05_experiments.txt
    // item 13): 2^20 3.40 -> 3.33 ms, 2^22 10.86 -> 10.67, 2^23 19.88 -> 19.56, 2^26 133.9 -> 132.1; with wave 0 doing all nine
    // products of the serial form alone: 3.36 / 10.73 / 19.54 / 132.0.
    constexpr int NLI = F::NL;
    __shared__ uint32_t xch[NLI * BA_THREADS];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (int l = 0; l < NLI; l++) xch[l * BA_THREADS + threadIdx.x] = acc.l[l];
    __syncthreads();
    Fe<F> e;
    {
      Fe<F> b0, b1, b2;
#pragma unroll
      for (int l = 0; l < NLI; l++) {
        b0.l[l] = xch[l * BA_THREADS + ((wave + 1) & 3u) * 64 + lane];
        b1.l[l] = xch[l * BA_THREADS + ((wave + 2) & 3u) * 64 + lane];
        b2.l[l] = xch[l * BA_THREADS + ((wave + 3) & 3u) * 64 + lane];
      }
      fe_mul<F>(e, b0, b1);
      fe_mul<F>(e, e, b2);             // the product of the other three waves' running products
    }
    __syncthreads();                   // (everyone has read the running products: the buffer may take the inverse)
    if (wave == 0) {
      Fe<F> t, ti;
      fe_mul<F>(t, e, acc);
      fe_inv<F>(ti, t);                // 1 / (a0 a1 a2 a3)
#pragma unroll
      for (int l = 0; l < NLI; l++) xch[l * BA_THREADS + lane] = ti.l[l];
    }
    __syncthreads();
    {
      Fe<F> ti;
#pragma unroll
      for (int l = 0; l < NLI; l++) ti.l[l] = xch[l * BA_THREADS + lane];
      fe_mul<F>(inv, ti, e);
    }
  }

  // ---- backward sweep ------------------------------------------------------------------------
  // Order inside one step (what is live where decides the register allocation, hence the resident waves):
  //   d = inv * pre | den, num (all rare cases here, while little else is live) |
  //   inv *= den | next x and prefix product requested | m = num * d | m^2 | x3 = m^2 - x1 - x2 |
  //   y3 = m (x1 - x3) - y1 | loads of the next pair's y | stores
  // Everything stays in registers at two waves per SIMD (256 VGPRs per lane).
  {
    PairLoc<MODE> L, Ln;
    Pk x1, x2, y1, y2, nx1, nx2;
    Fe<F> pre, npre;
    ba_locate<MODE>(L, a, steps - 1, T, t, is_active(steps - 1));
    ba_load_x<F, MODE>(x1, x2, L, a, t);
    ba_load_pre(pre.l, a, steps - 1, T, t);
    ba_load_y<F, MODE>(y1, y2, L, a, t);
    Ln = L;
#pragma unroll 1
    for (int i = (int)steps - 1; i >= 0; i--) {
      const bool active = is_active((uint32_t)i);
      const uint64_t e_cur = (uint64_t)i * T + t;
      uint32_t o_local = 0;
      if (MODE == MODE_GATHER && a.dest && active) o_local = a.dest[e_cur];
      Fe<F> d, den, num;
      fe_mul<F>(d, inv, pre);                         // 1 / den_i
      BA_FENCE();
      const uint32_t kind = ba_denominator<F, MODE, true>(den, x1, x2, &y1, &y2, L, a, t, active);
      {
        Pk dy;
        pk_sub_mod<F>(dy, y2, y1);                    // (y2 - y1) mod p (gather mode may hold y = p: the result stays in [0, p])
        pk_unpack<F>(num, dy);
      }
      if (__any(kind & BA_DOUBLE)) {                  // P + P: num = 3 x^2
        if (kind & BA_DOUBLE) {
          Fe<F> xl, sq;
          pk_unpack<F>(xl, x1);
          fe_mul<F>(sq, xl, xl);
          fe_add<F>(num, sq, sq);
          fe_add<F>(num, num, sq);
        }
      }
      BA_FENCE();
      fe_mul<F>(inv, inv, den);                       // strip den_i from the running inverse
      BA_FENCE();
      // The next pair's x and prefix product are requested here, three multiplications before they are needed:
      // 37 registers that are free from now on (the widest point of the step, inv * den with d and num waiting, is behind)
      if (i > 0) {
        ba_locate<MODE>(Ln, a, (uint32_t)i - 1, T, t, is_active((uint32_t)i - 1));
        ba_load_x<F, MODE>(nx1, nx2, Ln, a, t);
        ba_load_pre(npre.l, a, (uint32_t)i - 1, T, t);
      }
      BA_FENCE();
      Fe<F> m;
      fe_mul<F>(m, num, d);
      BA_FENCE();
      Pk x3, y3;
      {
        Fe<F> mm;
        fe_sqr<F>(mm, m);
        pk_pack<F>(x3, mm);                           // < p (1 + 2^-11)
      }
      BA_FENCE();
      pk_sub_mod<F>(x3, x3, x1);                      // m^2 - x1 - x2, every step in [0, p (1 + 2^-11))
      pk_sub_mod<F>(x3, x3, x2);
      pk_reduce_product<F>(x3);
      {
        Pk tw;
        Fe<F> tt, y3l;
        pk_sub_mod<F>(tw, x1, x3);
        pk_unpack<F>(tt, tw);
        BA_FENCE();
        fe_mul<F>(y3l, m, tt);
        pk_pack<F>(y3, y3l);
      }
      BA_FENCE();
      // y3 = m (x1 - x3) - y1; gather mode may hold y1 = p (a negated zero): reduce it first in that rare case
      if (MODE == MODE_GATHER) pk_reduce_product<F>(y1);
      pk_sub_mod<F>(y3, y3, y1);
      pk_reduce_product<F>(y3);
      // Patches.  Padding behind a bucket's last element makes "second operand missing" (copy A) and "both missing"
      // (identity) ordinary at large windows, where a wave spans several buckets: both are selects on registers that
      // are still at hand.  Only "first operand is the identity" fetches the second operand again.
      if (__any(kind & ~BA_DOUBLE)) {
        const bool ca = kind & BA_COPY_A, cb = kind & BA_COPY_B, z = kind & BA_ZERO;
#pragma unroll
        for (int j = 0; j < NW; j++) {
          x3.w[j] = z ? INF_WORD : ca ? x1.w[j] : x3.w[j];
          y3.w[j] = z ? 0u : ca ? y1.w[j] : y3.w[j];
        }
        if (__any(cb)) {
          Pk bx, by;
          ba_load_b<F, MODE>(bx, by, L, a, t);
#pragma unroll
          for (int j = 0; j < NW; j++) {
            x3.w[j] = cb ? bx.w[j] : x3.w[j];
            y3.w[j] = cb ? by.w[j] : y3.w[j];
          }
        }
      }
      asm volatile("" ::: "memory");
      BA_FENCE();
      if (i > 0) ba_load_y<F, MODE>(y1, y2, Ln, a, t);   // the next pair's y: in flight during the stores and d = inv * pre
      if (!(kind & BA_SKIP)) {
        if (MODE == MODE_GATHER && a.out_rows) {
          // Uniform: round 1 over the sort's tile-ordered pairs.  The pair's element index comes from the table, and the element leaves as whole
          // 64-byte records: a permuted store of 16-byte plane pieces leaves every line partly written by several workgroups (on
          // different XCDs), and the memory side then reads, merges and rewrites each of them (measured: round 1 68 -> 107 ms);
          // aligned 64-byte pieces go through.  12-word fields: an x record [x | 16 bytes] and, a.out_y_off bytes further, a y record
          // -- the two x of the next round's pair (2k, 2k + 1) are then one 128-byte line, and its forward sweep touches nothing
          // else.  8-word fields: one record [x | y].
          const uint64_t o = a.dest ? (uint64_t)o_local : e_cur;
          uint4* rx = reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.out_rows) + o * 64);
          if constexpr (NP == 3) {
            uint4* ry = reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.out_rows) + a.out_y_off + o * 64);
#pragma unroll
            for (int j = 0; j < 3; j++) {
              rx[j] = make_uint4(x3.w[4 * j], x3.w[4 * j + 1], x3.w[4 * j + 2], x3.w[4 * j + 3]);
              ry[j] = make_uint4(y3.w[4 * j], y3.w[4 * j + 1], y3.w[4 * j + 2], y3.w[4 * j + 3]);
            }
            rx[3] = make_uint4(0, 0, 0, 0);
            ry[3] = make_uint4(0, 0, 0, 0);
          } else {
            static_assert(NP == 2 || NP == 3, "records are laid out for 8- and 12-word coordinates");
#pragma unroll
            for (int j = 0; j < 2; j++) {
              rx[j] = make_uint4(x3.w[4 * j], x3.w[4 * j + 1], x3.w[4 * j + 2], x3.w[4 * j + 3]);
              rx[2 + j] = make_uint4(y3.w[4 * j], y3.w[4 * j + 1], y3.w[4 * j + 2], y3.w[4 * j + 3]);
            }
          }
        } else if (MODE == MODE_SEARCH && a.inplace) {   // uniform: the sum replaces the first operand
          if constexpr (MODE == MODE_SEARCH) {
            char* ob = reinterpret_cast<char*>(a.out) + L.a;
            ba_store3(ob, 0u, a.out_cap * 16, x3);
            ba_store3(ob + NP * a.out_cap * 16, 0u, a.out_cap * 16, y3);
          }
        } else
        {
        char* ob = reinterpret_cast<char*>(a.out + (uint64_t)i * T);
        ba_store3(ob, 16u * t, a.out_cap * 16, x3);
        ba_store3(ob + NP * a.out_cap * 16, 16u * t, a.out_cap * 16, y3);
        }
      }
      x1 = nx1; x2 = nx2; pre = npre; L = Ln;
    }
  }
}

}  // namespace msm
