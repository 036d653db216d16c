// Curve-independent kernels around the tree rounds: operand descriptors of the tail rounds (k_tail_desc), the ordering of the
// buckets for k_bucket_finish (k_finish_hist / k_finish_perm), and the rows -> planes copy of the operator tests.
// Defined in sort_kernels.hip (MSM_SORT_TU); the host translation units see declarations.
#pragma once
#include "msm_kernels.h"

namespace msm {

// ---------------------------------------------------------------------------------------------
// k_tail_desc: per tail round, the operand locations of every output element, found once by binary
// search here (thousands of resident waves hide the dependent loads) instead of twice per pair inside
// the latency-critical batch-add kernel.  desc[e] = (index of the first operand << 1) | second operand present.
// ---------------------------------------------------------------------------------------------

// Big windows have millions of buckets (23 dependent loads per output in a plain binary search, 1.4 ms per call at 2^26 /
// c = 22): the 256 consecutive outputs of a block belong to a short run of buckets, so the block's first and last lane
// search the whole table once, and every lane then searches only that run (a handful of steps on lines the block has just
// touched).
__global__ void __launch_bounds__(256) k_tail_desc(uint32_t* desc, const uint32_t* off_in, const uint32_t* off_out, uint32_t nb,
                                                   uint32_t n_out)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint32_t run[2];
  const uint32_t e0 = blockIdx.x * blockDim.x;
  const uint32_t e = e0 + threadIdx.x;
  if (threadIdx.x == 0 || threadIdx.x == blockDim.x - 1) {
    const uint32_t ee = min(threadIdx.x == 0 ? e0 : e0 + blockDim.x - 1, n_out - 1);
    uint32_t lo = 0, hi = nb;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (off_out[mid] <= ee) lo = mid; else hi = mid;
    }
    run[threadIdx.x == 0 ? 0 : 1] = lo;
  }
  __syncthreads();
  if (e >= n_out) return;
  uint32_t lo = run[0], hi = run[1] + 1;   // the bucket of e lies in [run[0], run[1]]: off_out[lo] <= e < off_out[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off_out[mid] <= e) lo = mid; else hi = mid;
  }
  uint32_t j = e - off_out[lo];
  uint32_t ia = off_in[lo] + 2 * j;
  desc[e] = (ia << 1) | ((ia + 1) < off_in[lo + 1] ? 1u : 0u);
}
#endif

// point rows (k_points_from_wire) -> tree planes, element e of the planes = row e: test input of the plane-reading modes
template <int W>   // W = packed words per coordinate (12 or 8)
__global__ void __launch_bounds__(256) k_test_rows_to_planes(uint4* planes, uint64_t cap, const uint32_t* rows, uint32_t n) {
  uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  uint32_t w[W];
  load_words12(w, rows + (uint64_t)e * ROW_WORDS);
  store_planes3(planes, cap, 0, e, w);
  load_words12(w, rows + (uint64_t)e * ROW_WORDS + W);
  store_planes3(planes, cap, W / 4, e, w);
}

// ---------------------------------------------------------------------------------------------
// k_finish_hist / k_finish_perm: order the buckets by the number of elements they still hold when the tree stops,
// largest first, so that the 64 lanes of a k_bucket_finish wave run the same number of additions (a wave costs its
// longest lane: with counts of 3..7 in natural order that is ~1.5x the mean).  Counting sort over <= 64 distinct
// counts, block-private in LDS: one global atomic per distinct count per 1024 buckets.
// ---------------------------------------------------------------------------------------------

constexpr int FINISH_BINS = 64;

MSM_DEV uint32_t finish_bin(const uint32_t* off, uint32_t b) {
  uint32_t c = off[b + 1] - off[b];
  return c < FINISH_BINS ? c : FINISH_BINS - 1;
}

constexpr int FINISH_THREADS = 1024;

__global__ void __launch_bounds__(FINISH_THREADS) k_finish_hist(const uint32_t* off, uint32_t nb, uint32_t* hist)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint32_t lh[FINISH_BINS];
  if (threadIdx.x < FINISH_BINS) lh[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) atomicAdd(&lh[finish_bin(off, b)], 1u);
  __syncthreads();
  if (threadIdx.x < FINISH_BINS && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}
#endif

// cursor: FINISH_BINS zeroed words; perm[j] = j-th bucket in descending order of count
__global__ void __launch_bounds__(FINISH_THREADS) k_finish_perm(const uint32_t* off, uint32_t nb, const uint32_t* hist,
                                                                uint32_t* cursor, uint32_t* perm)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint32_t lcnt[FINISH_BINS], lbase[FINISH_BINS];
  if (threadIdx.x < FINISH_BINS) lcnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = b < nb;
  const uint32_t key = active ? finish_bin(off, b) : 0u;
  uint32_t rank = 0;
  if (active) rank = atomicAdd(&lcnt[key], 1u);
  __syncthreads();
  if (threadIdx.x < FINISH_BINS && lcnt[threadIdx.x]) {
    uint32_t base = 0;
    for (uint32_t k = FINISH_BINS - 1; k > threadIdx.x; k--) base += hist[k];   // larger counts first
    lbase[threadIdx.x] = base + atomicAdd(&cursor[threadIdx.x], lcnt[threadIdx.x]);
  }
  __syncthreads();
  if (active) perm[lbase[key] + rank] = b;
}
#endif

}  // namespace msm
