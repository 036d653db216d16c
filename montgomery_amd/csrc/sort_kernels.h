// Curve-independent kernels of the MSM pipeline: bucket-size scans, the LDS-privatised counting sort (one level for small
// inputs, LDS-staged two-pass split for c <= 16, three-pass split up to the largest accepted window c = 24) and the operand
// descriptors of the tail rounds (tree_kernels.h).
// (reference phases: integrateBucketCounts src/msm-batched-affine.ts:423-447, sortPoints :456-502)
// Defined in sort_kernels.hip (MSM_SORT_TU); the host translation units see declarations.  The curve-templated kernels live in
// msm_kernels.h, the kernels around the tree rounds in tree_kernels.h.
#pragma once
#include "msm_kernels.h"

namespace msm {

// ---------------------------------------------------------------------------------------------
// scans: bucket sizes -> padded slot offsets, cursor, tail-round offsets (k_pscan_* below)
//   info[0] = total slots, info[1] = max bucket size,
//   info[3 + r] = number of elements entering tail round r
//   tail_off[r] has nb + 1 entries: offsets of ceil(ceil(n/G) / 2^r)
// ---------------------------------------------------------------------------------------------

constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_ITEMS = 4;

MSM_DEV uint32_t block_excl_scan(uint32_t v, uint32_t* lds_wave, uint32_t& total) {
  // inclusive scan inside the wave
  uint32_t x = v;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  __syncthreads();
  if (lane == 63) lds_wave[wave] = x;
  __syncthreads();
  uint32_t wave_base = 0, tot = 0;
  const int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; w++) {
    uint32_t t = lds_wave[w];
    if (w < wave) wave_base += t;
    tot += t;
  }
  total = tot;
  return wave_base + x - v;
}

// ---------------------------------------------------------------------------------------------
// Multi-block scan of the bucket sizes (replaces the single-workgroup k_scan on the LDS-sort path):
//   quantity 0      : padded slot count  roundup(n, G)          -> cursor (slot offsets), total -> info[0]
//   quantity 1 + r  : ceil(ceil(n / G) / 2^r), r = 0..RT          -> tail_off[r] (nb + 1 entries), totals -> info[3 + r]
// k_pscan_partial sums each quantity per block of PS_BLOCK * PS_ITEMS buckets, k_pscan_top scans the
// block sums (one workgroup), k_pscan_final rescans each block with its base.  RT follows from the largest bucket, which
// k_bucket_max leaves in info[1]: the scan kernels take it from there (pscan_nq), so the host reads the largest bucket back
// together with the totals, in one read-back behind k_pscan_final, and sizes the scan for the largest RT there can be.
// ---------------------------------------------------------------------------------------------

constexpr int PS_BLOCK = 256;
constexpr int PS_ITEMS = 16;
constexpr int PS_SPAN = PS_BLOCK * PS_ITEMS;

MSM_DEV uint32_t scan_quantity(uint32_t n, uint32_t logG, int q) {
  const uint32_t cg = (n + ((1u << logG) - 1)) >> logG;
  if (q == 0) return cg << logG;
  const int r = q - 1;
  return (cg + ((1u << r) - 1)) >> r;
}

constexpr int PS_MAX_NQ = 34;   // RT <= 32
// number of scanned quantities = RT + 2 with 2^RT >= ceil(largest bucket / G)   (the host repeats this after its read-back)
MSM_DEV int pscan_nq(const uint32_t* info, uint32_t logG) {
  const uint32_t capmax = (info[1] + ((1u << logG) - 1)) >> logG;
  int rt = 0;
  while (rt < 32 && (1u << rt) < capmax) rt++;
  return rt + 2;
}

// info[1] = the largest bucket; info[40..41] (one 64-bit counter) = sum over the non-empty buckets of (size - 1): the pair
// additions the bucket sums need whatever the tree looks like (msm_result.n_pairs_algo; the tree also issues the additions
// of its padding lanes, msm_result.n_pairs)
constexpr int INFO_ALGO_PAIRS = 40;
__global__ void __launch_bounds__(256) k_bucket_max(const uint32_t* counts, uint32_t nb, uint32_t* info)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint32_t lds_max;
  __shared__ unsigned long long lds_sum;
  if (threadIdx.x == 0) { lds_max = 0; lds_sum = 0; }
  __syncthreads();
  uint32_t mx = 0;
  unsigned long long sum = 0;
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
    const uint32_t cnt = counts[b];
    mx = max(mx, cnt);
    sum += cnt ? cnt - 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
  atomicMax(&lds_max, mx);
  if ((threadIdx.x & 63) == 0) atomicAdd(&lds_sum, sum);
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(&info[1], lds_max);
    atomicAdd(reinterpret_cast<unsigned long long*>(info + INFO_ALGO_PAIRS), lds_sum);
  }
}
#endif

__global__ void __launch_bounds__(PS_BLOCK) k_pscan_partial(const uint32_t* counts, uint32_t nb, uint32_t logG, const uint32_t* info,
                                                            uint32_t* partial, uint32_t nblocks)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint32_t lds_wave[PS_BLOCK / 64];
  const int nq = pscan_nq(info, logG);
  const uint32_t b0 = blockIdx.x * PS_SPAN + threadIdx.x * PS_ITEMS;
  uint32_t n[PS_ITEMS];
#pragma unroll
  for (int j = 0; j < PS_ITEMS; j++) n[j] = (b0 + j) < nb ? counts[b0 + j] : 0u;
  for (int q = 0; q < nq; q++) {
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < PS_ITEMS; j++) sum += scan_quantity(n[j], logG, q);
    uint32_t tot;
    block_excl_scan(sum, lds_wave, tot);
    if (threadIdx.x == 0) partial[(uint64_t)q * nblocks + blockIdx.x] = tot;
  }
}
#endif

__global__ void __launch_bounds__(SCAN_THREADS) k_pscan_top(uint32_t* partial, uint32_t nblocks, uint32_t logG, uint32_t* info)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint32_t lds_wave[SCAN_THREADS / 64];
  const int nq = pscan_nq(info, logG);
  for (int q = 0; q < nq; q++) {
    uint32_t* p = partial + (uint64_t)q * nblocks;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += SCAN_THREADS) {
      uint32_t i = base + threadIdx.x;
      uint32_t v = i < nblocks ? p[i] : 0u;
      uint32_t tot;
      uint32_t ex = block_excl_scan(v, lds_wave, tot) + carry;
      if (i < nblocks) p[i] = ex;
      carry += tot;
    }
    if (threadIdx.x == 0) {
      if (q == 0) info[0] = carry; else info[3 + (q - 1)] = carry;
    }
  }
}
#endif

__global__ void __launch_bounds__(PS_BLOCK) k_pscan_final(const uint32_t* counts, uint32_t nb, uint32_t logG,
                                                          const uint32_t* partial, uint32_t nblocks, uint32_t* cursor,
                                                          uint32_t* tail_off, const uint32_t* info)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint32_t lds_wave[PS_BLOCK / 64];
  const int nq = pscan_nq(info, logG);
  const uint32_t b0 = blockIdx.x * PS_SPAN + threadIdx.x * PS_ITEMS;
  uint32_t n[PS_ITEMS];
#pragma unroll
  for (int j = 0; j < PS_ITEMS; j++) n[j] = (b0 + j) < nb ? counts[b0 + j] : 0u;
  for (int q = 0; q < nq; q++) {
    uint32_t v[PS_ITEMS], sum = 0;
#pragma unroll
    for (int j = 0; j < PS_ITEMS; j++) { v[j] = scan_quantity(n[j], logG, q); sum += v[j]; }
    uint32_t tot;
    uint32_t ex = block_excl_scan(sum, lds_wave, tot) + partial[(uint64_t)q * nblocks + blockIdx.x];
    uint32_t* out = q == 0 ? cursor : tail_off + (uint64_t)(q - 1) * (nb + 1);
#pragma unroll
    for (int j = 0; j < PS_ITEMS; j++) {
      if (b0 + j < nb) out[b0 + j] = ex;
      ex += v[j];
    }
    if (q > 0 && blockIdx.x == 0 && threadIdx.x == 0) out[nb] = info[3 + (q - 1)];
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// LDS-privatised counting sort (used whenever one window's L counters fit the 160 KB LDS, c <= 16).
//   k_hist      : grid (B, Kg); block (b, kk) histograms its slice of window kk's digits in LDS and
//                 writes the L counters to block_hist[kk][b][.] with plain coalesced stores
//   k_colscan   : per bucket, exclusive prefix over the B blocks (in place) and the bucket total
//   k_scatter_lds: block (b, kk) loads its L start positions (bucket slot offset + block prefix) into
//                 LDS and ranks its entries with returning LDS atomics
// The only global atomics left are none at all; the reference's Atomics.add histogram
// (src/msm-batched-affine.ts:197) becomes ds_add_u32 on a CU-private copy.
// ---------------------------------------------------------------------------------------------

constexpr int SORT_THREADS = 1024;

// How window kk's bucket index (l - 1) is cut for the LDS-staged passes: | ab coarse bits | mb mid bits | fb fine bits |.
// Two-pass split (c <= 16): mb = 0, fb = 7.  Three-pass split (c > 16): the cut is made on the window's EFFECTIVE bits -- the
// top window of a scalar usually holds fewer than c - 1 bits, its digits then fill only the low end of the bucket range, and
// with a fixed cut all of them would land in a handful of coarse bins and fine windows (= blocks of the later passes).  A
// window of eff bits keeps as many fine windows as a full one has (2^(c-1-7), fewer buckets each: fb = eff - (c - 1 - 7), 0
// if there are not even that many buckets), ab <= 8 of the bits above them are the coarse bins, mb the rest.
struct WinSplit {
  uint8_t ab[16], mb[16], fb[16];
};


// Ranking with few bins (<= 256, `bits` = log2 of their number): the 64 lanes of a wave then mostly hit the same LDS
// counters and per-lane atomics serialise (round 1 measured the 128-bin pass slower than the 32768-bin one for exactly
// this reason).  Here the lanes of a wave that share a bin find each other with one ballot per bin bit, the lowest of
// them adds the group's size with ONE atomic, and every lane gets (old counter + its rank inside the group).
// bits == 0: plain per-lane atomic.  Must be called by all lanes of the wave (valid = has an entry).
__device__ __forceinline__ uint32_t lds_rank_add(uint32_t* lds, uint32_t bin, bool valid, uint32_t bits) {
  if (bits == 0) return valid ? atomicAdd(&lds[bin], 1u) : 0u;
  uint64_t peers = __ballot(valid);
  for (uint32_t b = 0; b < bits; b++) {
    const bool bit = (bin >> b) & 1u;
    const uint64_t m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
  uint32_t base = 0;
  if (valid && rank == 0) base = atomicAdd(&lds[bin], (uint32_t)__popcll(peers));
  const int leader = valid ? __ffsll((long long)peers) - 1 : 0;
  base = __shfl(base, leader, 64);
  return base + rank;
}

// Window kk owns digits dig[kk * two_n ..) and block b the slice [b * chunk, (b+1) * chunk); L = number of bins;
// bin = l - 1, or (l - 1) >> ws.fb[kk] with fine_windows set (the three-pass split counts its fine windows).
__global__ void __launch_bounds__(SORT_THREADS) k_hist(uint32_t* block_hist, const uint32_t* dig, uint64_t two_n,
                                                       uint64_t chunk, uint32_t L, WinSplit ws, uint32_t fine_windows,
                                                       uint32_t agg_bits)
#ifndef MSM_SORT_TU
    ;
#else
{
  extern __shared__ uint32_t lds_hist[];
  const uint32_t b = blockIdx.x, kk = blockIdx.y, B = gridDim.x;
  const uint32_t shift = fine_windows ? ws.fb[kk] : 0u;
  const uint64_t hist_row = (uint64_t)kk * B + b;
  for (uint32_t l = threadIdx.x; l < L; l += SORT_THREADS) lds_hist[l] = 0;
  __syncthreads();
  const uint64_t beg = (uint64_t)b * chunk, end = min(beg + chunk, two_n);
  const uint32_t* d = dig + (uint64_t)kk * two_n;
  // 16-byte loads, two per thread and trip, while the slice allows it (one dword per thread and trip left the kernel
  // waiting on single 256-byte wave loads: 1.2 ms per window group at 2^26, 1.75 TB/s); whole waves stay in both loops
  // (lds_rank_add ballots)
  uint64_t j0 = beg;
  if (end > beg) {
    // up to three entries in front of the first 16-byte boundary (odd N, odd slice starts)
    const uint64_t head = min<uint64_t>(end - beg, (16u - (uint32_t)(reinterpret_cast<uintptr_t>(d + beg) & 15u)) % 16u / 4u);
    if (head) {   // uniform: all lanes of the block take part (lds_rank_add ballots)
      const uint32_t l = threadIdx.x < head ? d[beg + threadIdx.x] & 0x7FFFFFFFu : 0u;
      (void)lds_rank_add(lds_hist, l ? (l - 1) >> shift : 0u, l != 0, agg_bits);
      j0 = beg + head;
    }
  }
  if (end > j0) {
    const uint4* dv = reinterpret_cast<const uint4*>(d + j0);
    const uint64_t nvec = (end - j0) / 4;
    for (uint64_t q0 = 0; q0 < nvec; q0 += 2 * SORT_THREADS) {
      const uint64_t qa = q0 + threadIdx.x, qb = qa + SORT_THREADS;
      const uint4 va = qa < nvec ? dv[qa] : make_uint4(0, 0, 0, 0);
      const uint4 vb = qb < nvec ? dv[qb] : make_uint4(0, 0, 0, 0);
      const uint32_t w[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint32_t l = w[i] & 0x7FFFFFFFu;
        (void)lds_rank_add(lds_hist, l ? (l - 1) >> shift : 0u, l != 0, agg_bits);
      }
    }
    j0 += nvec * 4;
  }
  for (; j0 < end; j0 += SORT_THREADS) {
    const uint64_t j = j0 + threadIdx.x;
    const uint32_t l = j < end ? d[j] & 0x7FFFFFFFu : 0u;
    (void)lds_rank_add(lds_hist, l ? (l - 1) >> shift : 0u, l != 0, agg_bits);
  }
  __syncthreads();
  uint32_t* out = block_hist + hist_row * L;
  for (uint32_t l = threadIdx.x; l < L; l += SORT_THREADS) out[l] = lds_hist[l];
}
#endif

__global__ void __launch_bounds__(256) k_colscan(uint32_t* block_hist, uint32_t* counts, uint32_t B, uint32_t L, uint32_t k_cnt)
#ifndef MSM_SORT_TU
    ;
#else
{
  uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (uint64_t)k_cnt * L) return;
  uint32_t kk = (uint32_t)(id / L), l = (uint32_t)(id - (uint64_t)kk * L);
  uint32_t* p = block_hist + (uint64_t)kk * B * L + l;
  uint32_t run = 0, b = 0;
  for (; b + 8 <= B; b += 8) {   // eight loads in flight per thread (one at a time made this a chain of B latencies)
    uint32_t v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = p[(uint64_t)(b + i) * L];
#pragma unroll
    for (int i = 0; i < 8; i++) { p[(uint64_t)(b + i) * L] = run; run += v[i]; }
  }
  for (; b < B; b++) {
    uint32_t v = p[(uint64_t)b * L];
    p[(uint64_t)b * L] = run;
    run += v;
  }
  counts[id] = run;
}
#endif

__global__ void __launch_bounds__(SORT_THREADS) k_scatter_lds(uint32_t* slots, const uint32_t* cursor,
                                                              const uint32_t* block_hist, const uint32_t* dig,
                                                              uint64_t two_n, uint64_t chunk, uint32_t L, uint32_t agg_bits)
#ifndef MSM_SORT_TU
    ;
#else
{
  extern __shared__ uint32_t lds_pos[];
  const uint32_t b = blockIdx.x, kk = blockIdx.y, B = gridDim.x;
  const uint32_t* base = block_hist + ((uint64_t)kk * B + b) * L;
  const uint32_t* cur = cursor + (uint64_t)kk * L;
  for (uint32_t l = threadIdx.x; l < L; l += SORT_THREADS) lds_pos[l] = cur[l] + base[l];
  __syncthreads();
  const uint64_t beg = (uint64_t)b * chunk, end = min(beg + chunk, two_n);
  const uint32_t* d = dig + (uint64_t)kk * two_n;
  for (uint64_t j0 = beg; j0 < end; j0 += SORT_THREADS) {
    const uint64_t j = j0 + threadIdx.x;
    const uint32_t v = j < end ? d[j] : 0u;
    const uint32_t l = v & 0x7FFFFFFFu;
    const uint32_t pos = lds_rank_add(lds_pos, l ? l - 1 : 0u, l != 0, agg_bits);
    if (l) slots[pos] = ((uint32_t)j << 1) | (v >> 31);
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// LDS-staged radix split (big inputs at c <= 16; replaces the one-level scatter there).
//
// Why: the scatter is bound by the NUMBER of store requests, not by bytes -- a wave of the one-level kernel sends its
// 64 four-byte payloads to ~64 different cache lines, and ~5 x 10^10 such partial-line stores per second is what the
// memory system takes (round 1: 7.7x the algorithmic bytes written).  Here every block first sorts a tile of its entries
// by bin INSIDE the LDS and then copies the tile out with consecutive lanes on consecutive addresses, so a wave store is
// one or two full segments.  Two passes keep the runs long: pass A splits a window's 2^(c-1) buckets into 2^(c-8) coarse
// bins (runs of ~32 entries per tile and bin in two 4-byte arrays), pass B sorts every coarse bin by its 128 buckets
// (runs of ~64 payloads).
//   k_hist (L bins) + k_colscan : as in the one-level sort -- bucket totals `counts` and, per block, the exclusive
//                                 prefix over blocks of every bucket (block_hist, in place)
//   k_coarse_offsets            : per (window, block, coarse bin) the block's first position inside the bin's range,
//                                 = sum over the bin's 128 buckets of that prefix; per virtual window v = (window, bin)
//                                 its total
//   k_vscan                     : exclusive scan of the V totals -> v_start[V + 1]
//   k_radix_coarse              : pass A, records (fine digit | sign, entry index) -> dig2, idx2
//   k_radix_fine                : pass B, ONE block per virtual window walks its range tile by tile with running
//                                 per-bucket cursors in the LDS (no second histogram), payloads -> slots
// ---------------------------------------------------------------------------------------------

constexpr int RX_THREADS = 1024;
#ifndef RXA_WAVES
#define RXA_WAVES 8   // pass A: two workgroups per CU (60 KB of LDS each) -- one loads its tile while the other ranks
#endif
constexpr int RXA_ITEMS = 7, RXA_TILE = RX_THREADS * RXA_ITEMS;    // pass A: 7168 records of 8 bytes staged per tile (56 KB)
#ifndef RXB_THREADS
#define RXB_THREADS 512   // pass B: 512-thread workgroups, two resident per CU (its ~108 VGPRs allow four waves per SIMD): one walks
#endif                    // its virtual window's next tile in while the other ranks (1024 threads: one workgroup per CU, 2.46 ms)
constexpr int RXB_ITEMS = 12, RXB_TILE = RXB_THREADS * RXB_ITEMS;   // payloads + bucket bytes staged per tile (30 KB)
constexpr uint32_t RX_FINE_BITS = 7;

// exclusive scan of `nbins` (<= 256) LDS counters by the first 256 threads; `tot` counters become `start`
__device__ __forceinline__ void rx_scan_bins(uint32_t* start, const uint32_t* cnt, uint32_t nbins, uint32_t* lds_wave) {
  uint32_t v = threadIdx.x < nbins ? cnt[threadIdx.x] : 0u, tot;
  uint32_t ex = block_excl_scan(v, lds_wave, tot);
  if (threadIdx.x < nbins) start[threadIdx.x] = ex;
  __syncthreads();
}

__global__ void __launch_bounds__(256) k_coarse_offsets(uint32_t* blk_off, uint32_t* v_tot, const uint32_t* block_hist,
                                                        const uint32_t* counts, uint32_t B, uint32_t L, uint32_t Hn, uint32_t kc)
#ifndef MSM_SORT_TU
    ;
#else
{
  // one thread per (kk, b, h): b == B is the extra row that produces the totals from `counts`
  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t total = (uint64_t)kc * (B + 1) * Hn;
  if (id >= total) return;
  const uint32_t h = (uint32_t)(id % Hn);
  const uint32_t b = (uint32_t)((id / Hn) % (B + 1));
  const uint32_t kk = (uint32_t)(id / ((uint64_t)Hn * (B + 1)));
  const uint32_t fine = L / Hn;
  const uint32_t* src = (b == B) ? counts + (uint64_t)kk * L + (uint64_t)h * fine
                                 : block_hist + ((uint64_t)kk * B + b) * L + (uint64_t)h * fine;
  uint32_t sum = 0;
  for (uint32_t i = 0; i < fine; i += 4) {
    const uint4 q = *reinterpret_cast<const uint4*>(src + i);
    sum += q.x + q.y + q.z + q.w;
  }
  if (b == B) v_tot[(uint64_t)kk * Hn + h] = sum;
  else blk_off[((uint64_t)kk * B + b) * Hn + h] = sum;
}
#endif

__global__ void __launch_bounds__(SCAN_THREADS) k_vscan(uint32_t* v_start, const uint32_t* v_tot, uint32_t V)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint32_t lds_wave[SCAN_THREADS / 64];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < V; base += SCAN_THREADS) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < V ? v_tot[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan(v, lds_wave, tot) + carry;
    if (i < V) v_start[i] = ex;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) v_start[V] = carry;
}
#endif

// pass A.  grid (B, kc); block (b, kk) owns entries [b * chunk, (b + 1) * chunk) of window kk.  Coarse bin h of window kk
// starts at v_start[kk * vs_stride + (h << mb)]; this block's share of it at + blk_off[(kk * B + b) * bo_stride + h].
__global__ void __launch_bounds__(RX_THREADS, RXA_WAVES) k_radix_coarse(uint32_t* dig2, uint32_t* idx2, const uint32_t* v_start,
                                                             const uint32_t* blk_off, const uint32_t* dig, uint64_t two_n,
                                                             uint64_t chunk, uint32_t vs_stride, uint32_t bo_stride, WinSplit ws)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint2 stage[RXA_TILE];
  __shared__ uint32_t t_cnt[256], t_start[256], g_base[256], lds_wave[RX_THREADS / 64];
  const uint32_t b = blockIdx.x, kk = blockIdx.y, B = gridDim.x, tid = threadIdx.x;
  const uint32_t hbits = ws.ab[kk], mb = ws.mb[kk], Hn = 1u << hbits, low_bits = ws.fb[kk] + mb;
  if (tid < Hn) g_base[tid] = v_start[(uint64_t)kk * vs_stride + ((uint64_t)tid << mb)] + blk_off[((uint64_t)kk * B + b) * bo_stride + tid];
  const uint64_t beg = (uint64_t)b * chunk, end = min(beg + chunk, two_n);
  const uint32_t* d = dig + (uint64_t)kk * two_n;
  const uint32_t lo_mask = (1u << low_bits) - 1;
  for (uint64_t t0 = beg; t0 < end; t0 += RXA_TILE) {
    if (tid < 256) t_cnt[tid] = 0;
    __syncthreads();
    uint32_t v[RXA_ITEMS], rk[RXA_ITEMS];
    const uint32_t left = (uint32_t)min<uint64_t>(end - t0, RXA_TILE);   // 32-bit tile-relative indices: fewer registers
    const uint32_t* dt = d + t0;
#pragma unroll
    for (int i = 0; i < RXA_ITEMS; i++) {
      const uint32_t j = (uint32_t)i * RX_THREADS + tid;
      v[i] = j < left ? dt[j] : 0u;
    }
#pragma unroll
    for (int i = 0; i < RXA_ITEMS; i++) {
      const uint32_t l = v[i] & 0x7FFFFFFFu;
      rk[i] = lds_rank_add(t_cnt, l ? (l - 1) >> low_bits : 0u, l != 0, hbits ? hbits : 1u);   // 0 bits would mean per-lane atomics
      __builtin_amdgcn_sched_barrier(0);   // one ranking at a time: interleaved, the seven of them cost 16 more registers
    }
    __syncthreads();
    rx_scan_bins(t_start, t_cnt, Hn, lds_wave);
#pragma unroll
    for (int i = 0; i < RXA_ITEMS; i++) {
      const uint32_t l = v[i] & 0x7FFFFFFFu;
      if (l) {
        const uint32_t h = (l - 1) >> low_bits;
        // record: low bits + 1 (so 0 still means "no entry"; <= 2^15) | coarse bin << 16 (stripped on the way out) | sign
        stage[t_start[h] + rk[i]] = make_uint2((((l - 1) & lo_mask) + 1) | (h << 16) | (v[i] & 0x80000000u),
                                               (uint32_t)(t0 + (uint64_t)i * RX_THREADS + tid));
      }
    }
    __syncthreads();
    const uint32_t n_tile = t_start[Hn - 1] + t_cnt[Hn - 1];
    for (uint32_t i = tid; i < n_tile; i += RX_THREADS) {
      const uint2 r = stage[i];
      const uint32_t h = (r.x >> 16) & 0xFFu;
      const uint32_t pos = g_base[h] + (i - t_start[h]);
      dig2[pos] = r.x & 0x8000FFFFu;
      idx2[pos] = r.y;
    }
    __syncthreads();
    if (tid < Hn) g_base[tid] += t_cnt[tid];
  }
}
#endif

// pass B.  One block per fine window: block v = kk * Lp + f owns the 2^fb buckets from f << fb of window kk (heaviest -- the
// top window's few coarse bins on the two-pass path -- first: v = V - 1 - blockIdx.x).  Records: (bucket's low fb bits) + 1 |
// sign << 31, entry index.  A window with fb = 0 has one bucket per fine window and possibly very few of them (a top window
// of a handful of bits): ranking is the identity there, and all Lp blocks of the window copy an equal slice of its records.
__global__ void __launch_bounds__(RXB_THREADS) k_radix_fine(uint32_t* slots, const uint32_t* cursor, const uint32_t* v_start,
                                                           const uint32_t* dig2, const uint32_t* idx2, uint32_t Lp, uint32_t L,
                                                           WinSplit ws)
#ifndef MSM_SORT_TU
    ;
#else
{
  constexpr uint32_t NBMAX = 256;
  __shared__ uint32_t stage[RXB_TILE];
  __shared__ uint8_t stage_b[RXB_TILE];
  __shared__ uint32_t t_cnt[NBMAX], t_start[NBMAX], g_cur[NBMAX], lds_wave[RXB_THREADS / 64];
  const uint32_t v = gridDim.x - 1 - blockIdx.x, tid = threadIdx.x;
  const uint32_t kk = v / Lp, f = v - kk * Lp;
  const uint32_t fbits = ws.fb[kk], NB = 1u << fbits;
  if (fbits == 0) {
    const uint32_t nfw = 1u << (ws.ab[kk] + ws.mb[kk]);          // fine windows (= buckets) in use
    const uint32_t* vs = v_start + (uint64_t)kk * Lp;
    const uint64_t wbeg = vs[0], wlen = (uint64_t)vs[nfw] - wbeg;
    const uint64_t beg = wbeg + wlen * f / Lp, end = wbeg + wlen * (f + 1) / Lp;
    for (uint64_t j = beg + tid; j < end; j += RXB_THREADS) {
      uint32_t lo = 0, hi = nfw;                                  // bucket of record j: last fine window starting at or before it
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (vs[mid] <= j) lo = mid; else hi = mid;
      }
      slots[cursor[(uint64_t)kk * L + lo] + (uint32_t)(j - vs[lo])] = (idx2[j] << 1) | (dig2[j] >> 31);
    }
    return;
  }
  const uint64_t beg = v_start[v], end = v_start[v + 1];
  if (beg == end) return;
  if (tid < NB) g_cur[tid] = cursor[(uint64_t)kk * L + ((uint64_t)f << fbits) + tid];
  for (uint64_t t0 = beg; t0 < end; t0 += RXB_TILE) {
    if (tid < NB) t_cnt[tid] = 0;
    __syncthreads();
    uint32_t dv[RXB_ITEMS], iv[RXB_ITEMS], rk[RXB_ITEMS];
#pragma unroll
    for (int i = 0; i < RXB_ITEMS; i++) {
      const uint64_t j = t0 + (uint64_t)i * RXB_THREADS + tid;
      dv[i] = j < end ? dig2[j] : 0u;
      iv[i] = j < end ? idx2[j] : 0u;
    }
#pragma unroll
    for (int i = 0; i < RXB_ITEMS; i++) {
      const uint32_t l = dv[i] & 0xFFFFu;
      rk[i] = lds_rank_add(t_cnt, l ? l - 1 : 0u, l != 0, fbits);
    }
    __syncthreads();
    rx_scan_bins(t_start, t_cnt, NB, lds_wave);
#pragma unroll
    for (int i = 0; i < RXB_ITEMS; i++) {
      const uint32_t l = dv[i] & 0xFFFFu;
      if (l) {
        const uint32_t p = t_start[l - 1] + rk[i];
        stage[p] = (iv[i] << 1) | (dv[i] >> 31);
        stage_b[p] = (uint8_t)(l - 1);
      }
    }
    __syncthreads();
    const uint32_t n_tile = t_start[NB - 1] + t_cnt[NB - 1];
    for (uint32_t i = tid; i < n_tile; i += RXB_THREADS) {
      const uint32_t bk = stage_b[i];
      slots[g_cur[bk] + (i - t_start[bk])] = stage[i];
    }
    __syncthreads();
    if (tid < NB) g_cur[tid] += t_cnt[tid];
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// Three-pass split for windows of more than 2^15 buckets (c > 16), where one window's counters no longer fit the LDS.
//   k_hist (shift = fb)    : per block the histogram over the 2^(c-1-fb) groups of 2^fb buckets ("fine windows")
//   k_colscan, k_vscan     : totals per fine window -> v2_start (these are also the starts of the mid and coarse bins)
//   k_coarse_offsets3      : per (window, block, coarse bin) the block's first position in the bin
//   k_radix_coarse         : pass A by the top ab bits -> (dig2, idx2), records keep mb + fb low bits
//   k_radix_mid            : pass M, one block per coarse bin, by the next mb bits -> (dig3, idx3), records keep fb bits
//   k_fine_hist            : bucket sizes, one block per fine window (the scans of the padded slot offsets need them)
//   k_radix_fine           : pass B as above, payloads to their padded slots
// ---------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) k_coarse_offsets3(uint32_t* blk_off, const uint32_t* block_hist, uint32_t B, uint32_t Lp,
                                                         uint32_t kc, WinSplit ws)
#ifndef MSM_SORT_TU
    ;
#else
{
  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (uint64_t)kc * B * 256) return;
  const uint32_t h = (uint32_t)(id & 255u);
  const uint64_t row = id >> 8;   // kk * B + b
  const uint32_t kk = (uint32_t)(row / B);
  const uint32_t mb = ws.mb[kk];
  uint32_t sum = 0;
  if (h < (1u << ws.ab[kk])) {
    const uint32_t* src = block_hist + row * Lp + ((uint64_t)h << mb);
    for (uint32_t i = 0; i < (1u << mb); i++) sum += src[i];
  }
  blk_off[id] = sum;
}
#endif

// pass M.  grid (256, kc): block (h, kk) owns coarse bin h of window kk = the fine windows [h << mb, (h + 1) << mb).
// A window without mid bits (mb = 0) keeps its order: all 256 blocks copy an equal slice of its records.
__global__ void __launch_bounds__(RXB_THREADS) k_radix_mid(uint32_t* dig3, uint32_t* idx3, const uint32_t* v2_start,
                                                          const uint32_t* dig2, const uint32_t* idx2, uint32_t Lp, WinSplit ws)
#ifndef MSM_SORT_TU
    ;
#else
{
  constexpr uint32_t NBMAX = 128;
  __shared__ uint2 stage[RXB_TILE];
  __shared__ uint8_t stage_b[RXB_TILE];
  __shared__ uint32_t t_cnt[NBMAX], t_start[NBMAX], g_cur[NBMAX], lds_wave[RXB_THREADS / 64];
  const uint32_t kk = blockIdx.y, h = gridDim.x - 1 - blockIdx.x, tid = threadIdx.x;   // high bins (sparse in a short top window) last
  const uint32_t mb = ws.mb[kk], NB = 1u << mb, fb = ws.fb[kk];
  const uint32_t fmask = (1u << fb) - 1;
  if (mb == 0) {
    const uint32_t* vs = v2_start + (uint64_t)kk * Lp;
    const uint64_t wbeg = vs[0], wlen = (uint64_t)vs[1u << ws.ab[kk]] - wbeg;
    const uint64_t beg = wbeg + wlen * h / gridDim.x, end = wbeg + wlen * (h + 1) / gridDim.x;
    for (uint64_t j = beg + tid; j < end; j += RXB_THREADS) {
      const uint32_t d = dig2[j], l = d & 0xFFFFu;   // every record of the range is an entry (l >= 1)
      dig3[j] = (((l - 1) & fmask) + 1) | (d & 0x80000000u);
      idx3[j] = idx2[j];
    }
    return;
  }
  if (h >= (1u << ws.ab[kk])) return;
  const uint32_t* vs = v2_start + (uint64_t)kk * Lp + ((uint64_t)h << mb);
  const uint64_t beg = vs[0], end = vs[NB];
  if (beg == end) return;
  if (tid < NB) g_cur[tid] = vs[tid];
  for (uint64_t t0 = beg; t0 < end; t0 += RXB_TILE) {
    if (tid < NB) t_cnt[tid] = 0;
    __syncthreads();
    uint32_t dv[RXB_ITEMS], iv[RXB_ITEMS], rk[RXB_ITEMS];
#pragma unroll
    for (int i = 0; i < RXB_ITEMS; i++) {
      const uint64_t j = t0 + (uint64_t)i * RXB_THREADS + tid;
      dv[i] = j < end ? dig2[j] : 0u;
      iv[i] = j < end ? idx2[j] : 0u;
    }
#pragma unroll
    for (int i = 0; i < RXB_ITEMS; i++) {
      const uint32_t l = dv[i] & 0xFFFFu;
      rk[i] = lds_rank_add(t_cnt, l ? (l - 1) >> fb : 0u, l != 0, mb);
    }
    __syncthreads();
    rx_scan_bins(t_start, t_cnt, NB, lds_wave);
#pragma unroll
    for (int i = 0; i < RXB_ITEMS; i++) {
      const uint32_t l = dv[i] & 0xFFFFu;
      if (l) {
        const uint32_t m = (l - 1) >> fb, p = t_start[m] + rk[i];
        stage[p] = make_uint2((((l - 1) & fmask) + 1) | (dv[i] & 0x80000000u), iv[i]);
        stage_b[p] = (uint8_t)m;
      }
    }
    __syncthreads();
    const uint32_t n_tile = t_start[NB - 1] + t_cnt[NB - 1];
    for (uint32_t i = tid; i < n_tile; i += RXB_THREADS) {
      const uint32_t bk = stage_b[i];
      const uint32_t pos = g_cur[bk] + (i - t_start[bk]);
      const uint2 r = stage[i];
      dig3[pos] = r.x;
      idx3[pos] = r.y;
    }
    __syncthreads();
    if (tid < NB) g_cur[tid] += t_cnt[tid];
  }
}
#endif

// bucket sizes of fine window v = kk * Lp + f (2^fb buckets from f << fb) from its records; `counts` is zeroed before
// (a short top window does not reach the upper buckets)
__global__ void __launch_bounds__(256) k_fine_hist(uint32_t* counts, const uint32_t* v2_start, const uint32_t* dig3, uint32_t Lp,
                                                   uint32_t L, WinSplit ws)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint32_t hist[256];
  const uint32_t v = blockIdx.x, tid = threadIdx.x;
  const uint32_t kk = v / Lp, f = v - kk * Lp, fb = ws.fb[kk], NB = 1u << fb;
  const uint64_t beg = v2_start[v], end = v2_start[v + 1];
  if (beg == end) return;
  uint32_t* out = counts + (uint64_t)kk * L + ((uint64_t)f << fb);
  if (fb == 0) {   // one bucket
    if (tid == 0) out[0] = (uint32_t)(end - beg);
    return;
  }
  hist[tid] = 0;
  __syncthreads();
  for (uint64_t j = beg + tid; j < end; j += 256) {
    const uint32_t l = dig3[j] & 0xFFFFu;
    if (l) atomicAdd(&hist[l - 1], 1u);
  }
  __syncthreads();
  if (tid < NB) out[tid] = hist[tid];
}
#endif

// ---------------------------------------------------------------------------------------------
// k_chunk_order: round 1 of the tree gathers its operands from the point rows, and scattered 128-byte line reads run 2.3x
// slower once the 64 lanes of one load instruction spread over more than ~1 GB of the table (tools/ubench_gather2.hip:
// 30 G lines/s inside 1 GB windows, 20 inside 2 GB, 13 over 16 GB, however the table is allocated).  A bucket's payloads
// are in point order, so at c = 16 a wave's 64 pairs cover 1/32 of the table -- but 1/4 at c = 19 and all of it at c = 22.
// This pass restores the locality for big windows without touching the sort: every block of CO_PAIRS consecutive pairs of
// the bucket-sorted slots (~100 buckets at c = 22) is stably partitioned by the 1 GB chunk of the table its first operand
// lives in.  Round 1 then walks the pairs in that order -- consecutive lanes read inside one chunk -- and writes each sum
// to the element index the pair had before (oidx), so round 2 still finds a bucket's elements side by side.  Pairs of
// pads (both operands missing) go last.
// ---------------------------------------------------------------------------------------------

constexpr int CO_THREADS = 256, CO_PPT = CO_PAIRS / CO_THREADS;   // 16 consecutive pairs per thread
#ifndef MSM_CO_MAX_KEYS
#define MSM_CO_MAX_KEYS 65
#endif
constexpr int CO_MAX_KEYS = MSM_CO_MAX_KEYS;                      // up to 64 chunks + the pads

__global__ void __launch_bounds__(CO_THREADS) k_chunk_order(uint2* pairs_out, uint16_t* oidx, const uint2* pairs_in, uint64_t n_pairs,
                                                            uint32_t row_shift, uint32_t nkeys)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint2 stage[CO_PAIRS];
  __shared__ uint16_t stage_o[CO_PAIRS];
  __shared__ uint16_t cnt[CO_MAX_KEYS * CO_THREADS];   // [key][thread]
  __shared__ uint32_t lds_wave[CO_THREADS / 64];
  const uint32_t tid = threadIdx.x;
  const uint64_t base = (uint64_t)blockIdx.x * CO_PAIRS;
  const uint32_t n_valid = (uint32_t)min<uint64_t>(CO_PAIRS, n_pairs - base);
  for (uint32_t i = tid; i < nkeys * CO_THREADS; i += CO_THREADS) cnt[i] = 0;
  uint2 pr[CO_PPT];
  uint32_t key[CO_PPT];
#pragma unroll
  for (int q = 0; q < CO_PPT / 2; q++) {   // 16 pairs = 8 x 16 bytes per thread
    const uint32_t j = tid * CO_PPT + 2 * q;
    uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    if (j + 1 < n_valid) v = *reinterpret_cast<const uint4*>(pairs_in + base + j);
    else if (j < n_valid) { const uint2 u = pairs_in[base + j]; v.x = u.x; v.y = u.y; }
    pr[2 * q] = make_uint2(v.x, v.y);
    pr[2 * q + 1] = make_uint2(v.z, v.w);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CO_PPT; j++) {
    // first operand present: its chunk; else the second one's; both missing: the last key
    const uint32_t pa = pr[j].x != 0xFFFFFFFFu ? pr[j].x : pr[j].y;
    key[j] = pa != 0xFFFFFFFFu ? min((pa >> 2) >> row_shift, nkeys - 2) : nkeys - 1;
    cnt[key[j] * CO_THREADS + tid]++;
  }
  __syncthreads();
  // exclusive scan of the flattened [key][thread] counters: thread t owns the nkeys consecutive entries from t * nkeys
  {
    uint32_t sum = 0;
    for (uint32_t i = 0; i < nkeys; i++) sum += cnt[tid * nkeys + i];
    uint32_t tot;
    uint32_t ex = block_excl_scan(sum, lds_wave, tot);
    for (uint32_t i = 0; i < nkeys; i++) {
      const uint32_t c = cnt[tid * nkeys + i];
      cnt[tid * nkeys + i] = (uint16_t)ex;
      ex += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CO_PPT; j++) {
    const uint32_t p = cnt[key[j] * CO_THREADS + tid]++;
    stage[p] = pr[j];
    stage_o[p] = (uint16_t)(tid * CO_PPT + j);
  }
  __syncthreads();
  for (uint32_t i = tid; i < n_valid; i += CO_THREADS) {
    pairs_out[base + i] = stage[i];
    oidx[base + i] = stage_o[i];
  }
}
#endif

}  // namespace msm
