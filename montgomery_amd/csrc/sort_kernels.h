// Curve-independent kernels of the MSM pipeline: bucket-size scans and the LDS-privatised counting sort -- one level for small
// inputs, the LDS-staged two-pass radix split for c <= 16, the two-pass bin split (8-byte records, pairs of round 1 out of its
// second pass, heavy bins cut into parts) up to the largest accepted window c = 24.
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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence too: it waits for every global load and store
// of the wave (s_waitcnt vmcnt(0)) before the barrier, so a tile loop with one workgroup per CU stalls at each barrier until its
// copy-out has drained to the L2 and the next tile's loads are back.  The passes of the bin split exchange data between
// threads through the LDS alone (what they load from or store to global memory is private to the thread), so their barriers
// need not wait for it.
__device__ __forceinline__ void lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

template <bool LDS_ONLY = false>
MSM_DEV uint32_t block_excl_scan(uint32_t v, uint32_t* lds_wave, uint32_t& total) {
  // inclusive scan inside the wave
  uint32_t x = v;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (LDS_ONLY) lds_barrier(); else __syncthreads();
  if (lane == 63) lds_wave[wave] = x;
  if (LDS_ONLY) lds_barrier(); else __syncthreads();
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

// (four consecutive buckets per lane: a wave instruction of the scans touches 16 sectors, not 64)
constexpr int PS_BLOCK = 1024;
constexpr int PS_ITEMS = 4;
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
    if (blockIdx.x == 0 && threadIdx.x == 0) out[nb] = q == 0 ? info[0] : info[3 + (q - 1)];   // (the cursor has nb + 1 entries too)
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



// Ranking a key into its LDS counter: one returning ds_add per lane.  Measured (tools/ubench_lds_rank.hip,
// profiles/r05_ubench_lds_rank.txt): 3.7 keys per clock and CU on random keys over 2^8 or 2^10 counters, 1.8 over 8 -- the
// match-by-ballot form of rounds 1-4 (one ballot per bin bit, one atomic per group of equal lanes) reaches 0.6-0.7 and 1.1, so it
// lost everywhere on uniform digits.  What it did buy is a bound on skewed inputs, where many lanes of a wave share ONE bin and
// their atomics serialise: that case is kept as a wave-uniform side branch -- if at least 16 lanes hold the first lane's bin,
// those lanes are ranked with one ballot and one atomic.  Must be called by all lanes of the wave (valid = has an entry).
__device__ __forceinline__ uint32_t lds_rank_add(uint32_t* lds, uint32_t bin, bool valid) {
  const uint32_t first = __builtin_amdgcn_readfirstlane(bin);
  const uint64_t same = __ballot(valid && bin == first);
  if (__popcll(same) >= 16) {
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __ffsll((long long)same) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&lds[first], (uint32_t)__popcll(same));
    base = __shfl(base, leader, 64);
    const uint32_t r = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
    if ((same >> lane) & 1ull) return r;
    return valid ? atomicAdd(&lds[bin], 1u) : 0u;
  }
  return valid ? atomicAdd(&lds[bin], 1u) : 0u;
}

// Window kk owns digits dig[kk * two_n ..) and block b the slice [b * chunk, (b+1) * chunk); L = number of bins; bin = l - 1.
__global__ void __launch_bounds__(SORT_THREADS) k_hist(uint32_t* block_hist, const uint32_t* dig, uint64_t two_n,
                                                       uint64_t chunk, uint32_t L)
#ifndef MSM_SORT_TU
    ;
#else
{
  extern __shared__ uint32_t lds_hist[];
  const uint32_t b = blockIdx.x, kk = blockIdx.y, B = gridDim.x;
  constexpr uint32_t shift = 0;
  const uint64_t hist_row = (uint64_t)kk * B + b;
  for (uint32_t l = threadIdx.x; l < L; l += SORT_THREADS) lds_hist[l] = 0;
  __syncthreads();
  const uint64_t beg = (uint64_t)b * chunk, end = min(beg + chunk, two_n);
  const uint32_t* d = dig + (uint64_t)kk * two_n;
  // 16-byte loads, two per thread and trip, while the slice allows it (one dword per thread and trip left the kernel
  // waiting on single 256-byte wave loads: 1.2 ms per window group at 2^26, 1.75 TB/s); whole waves stay in both loops
  // (lds_rank_add looks at the whole wave)
  uint64_t j0 = beg;
  if (end > beg) {
    // up to three entries in front of the first 16-byte boundary (odd N, odd slice starts)
    const uint64_t head = min<uint64_t>(end - beg, (16u - (uint32_t)(reinterpret_cast<uintptr_t>(d + beg) & 15u)) % 16u / 4u);
    if (head) {   // uniform: all lanes of the block take part (lds_rank_add looks at the whole wave)
      const uint32_t l = threadIdx.x < head ? d[beg + threadIdx.x] & 0x7FFFFFFFu : 0u;
      (void)lds_rank_add(lds_hist, l ? (l - 1) >> shift : 0u, l != 0);
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
        (void)lds_rank_add(lds_hist, l ? (l - 1) >> shift : 0u, l != 0);
      }
    }
    j0 += nvec * 4;
  }
  for (; j0 < end; j0 += SORT_THREADS) {
    const uint64_t j = j0 + threadIdx.x;
    const uint32_t l = j < end ? d[j] & 0x7FFFFFFFu : 0u;
    (void)lds_rank_add(lds_hist, l ? (l - 1) >> shift : 0u, l != 0);
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
                                                              uint64_t two_n, uint64_t chunk, uint32_t L)
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
    const uint32_t pos = lds_rank_add(lds_pos, l ? l - 1 : 0u, l != 0);
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
constexpr int RXA_WAVES = 8;   // pass A: two workgroups per CU (60 KB of LDS each) -- one loads its tile while the other ranks
constexpr int RXA_ITEMS = 7, RXA_TILE = RX_THREADS * RXA_ITEMS;    // pass A: 7168 records of 8 bytes staged per tile (56 KB)
// pass B: 512-thread workgroups, two resident per CU (its ~108 VGPRs allow four waves per SIMD): one walks its virtual window's
// next tile in while the other ranks (1024 threads: one workgroup per CU, 2.46 ms)
constexpr int RXB_THREADS = 512;
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

// (zero64: 64 words this launch clears on the way -- the `info` block of the bucket scans -- or null: one fill fewer in the stream)
// The bin split also takes its part table from here (see "Parts of heavy bins" below; v_tot[v] is the size of bin v): extra_first[v]
// = parts beyond the first of the bins before v, mp_first[v] = rows of `sub` of the bins before v (V + 1 entries each), and
// part_pair_off (V + 2 words) is cleared for k_part_scan -- one launch instead of three in front of the count pass.
struct PartTables {
  uint32_t* extra_first;   // nullptr: no part table (radix split)
  uint32_t* mp_first;
  uint32_t* part_pair_off;
  uint32_t hb, part_len;
  uint8_t fb[16];
};
__global__ void __launch_bounds__(SCAN_THREADS) k_vscan(uint32_t* v_start, const uint32_t* v_tot, uint32_t V, uint32_t* zero64, PartTables pt)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint32_t lds_wave[SCAN_THREADS / 64];
  if (zero64 && threadIdx.x < 64) zero64[threadIdx.x] = 0;
  uint32_t carry = 0, carry_e = 0, carry_m = 0;
  for (uint32_t base = 0; base < V; base += SCAN_THREADS) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < V ? v_tot[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan(v, lds_wave, tot) + carry;
    if (i < V) v_start[i] = ex;
    carry += tot;
    __syncthreads();
    if (pt.extra_first) {   // uniform
      uint32_t np = 1;
      if (i < V && pt.fb[i / pt.hb] != 0 && v > pt.part_len) np = (v + pt.part_len - 1) / pt.part_len;   // (a window pass A sorted outright has no parts)
      uint32_t tot_e, tot_m;
      const uint32_t ex_e = block_excl_scan(np - 1, lds_wave, tot_e) + carry_e;
      __syncthreads();
      const uint32_t ex_m = block_excl_scan(np > 1 ? np : 0u, lds_wave, tot_m) + carry_m;
      __syncthreads();
      if (i < V) { pt.extra_first[i] = ex_e; pt.mp_first[i] = ex_m; }
      carry_e += tot_e;
      carry_m += tot_m;
    }
  }
  if (threadIdx.x == 0) v_start[V] = carry;
  if (pt.extra_first) {
    if (threadIdx.x == 0) { pt.extra_first[V] = carry_e; pt.mp_first[V] = carry_m; }
    for (uint32_t j = threadIdx.x; j < V + 2; j += SCAN_THREADS) pt.part_pair_off[j] = 0;
  }
}
#endif

// pass A.  grid (B, kc); block (b, kk) owns entries [b * chunk, (b + 1) * chunk) of window kk.  Coarse bin h of window kk
// starts at v_start[kk * Hn + h]; this block's share of it at + blk_off[(kk * B + b) * Hn + h].
__global__ void __launch_bounds__(RX_THREADS, RXA_WAVES) k_radix_coarse(uint32_t* dig2, uint32_t* idx2, const uint32_t* v_start,
                                                             const uint32_t* blk_off, const uint32_t* dig, uint64_t two_n,
                                                             uint64_t chunk, uint32_t Hn_all, WinSplit ws)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint2 stage[RXA_TILE];
  __shared__ uint32_t t_cnt[256], t_start[256], g_base[256], lds_wave[RX_THREADS / 64];
  const uint32_t b = blockIdx.x, kk = blockIdx.y, B = gridDim.x, tid = threadIdx.x;
  const uint32_t hbits = ws.ab[kk], Hn = 1u << hbits, low_bits = ws.fb[kk];
  if (tid < Hn) g_base[tid] = v_start[(uint64_t)kk * Hn_all + tid] + blk_off[((uint64_t)kk * B + b) * Hn_all + tid];
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
      rk[i] = lds_rank_add(t_cnt, l ? (l - 1) >> low_bits : 0u, l != 0);
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

// pass B.  One block per coarse bin: block v = kk * Lp + f owns the 2^fb buckets from f << fb of window kk (heaviest -- the
// top window's few coarse bins -- first: v = V - 1 - blockIdx.x).  Records: (bucket's low fb bits) + 1 | sign << 31, entry index.
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
      rk[i] = lds_rank_add(t_cnt, l ? l - 1 : 0u, l != 0);
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
// Bin split for windows of more than 2^15 buckets (c > 16), where one window's counters no longer fit the LDS: two
// LDS-staged passes over 8-byte records (round 5; rounds 3-4 ran three passes over pairs of 4-byte arrays plus a histogram
// pass before and a chunk-ordering pass behind them).  The bucket index l - 1 of window kk is cut on its EFFECTIVE bits
// into | ab coarse | fb fine | (WinSplit; ab <= 11, fb <= 12):
//   k_digits / k_te_digits : digits, and per slice of the points the histogram of the coarse bins (fused: msm_kernels.h)
//   k_colscan, k_vscan     : per (window, bin) the prefix over the slices; the bin starts
//   k_bin_split  (pass A)  : slice (b, kk) ranks 16 k-entry tiles by coarse bin in the LDS and copies them out as runs of
//                            records (fine bits + 1 | sign << 31, entry index) -- one 8-byte store stream
//   k_bin_count            : one block per bin: the bucket sizes (the scans of the padded slots need them)
//   k_bin_pairs  (pass B)  : one block per bin walks its records in tiles of 4 k and emits the PAIRS round 1 of the tree adds,
//                            tile by tile, each with the element index its sum belongs to (`dest`).  A bin's records are in
//                            point order, so a tile -- and with it every run of 64 consecutive pairs -- gathers from one
//                            ~0.5 GB range of the point rows: the locality the scattered row reads of round 1 need
//                            (tools/ubench_gather2.hip: 30 G lines/s inside 1 GB per wave instruction, 13 beyond), which
//                            rounds 3-4 bought with a separate k_chunk_order pass over the bucket-sorted slots.  The list
//                            leaves as full sequential lines; nothing is scattered by the sort itself any more.
//   k_bin_slots  (pass B') : the plain form for small inputs -- payloads to their padded slots, bucket order
// A window of at most 10 effective bits (a short top window) is bucket-sorted by pass A alone (fb = 0): all blocks of
// pass B then share it by position.
// ---------------------------------------------------------------------------------------------

// exclusive scan over NB per-bin values val(b) by a block of THREADS: thread t owns the bins [t * per, (t + 1) * per); calls
// put(b, prefix) in order; returns the total
template <int THREADS, bool LDS_ONLY = false, class Val, class Put>
__device__ __forceinline__ uint32_t bucket_scan(uint32_t NB, uint32_t* lds_wave, Val val, Put put) {
  const uint32_t per = (NB + THREADS - 1) / THREADS, b0 = threadIdx.x * per;
  uint32_t sum = 0;
  for (uint32_t j = 0; j < per; j++)
    if (b0 + j < NB) sum += val(b0 + j);
  uint32_t tot;
  uint32_t ex = block_excl_scan<LDS_ONLY>(sum, lds_wave, tot);
  for (uint32_t j = 0; j < per; j++)
    if (b0 + j < NB) {
      const uint32_t x = val(b0 + j);
      put(b0 + j, ex);
      ex += x;
    }
  return tot;
}

// pass A geometry (tools/ab_kernels.sh, profiles/r05_experiments.txt items 2 and 10).  The pass is bound by the NUMBER of runs it
// writes, not by their bytes: a tile leaves tile / 2^ab records per bin, one run each, at 3 k places of the record array per
// block and 0.8 M over the chip -- every run opens another DRAM page.  Measured per window group at 2^26 (403 M entries):
//   16 k tiles, 2^10 bins (128-byte runs)   2.1 - 2.4 ms      the same with 2^9 bins (256-byte runs) 1.7
//   ... with every run cut at 64-byte units (partial units held back in the LDS)   2.4: partial writes are not it
//   ... with barriers that do not wait for the stores to drain                      2.35: nor the fences
//   32 k tiles (256-byte runs), records staged in 4 bytes                           see item 10
// 512 threads x 16 or x 24 with two workgroups per CU: 2.4 / 2.13; the next tile's digits requested early: no change.
constexpr int BS_THREADS = 1024, BS_ITEMS = 32, BS_TILE = BS_THREADS * BS_ITEMS;
// k_colscan for the bin split's slice histograms: few columns (the coarse bins of the group), many rows (one per slice -- on
// window tables kc times as many).  k_colscan walks a column with one thread: B latencies in sequence (0.5 ms for 8 192 rows).
// Here 32 row lanes share a column: each sums a contiguous range of the rows, the 32 sums are scanned in the LDS, and each lane
// walks its range a second time writing the exclusive prefixes.  grid (ceil(L / 32), k_cnt), 1024 threads.
__global__ void __launch_bounds__(1024) k_slice_scan(uint32_t* block_hist, uint32_t* totals, uint32_t B, uint32_t L, uint32_t k_cnt)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint32_t part[32][33];
  const uint32_t c = threadIdx.x & 31u, r = threadIdx.x >> 5, kk = blockIdx.y;
  const uint32_t l = blockIdx.x * 32 + c;
  const bool live = l < L;
  const uint32_t b0 = (uint32_t)((uint64_t)B * r / 32), b1 = (uint32_t)((uint64_t)B * (r + 1) / 32);
  uint32_t* p = block_hist + (uint64_t)kk * B * L + (live ? l : 0);
  uint32_t sum = 0;
  if (live) {
    uint32_t b = b0;
    for (; b + 8 <= b1; b += 8) {
      uint32_t v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = p[(uint64_t)(b + i) * L];
#pragma unroll
      for (int i = 0; i < 8; i++) sum += v[i];
    }
    for (; b < b1; b++) sum += p[(uint64_t)b * L];
  }
  part[r][c] = sum;
  __syncthreads();
  uint32_t run = 0;
  for (uint32_t q = 0; q < r; q++) run += part[q][c];
  if (live) {
    uint32_t b = b0;
    for (; b + 8 <= b1; b += 8) {
      uint32_t v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = p[(uint64_t)(b + i) * L];
#pragma unroll
      for (int i = 0; i < 8; i++) { p[(uint64_t)(b + i) * L] = run; run += v[i]; }
    }
    for (; b < b1; b++) {
      const uint32_t v = p[(uint64_t)b * L];
      p[(uint64_t)b * L] = run;
      run += v;
    }
    if (r == 31) totals[(uint64_t)kk * L + l] = run;
  }
}
#endif

constexpr uint32_t BS_MAX_AB = 11, BS_MAX_FB = 12;
// a block's entries are named by their offset from its first one inside the LDS: BS_SPAN_LOG bits (the host cuts the slices so)
constexpr uint32_t BS_SPAN_LOG = 17;
inline size_t bin_split_lds(uint32_t hb) { return (size_t)BS_TILE * 4 + (size_t)3 * hb * 4 + 64 * 4; }

// pass A.  grid (SB, kc); block (b, kk) owns entries [b * chunk, (b + 1) * chunk) of window kk.  Coarse bin h of window kk
// starts at bin_start[kk * hb + h]; this slice's share of it slice_off[(kk * SB + b) * hb + h] further.
// merged (window tables: the kc digit windows of the group are ONE window of kc * two_n entries for the sort): every block
// works for window 0, its slice is number kk * SB + b of that window, and entry indices count from the group's first digit.
// The histograms come per slice of the DIGIT kernel (SBd of them per window, `chunk` entries each); a block of this pass takes
// g consecutive ones, so that small inputs -- whose digit kernel still wants a block per 4 096 points -- sort whole tiles.
// The LDS holds a record in 4 bytes -- fine bits + 1 (13 bits) | offset of the entry from the block's first one << 13 (17 bits)
// | sign << 31 -- so that a tile is 32 k records; the copy-out walks the bins, 32 lanes each (a staged position no longer
// says which bin it belongs to), and widens the records to the (fine bits + 1 | sign, entry index) pairs the next pass reads.
__global__ void __launch_bounds__(BS_THREADS) k_bin_split(uint2* rec, const uint32_t* bin_start, const uint32_t* slice_off,
                                                          const uint32_t* dig, uint64_t two_n, uint64_t chunk, uint32_t hb, WinSplit ws,
                                                          uint32_t merged, uint32_t SBd, uint32_t g)
#ifndef MSM_SORT_TU
    ;
#else
{
  extern __shared__ uint32_t lds_bs[];
  uint32_t* stage = lds_bs;                               // BS_TILE packed records, bin by bin
  uint32_t* t_cnt = stage + BS_TILE;
  uint32_t* t_start = t_cnt + hb;
  uint32_t* g_base = t_start + hb;                        // where the bin's next record goes
  uint32_t* lds_wave = g_base + hb;                       // 64 words
  const uint32_t b = blockIdx.x, seg = blockIdx.y, tid = threadIdx.x;
  const uint32_t kk = merged ? 0u : seg;
  const uint32_t ab = ws.ab[kk], fb = ws.fb[kk], HN = 1u << ab;
  for (uint32_t h = tid; h < HN; h += BS_THREADS)
    g_base[h] = bin_start[(uint64_t)kk * hb + h] + slice_off[((uint64_t)seg * SBd + (uint64_t)b * g) * hb + h];
  const uint64_t beg = (uint64_t)b * g * chunk, end = min(beg + (uint64_t)g * chunk, two_n);
  const uint32_t* d = dig + (uint64_t)seg * two_n;
  const uint32_t e_first = (merged ? (uint32_t)((uint64_t)seg * two_n) : 0u) + (uint32_t)beg;   // entry index of offset 0
  const uint32_t fmask = (1u << fb) - 1;
  uint32_t v[BS_ITEMS];
  // the tile's digits through a buffer descriptor over [t0, t0 + left): one 32-bit lane offset for all 32 loads (64-bit
  // addresses for them would spill), and what lies behind the slice reads as 0 = "no entry" without a compare per load
  auto load_tile = [&](uint64_t t0) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t left = t0 < end ? (uint32_t)min<uint64_t>(end - t0, BS_TILE) : 0u;
    const uint64_t a = (uint64_t)(d + t0);
    const uint32_t a_lo = __builtin_amdgcn_readfirstlane((uint32_t)a), a_hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)a_hi << 32) | a_lo), 0, __builtin_amdgcn_readfirstlane(left * 4u), 0x00020000);
#pragma unroll
    for (int i = 0; i < BS_ITEMS; i++) v[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, tid * 4u, i * BS_THREADS * 4, 0);
#endif
  };
  load_tile(beg);
  for (uint64_t t0 = beg; t0 < end; t0 += BS_TILE) {
    for (uint32_t h = tid; h < HN; h += BS_THREADS) t_cnt[h] = 0;
    lds_barrier();
    uint32_t rk[BS_ITEMS / 2];   // two ranks (< 2^15) per register: 32 digits and 32 ranks would not fit 128 registers
#pragma unroll
    for (int i = 0; i < BS_ITEMS; i++) {
      const uint32_t l = v[i] & 0x7FFFFFFFu;
      const uint32_t r = lds_rank_add(t_cnt, l ? (l - 1) >> fb : 0u, l != 0);
      if (i & 1) rk[i / 2] |= r << 16; else rk[i / 2] = r;
#if defined(__HIP_DEVICE_COMPILE__)
      if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // (eight in flight: all 32 with their temporaries would spill)
#endif
    }
    lds_barrier();
    (void)bucket_scan<BS_THREADS, true>(HN, lds_wave, [&](uint32_t h) { return t_cnt[h]; }, [&](uint32_t h, uint32_t ex) { t_start[h] = ex; });
    lds_barrier();
#if defined(__HIP_DEVICE_COMPILE__)
    // (the staging below derives the bin of a digit once more: kept alive from the ranking, 64 such values were spilled)
#pragma unroll
    for (int i = 0; i < BS_ITEMS; i++) asm volatile("" : "+v"(v[i]));
#endif
    const uint32_t off0 = (uint32_t)(t0 - beg) + tid;
#pragma unroll
    for (int i = 0; i < BS_ITEMS; i++) {
      const uint32_t l = v[i] & 0x7FFFFFFFu;
      if (l) stage[t_start[(l - 1) >> fb] + ((rk[i / 2] >> (16 * (i & 1))) & 0xFFFFu)] = (((l - 1) & fmask) + 1) | ((off0 + (uint32_t)i * BS_THREADS) << 13) | (v[i] & 0x80000000u);
#if defined(__HIP_DEVICE_COMPILE__)
      if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
#endif
    }
    lds_barrier();
    load_tile(t0 + BS_TILE);   // in flight during the copy-out
    const uint32_t lane = tid & 31u;
    for (uint32_t h = tid >> 5; h < HN; h += BS_THREADS / 32) {
      const uint32_t T = t_cnt[h], pos = g_base[h], ts = t_start[h];
      for (uint32_t o = lane; o < T; o += 32) {
        const uint32_t r = stage[ts + o];
        rec[pos + o] = make_uint2(r & 0x80001FFFu, e_first + ((r >> 13) & ((1u << BS_SPAN_LOG) - 1)));
      }
      if (lane == 0) g_base[h] = pos + T;
    }
    lds_barrier();
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// Parts of heavy bins (round 6).  The count pass and pass B give one block to a bin, which is right while the digits are
// spread -- a bin is 1 / 2^10 of a window -- and a bound on nothing when they are not: one scalar repeated puts every entry of
// a window into ONE bin (2^27 records for one block at 2^26 points: the reference walks any bucket-size distribution through the
// same loop, src/msm-batched-affine.ts:204,243-263).  A bin of more than `part_len` records (twice the mean bin, at least 2^16)
// is therefore cut into parts of part_len records, one block each:
//   * block p < V works for bin p's first part as before; blocks V .. V + extra_first[V] take the further parts of the bins
//     that have any (binary search in extra_first); the host launches V + V / 2 blocks, an upper bound without a read-back
//     (sum of ceil(size / part_len) <= V + entries / part_len <= V + V / 2);
//   * the count pass leaves the bucket sizes of a part of a multi-part bin in a row of `sub` (mp_first[v] + j) instead of
//     `counts`; k_part_scan turns the rows of a bin into exclusive prefixes over its parts and writes the bin's `counts`;
//   * pairs (k_bin_pairs): entries pair up inside their part only -- an odd one out is paired with nothing when its part ends --
//     so a bucket's round-1 elements are the concatenation of ceil(n_part / 2) per part: `counts` gets 2 * that sum (the slots a
//     bucket is padded from), the prefix is in elements, and part_pair_off[row] says where in the bin's run of the pair list the
//     part starts;  slots (k_bin_slots): the prefix is in entries and `counts` is the true size.
// The part table comes out of k_vscan (it has the bin sizes in hand).  Uniform digits never have a multi-part bin: one extra
// launch (k_part_scan, every block returns at once) and V / 2 blocks of the two passes that return at once.
// ---------------------------------------------------------------------------------------------
struct PartLoc {
  uint32_t v, j, np, row;   // bin, part of the bin, parts of the bin, row of `sub` (np > 1)
  uint64_t beg, end;        // the part's records
  bool valid;
};
// block p of a parts grid (see above); reverse: the first V blocks take the bins from the last (the heaviest: a short top window's)
__device__ __forceinline__ PartLoc locate_part(uint32_t p, uint32_t V, const uint32_t* bin_start, const uint32_t* extra_first,
                                               const uint32_t* mp_first, uint32_t part_len, bool reverse) {
  PartLoc L;
  L.valid = true;
  if (p < V) {
    L.v = reverse ? V - 1 - p : p;
    L.j = 0;
  } else {
    const uint32_t q = p - V;
    if (q >= extra_first[V]) { L.valid = false; L.v = L.j = L.np = L.row = 0; L.beg = L.end = 0; return L; }
    uint32_t lo = 0, hi = V;   // last bin whose extra parts start at or before q
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (extra_first[mid] <= q) lo = mid; else hi = mid;
    }
    L.v = lo;
    L.j = 1 + q - extra_first[lo];
  }
  L.np = extra_first[L.v + 1] - extra_first[L.v] + 1;
  L.row = mp_first[L.v] + L.j;
  const uint64_t b0 = bin_start[L.v], b1 = bin_start[L.v + 1];
  L.beg = b0 + (uint64_t)L.j * part_len;
  L.end = L.np == 1 ? b1 : min(b1, L.beg + (uint64_t)part_len);
  return L;
}

// Bins of several parts: sub[row of part j][bucket] = the part's entries of the bucket (k_bin_count) -> the exclusive prefix over
// the parts, in round-1 elements (pairs_mode: ceil(n / 2) per part) or in entries; counts[bucket] = 2 * the sum of the parts'
// elements, or the true size; part_pair_off[row] = pairs the parts before it emit (zeroed before the launch); the largest bucket
// and the algorithmic pair additions of these bins go to `info` as k_bin_count leaves them for the others.  One block per bin.
constexpr int PSC_THREADS = 1024;
__global__ void __launch_bounds__(PSC_THREADS) k_part_scan(uint32_t* counts, uint32_t* sub, uint32_t* part_pair_off, const uint32_t* extra_first,
                                                           const uint32_t* mp_first, uint32_t hb, uint32_t L, uint32_t nbmax, WinSplit ws,
                                                           uint32_t* info, uint32_t pairs_mode)
#ifndef MSM_SORT_TU
    ;
#else
{
  __shared__ uint32_t lds_wave[PSC_THREADS / 64];
  __shared__ uint32_t lds_max;
  __shared__ unsigned long long lds_sum;
  const uint32_t v = blockIdx.x, tid = threadIdx.x;
  const uint32_t np = extra_first[v + 1] - extra_first[v] + 1;
  if (np == 1) return;
  const uint32_t kk = v / hb, h = v - kk * hb, fb = ws.fb[kk], NB = 1u << fb, row0 = mp_first[v];
  uint32_t* out = counts + (uint64_t)kk * L + ((uint64_t)h << fb);
  if (tid == 0) { lds_max = 0; lds_sum = 0; }
  __syncthreads();
  constexpr int IT = ((1 << BS_MAX_FB) + PSC_THREADS - 1) / PSC_THREADS;   // buckets per thread
  uint32_t run[IT], tot[IT];
#pragma unroll
  for (int i = 0; i < IT; i++) run[i] = tot[i] = 0;
  for (uint32_t j = 0; j < np; j++) {
    uint32_t* r = sub + (uint64_t)(row0 + j) * nbmax;
    uint32_t n[IT], pp = 0;
#pragma unroll
    for (int i = 0; i < IT; i++) { const uint32_t bk = tid + i * PSC_THREADS; n[i] = bk < NB ? r[bk] : 0u; }
#pragma unroll
    for (int i = 0; i < IT; i++) {
      const uint32_t bk = tid + i * PSC_THREADS, el = pairs_mode ? (n[i] + 1) >> 1 : n[i];
      if (bk < NB) r[bk] = run[i];
      run[i] += el;
      tot[i] += n[i];
      pp += (n[i] + 1) >> 1;
    }
    if (pairs_mode) {
      for (int o = 32; o > 0; o >>= 1) pp += __shfl_down(pp, o);
      if ((tid & 63u) == 0 && pp) atomicAdd(&part_pair_off[row0 + j], pp);
    }
  }
  uint32_t mx = 0;
  unsigned long long sum = 0;
#pragma unroll
  for (int i = 0; i < IT; i++) {
    const uint32_t bk = tid + i * PSC_THREADS;
    if (bk < NB) {
      const uint32_t cnt = pairs_mode ? 2 * run[i] : tot[i];
      out[bk] = cnt;
      mx = max(mx, cnt);
      sum += tot[i] ? tot[i] - 1 : 0;
    }
  }
  for (int o = 32; o > 0; o >>= 1) { sum += __shfl_down(sum, o); mx = max(mx, (uint32_t)__shfl_down(mx, o)); }
  if ((tid & 63u) == 0 && (mx | sum)) { atomicMax(&lds_max, mx); atomicAdd(&lds_sum, sum); }
  __threadfence();
  __syncthreads();
  if (tid == 0) {
    atomicMax(&info[1], lds_max);
    atomicAdd(reinterpret_cast<unsigned long long*>(info + INFO_ALGO_PAIRS), lds_sum);
  }
  if (!pairs_mode) return;
  // pairs per part -> where the part starts in the bin's run of the pair list
  uint32_t carry = 0;
  for (uint32_t base = 0; base < np; base += PSC_THREADS) {
    const uint32_t j = base + tid;
    const uint32_t x = j < np ? __hip_atomic_load(&part_pair_off[row0 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    uint32_t t;
    const uint32_t ex = block_excl_scan(x, lds_wave, t) + carry;
    __syncthreads();
    if (j < np) part_pair_off[row0 + j] = ex;
    carry += t;
  }
}
#endif

// bucket sizes of bin v = kk * hb + h (the 2^fb buckets from h << fb of window kk) from its records; `counts` is zeroed before.
// 1024 threads and two records per 16-byte load: with 256 threads the 3 072 bins of a 21-bit window group were 1.5 rounds of
// the 2 048 blocks the chip holds and every lane load was half a request (0.92 -> see profiles/r05_experiments.txt item 10).
constexpr int BC_THREADS = 1024;
// Also what k_bucket_max does on the other sort paths: the largest bucket -> info[1], sum of (size - 1) -> info[40..41] (`info`
// is zeroed before; a bin without records writes its zeros itself, so `counts` needs no fill).
// Grid: V + V / 2 blocks, one per part (see "Parts of heavy bins"); a part of a multi-part bin leaves its sizes in its row of `sub`.
__global__ void __launch_bounds__(BC_THREADS) k_bin_count(uint32_t* counts, const uint32_t* bin_start, const uint2* rec, uint32_t hb, uint32_t L,
                                                          WinSplit ws, uint32_t* info, uint32_t V, const uint32_t* extra_first,
                                                          const uint32_t* mp_first, uint32_t part_len, uint32_t* sub, uint32_t nbmax)
#ifndef MSM_SORT_TU
    ;
#else
{
  extern __shared__ uint32_t lds_bc[];
  __shared__ uint32_t lds_max;
  __shared__ unsigned long long lds_sum;
  const uint32_t tid = threadIdx.x;
  const PartLoc pl = locate_part(blockIdx.x, V, bin_start, extra_first, mp_first, part_len, false);
  if (!pl.valid) return;
  const uint32_t v = pl.v;
  const uint32_t kk = v / hb, h = v - kk * hb, fb = ws.fb[kk], NB = 1u << fb;
  if (h >= (1u << ws.ab[kk])) return;
  const uint64_t beg = pl.beg, end = pl.end;
  uint32_t* out = counts + (uint64_t)kk * L + ((uint64_t)h << fb);
  if (beg == end) {
    for (uint32_t j = tid; j < NB; j += BC_THREADS) out[j] = 0;
    return;
  }
  if (fb == 0) {   // one bucket
    if (tid == 0) {
      out[0] = (uint32_t)(end - beg);
      atomicMax(&info[1], (uint32_t)(end - beg));
      atomicAdd(reinterpret_cast<unsigned long long*>(info + INFO_ALGO_PAIRS), (unsigned long long)(end - beg - 1));
    }
    return;
  }
  if (tid == 0) { lds_max = 0; lds_sum = 0; }
  for (uint32_t j = tid; j < NB; j += BC_THREADS) lds_bc[j] = 0;
  __syncthreads();
  const uint4* rec2 = reinterpret_cast<const uint4*>(rec);   // records 2 j and 2 j + 1
  for (uint64_t j0 = beg >> 1; 2 * j0 < end; j0 += BC_THREADS) {   // whole waves stay in the loop (lds_rank_add looks at the whole wave)
    const uint64_t j = j0 + tid;
    uint4 r = make_uint4(0, 0, 0, 0);
    if (2 * j < end) r = rec2[j];   // (the record array is allocated with slack beyond its last record)
    const uint32_t l0 = (2 * j >= beg && 2 * j < end) ? r.x & 0xFFFFu : 0u;
    const uint32_t l1 = (2 * j + 1 >= beg && 2 * j + 1 < end) ? r.z & 0xFFFFu : 0u;
    (void)lds_rank_add(lds_bc, l0 ? l0 - 1 : 0u, l0 != 0);
    (void)lds_rank_add(lds_bc, l1 ? l1 - 1 : 0u, l1 != 0);
  }
  __syncthreads();
  if (pl.np > 1) {   // uniform: the bin's counts, largest bucket and pair additions come from k_part_scan
    uint32_t* row = sub + (uint64_t)pl.row * nbmax;
    for (uint32_t j = tid; j < NB; j += BC_THREADS) row[j] = lds_bc[j];
    return;
  }
  uint32_t mx = 0;
  unsigned long long sum = 0;
  for (uint32_t j = tid; j < NB; j += BC_THREADS) {
    const uint32_t cnt = lds_bc[j];
    out[j] = cnt;
    mx = max(mx, cnt);
    sum += cnt ? cnt - 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) { sum += __shfl_down(sum, o); mx = max(mx, (uint32_t)__shfl_down(mx, o)); }
  if ((tid & 63u) == 0 && (mx | sum)) { atomicMax(&lds_max, mx); atomicAdd(&lds_sum, sum); }
  __syncthreads();
  if (tid == 0) {
    atomicMax(&info[1], lds_max);
    atomicAdd(reinterpret_cast<unsigned long long*>(info + INFO_ALGO_PAIRS), lds_sum);
  }
}
#endif

// pass B.  One block per bin v = kk * hb + h (heaviest -- a short top window's -- first: v = V - 1 - blockIdx.x).  `cursor` holds
// the padded slot offset of every bucket of the group and, at [nb], the total; a bucket of n entries owns roundup(n, G) / 2
// consecutive pairs of round 1, i.e. consecutive elements of its output.
constexpr int BP_THREADS = 512, BP_ITEMS = 8, BP_TILE = BP_THREADS * BP_ITEMS;   // 4 096 records per tile
inline size_t bin_pairs_lds(uint32_t nbmax) { return (size_t)5 * nbmax * 4 + 64 * 4 + (size_t)((BP_TILE + nbmax) / 2 + 1) * 12; }
inline size_t bin_slots_lds(uint32_t nbmax) { return (size_t)3 * nbmax * 4 + 64 * 4 + (size_t)BP_TILE * 6; }

__global__ void __launch_bounds__(BP_THREADS) k_bin_pairs(uint2* pairs, uint32_t* dest, const uint2* rec, const uint32_t* bin_start,
                                                          const uint32_t* cursor, uint32_t hb, uint32_t L, uint32_t nbmax, WinSplit ws,
                                                          uint32_t V, const uint32_t* extra_first, const uint32_t* mp_first,
                                                          uint32_t part_len, const uint32_t* sub, const uint32_t* part_pair_off)
#ifndef MSM_SORT_TU
    ;
#else
{
  extern __shared__ uint32_t lds_bp[];
  const uint32_t st_cap = (BP_TILE + nbmax) / 2 + 1;   // pairs one tile can emit
  uint2* st_pair = reinterpret_cast<uint2*>(lds_bp);
  uint32_t* st_dest = lds_bp + 2 * st_cap;
  uint32_t* t_cnt = st_dest + st_cap;  // entries of the tile per bucket
  uint32_t* p_start = t_cnt + nbmax;   // first pair of the bucket in the tile's output | "had a pending entry" << 31
  uint32_t* t_delta = p_start + nbmax; // element index of the pair staged at p = t_delta + p
  uint32_t* pend = t_delta + nbmax;    // the odd entry a bucket carries into the next tile (SLOT_EMPTY: none)
  uint32_t* g_next = pend + nbmax;     // next element of the bucket
  uint32_t* lds_wave = g_next + nbmax; // 64 words
  const uint32_t tid = threadIdx.x;
  const PartLoc pl = locate_part(blockIdx.x, V, bin_start, extra_first, mp_first, part_len, true);
  if (!pl.valid) return;
  const uint32_t v = pl.v;
  const uint32_t kk = v / hb, h = v - kk * hb;
  const uint32_t ab = ws.ab[kk], fb = ws.fb[kk], NB = 1u << fb, HN = 1u << ab;
  const uint32_t* cur = cursor + (uint64_t)kk * L;
  if (fb == 0) {
    // the window is bucket-sorted already (bucket = bin): its hb blocks share its pairs by position, in bucket order
    const uint32_t* bs = bin_start + (uint64_t)kk * hb;
    const uint64_t d0 = cur[0] >> 1, d1 = cur[HN] >> 1, len = d1 - d0;
    const uint64_t beg = d0 + len * h / hb, end = d0 + len * (h + 1) / hb;
    for (uint64_t dd = beg + tid; dd < end; dd += BP_THREADS) {
      uint32_t lo = 0, hi = HN;                                    // last bucket starting at or before element dd
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((cur[mid] >> 1) <= dd) lo = mid; else hi = mid;
      }
      const uint32_t q = (uint32_t)(dd - (cur[lo] >> 1)), nent = bs[lo + 1] - bs[lo];
      const uint64_t r0 = (uint64_t)bs[lo] + 2ull * q;
      uint2 pr = make_uint2(SLOT_EMPTY, SLOT_EMPTY);
      if (2 * q < nent) { const uint2 r = rec[r0]; pr.x = (r.y << 1) | (r.x >> 31); }
      if (2 * q + 1 < nent) { const uint2 r = rec[r0 + 1]; pr.y = (r.y << 1) | (r.x >> 31); }
      pairs[dd] = pr;
      dest[dd] = (uint32_t)dd;
    }
    return;
  }
  if (h >= HN) return;
  const uint32_t* curb = cur + ((uint64_t)h << fb);               // this bin's buckets
  const uint64_t beg = pl.beg, end = pl.end;
  if (beg == end) return;                                          // no entries, hence no slots and no pairs
  // (a part of a multi-part bin: its buckets' elements start behind those of the parts before it, and so do its pairs)
  const uint32_t* sub_row = pl.np > 1 ? sub + (uint64_t)pl.row * nbmax : nullptr;
  for (uint32_t bk = tid; bk < NB; bk += BP_THREADS) {
    pend[bk] = SLOT_EMPTY;
    g_next[bk] = (curb[bk] >> 1) + (sub_row ? sub_row[bk] : 0u);
  }
  uint64_t out_pos = (curb[0] >> 1) + (sub_row ? part_pair_off[pl.row] : 0u);   // the bin's pairs are consecutive in the list
  uint2 r[BP_ITEMS], nr[BP_ITEMS];
  auto load_tile = [&](uint2 (&dst)[BP_ITEMS], uint64_t t0) {
#pragma unroll
    for (int i = 0; i < BP_ITEMS; i++) {
      const uint64_t j = t0 + (uint64_t)i * BP_THREADS + tid;
      dst[i] = j < end ? rec[j] : make_uint2(0u, 0u);
    }
  };
  load_tile(r, beg);
  for (uint64_t t0 = beg; t0 < end; t0 += BP_TILE) {
    for (uint32_t bk = tid; bk < NB; bk += BP_THREADS) t_cnt[bk] = 0;
    lds_barrier();
    uint32_t rk[BP_ITEMS];
#pragma unroll
    for (int i = 0; i < BP_ITEMS; i++) {
      const uint32_t l = r[i].x & 0xFFFFu;
      rk[i] = lds_rank_add(t_cnt, l ? l - 1 : 0u, l != 0);
    }
    load_tile(nr, t0 + BP_TILE);   // the next tile's records are in flight while this one is paired and copied out
    __syncthreads();
    // per bucket: m = pending + the tile's entries -> m / 2 pairs now, m & 1 entries pending
    const uint32_t n_pairs = bucket_scan<BP_THREADS>(
        NB, lds_wave, [&](uint32_t bk) { return ((pend[bk] != SLOT_EMPTY ? 1u : 0u) + t_cnt[bk]) >> 1; },
        [&](uint32_t bk, uint32_t ex) {
          const uint32_t has = pend[bk] != SLOT_EMPTY ? 1u : 0u, m = has + t_cnt[bk], np = m >> 1;
          p_start[bk] = ex | (has << 31);
          t_delta[bk] = g_next[bk] - ex;
          g_next[bk] += np;
          if (has && np) {                 // the pending entry opens the bucket's first pair of this tile
            st_pair[ex].x = pend[bk];
            st_dest[ex] = t_delta[bk] + ex;
            pend[bk] = SLOT_EMPTY;
          }
        });
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BP_ITEMS; i++) {
      const uint32_t l = r[i].x & 0xFFFFu;
      if (l) {
        const uint32_t bk = l - 1, ps = p_start[bk], has = ps >> 31, m = has + t_cnt[bk], sq = rk[i] + has;
        const uint32_t payload = (r[i].y << 1) | (r[i].x >> 31);
        if ((m & 1u) && sq == m - 1) pend[bk] = payload;           // the odd one out waits for the next tile
        else {
          const uint32_t p = (ps & 0x7FFFFFFFu) + (sq >> 1);
          if (sq & 1u) st_pair[p].y = payload;
          else { st_pair[p].x = payload; st_dest[p] = t_delta[bk] + p; }
        }
      }
    }
    __syncthreads();
    for (uint32_t p = tid; p < n_pairs; p += BP_THREADS) {
      pairs[out_pos + p] = st_pair[p];
      dest[out_pos + p] = st_dest[p];
    }
    out_pos += n_pairs;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BP_ITEMS; i++) r[i] = nr[i];
  }
  if (pl.j + 1 < pl.np) {
    // not the bin's last part: an odd one out is paired with nothing here (the bucket's entries pair up inside a part)
    (void)bucket_scan<BP_THREADS>(
        NB, lds_wave, [&](uint32_t bk) { return pend[bk] != SLOT_EMPTY ? 1u : 0u; },
        [&](uint32_t bk, uint32_t ex) {
          if (pend[bk] != SLOT_EMPTY) {
            pairs[out_pos + ex] = make_uint2(pend[bk], SLOT_EMPTY);
            dest[out_pos + ex] = g_next[bk];
          }
        });
    return;
  }
  // what every bucket still owes: its pending entry paired with nothing, then pairs of pads (their sums are identities
  // that the index-free rounds behind round 1 read)
  (void)bucket_scan<BP_THREADS>(
      NB, lds_wave, [&](uint32_t bk) { return (curb[bk + 1] >> 1) - g_next[bk]; },
      [&](uint32_t bk, uint32_t ex) {
        const uint32_t rem = (curb[bk + 1] >> 1) - g_next[bk];
        for (uint32_t j = 0; j < rem; j++) {
          pairs[out_pos + ex + j] = make_uint2(j == 0 ? pend[bk] : SLOT_EMPTY, SLOT_EMPTY);
          dest[out_pos + ex + j] = g_next[bk] + j;
        }
      });
}
#endif

// pass B': payloads (entry << 1 | sign) to their padded slots, bucket order (slots are pre-filled with SLOT_EMPTY)
__global__ void __launch_bounds__(BP_THREADS) k_bin_slots(uint32_t* slots, const uint2* rec, const uint32_t* bin_start,
                                                          const uint32_t* cursor, uint32_t hb, uint32_t L, uint32_t nbmax, WinSplit ws,
                                                          uint32_t V, const uint32_t* extra_first, const uint32_t* mp_first,
                                                          uint32_t part_len, const uint32_t* sub)
#ifndef MSM_SORT_TU
    ;
#else
{
  extern __shared__ uint32_t lds_bp[];
  uint32_t* t_cnt = lds_bp;
  uint32_t* t_start = t_cnt + nbmax;
  uint32_t* g_cur = t_start + nbmax;
  uint32_t* lds_wave = g_cur + nbmax;
  uint32_t* stage = lds_wave + 64;
  uint16_t* stage_b = reinterpret_cast<uint16_t*>(stage + BP_TILE);
  const uint32_t tid = threadIdx.x;
  const PartLoc pl = locate_part(blockIdx.x, V, bin_start, extra_first, mp_first, part_len, true);
  if (!pl.valid) return;
  const uint32_t v = pl.v;
  const uint32_t kk = v / hb, h = v - kk * hb;
  const uint32_t ab = ws.ab[kk], fb = ws.fb[kk], NB = 1u << fb, HN = 1u << ab;
  const uint32_t* cur = cursor + (uint64_t)kk * L;
  if (fb == 0) {
    const uint32_t* bs = bin_start + (uint64_t)kk * hb;
    const uint64_t wbeg = bs[0], wlen = (uint64_t)bs[HN] - wbeg;
    const uint64_t beg = wbeg + wlen * h / hb, end = wbeg + wlen * (h + 1) / hb;
    for (uint64_t j = beg + tid; j < end; j += BP_THREADS) {
      uint32_t lo = 0, hi = HN;                                    // bucket of record j: last bin starting at or before it
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (bs[mid] <= j) lo = mid; else hi = mid;
      }
      const uint2 r = rec[j];
      slots[cur[lo] + (uint32_t)(j - bs[lo])] = (r.y << 1) | (r.x >> 31);
    }
    return;
  }
  if (h >= HN) return;
  const uint64_t beg = pl.beg, end = pl.end;
  if (beg == end) return;
  // (a part of a multi-part bin writes behind the entries the parts before it hold of every bucket)
  const uint32_t* sub_row = pl.np > 1 ? sub + (uint64_t)pl.row * nbmax : nullptr;
  for (uint32_t bk = tid; bk < NB; bk += BP_THREADS) g_cur[bk] = cur[((uint64_t)h << fb) + bk] + (sub_row ? sub_row[bk] : 0u);
  for (uint64_t t0 = beg; t0 < end; t0 += BP_TILE) {
    for (uint32_t bk = tid; bk < NB; bk += BP_THREADS) t_cnt[bk] = 0;
    lds_barrier();
    uint2 r[BP_ITEMS];
    uint32_t rk[BP_ITEMS];
#pragma unroll
    for (int i = 0; i < BP_ITEMS; i++) {
      const uint64_t j = t0 + (uint64_t)i * BP_THREADS + tid;
      r[i] = j < end ? rec[j] : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < BP_ITEMS; i++) {
      const uint32_t l = r[i].x & 0xFFFFu;
      rk[i] = lds_rank_add(t_cnt, l ? l - 1 : 0u, l != 0);
    }
    lds_barrier();
    const uint32_t n_tile = bucket_scan<BP_THREADS, true>(NB, lds_wave, [&](uint32_t bk) { return t_cnt[bk]; },
                                           [&](uint32_t bk, uint32_t ex) { t_start[bk] = ex; });
    lds_barrier();
#pragma unroll
    for (int i = 0; i < BP_ITEMS; i++) {
      const uint32_t l = r[i].x & 0xFFFFu;
      if (l) {
        const uint32_t p = t_start[l - 1] + rk[i];
        stage[p] = (r[i].y << 1) | (r[i].x >> 31);
        stage_b[p] = (uint16_t)(l - 1);
      }
    }
    lds_barrier();
    for (uint32_t i = tid; i < n_tile; i += BP_THREADS) {
      const uint32_t bk = stage_b[i];
      slots[g_cur[bk] + (i - t_start[bk])] = stage[i];
    }
    lds_barrier();
    for (uint32_t bk = tid; bk < NB; bk += BP_THREADS) g_cur[bk] += t_cnt[bk];
  }
}
#endif

}  // namespace msm
