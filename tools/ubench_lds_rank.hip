// What does ranking a tile of keys into LDS counters cost on gfx950?  The sort's passes (sort_kernels.h) rank every entry of a
// tile by its bin with a returning LDS atomic; the guide's LDS table has no row for DS atomics.  Variants, all on random
// bins, one workgroup of 1024 threads per CU (and two of 512), ITEMS keys per thread and tile:
//   add      ds_add_u32       (no return: the histogram fused into k_digits)
//   rtn      ds_add_rtn_u32   (rank = old value: k_bin_split, k_bin_pairs)
//   ballot   lds_rank_add     (lanes that share a bin find each other with one ballot per bin bit, one atomic per group)
//   wavepriv per-wave counter tables, match by ballot, the group's leader does a plain read-modify-write (no atomics)
// usage: ubench_lds_rank [bins_log2 = 10]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__device__ __forceinline__ uint32_t lds_rank_add(uint32_t* lds, uint32_t bin, bool valid, uint32_t bits) {
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

template <int MODE, int ITEMS>
__global__ void __launch_bounds__(1024) k_rank(uint32_t* out, uint32_t bits, uint32_t tiles) {
  extern __shared__ uint32_t lds[];
  const uint32_t nb = 1u << bits, tid = threadIdx.x, T = blockDim.x;
  const uint32_t wave = tid >> 6;
  uint32_t acc = 0;
  for (uint32_t t = 0; t < tiles; t++) {
    const uint32_t ncnt = MODE == 3 ? nb * (T >> 6) : nb;
    for (uint32_t i = tid; i < ncnt; i += T) lds[i] = 0;
    __syncthreads();
    uint32_t key[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) key[i] = mix((blockIdx.x * 977u + t) * 65536u + i * T + tid) & (nb - 1);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      if (MODE == 0) atomicAdd(&lds[key[i]], 1u);
      else if (MODE == 1) acc += atomicAdd(&lds[key[i]], 1u);
      else if (MODE == 2) acc += lds_rank_add(lds, key[i], true, bits);
      else {
        uint64_t peers = ~0ull;
        for (uint32_t b = 0; b < bits; b++) {
          const bool bit = (key[i] >> b) & 1u;
          const uint64_t m = __ballot(bit);
          peers &= bit ? m : ~m;
        }
        const uint32_t lane = tid & 63u;
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        uint32_t* mine = lds + wave * nb;
        uint32_t base = 0;
        if (rank == 0) { base = mine[key[i]]; mine[key[i]] = base + (uint32_t)__popcll(peers); }
        base = __shfl(base, __ffsll((long long)peers) - 1, 64);
        acc += base + rank;
      }
    }
    __syncthreads();
    acc += lds[tid & (nb - 1)];
  }
  out[blockIdx.x * T + tid] = acc;
}

template <int MODE, int ITEMS>
void run(const char* name, uint32_t bits, int threads, int blocks_per_cu, int n_cu, uint32_t* d_out) {
  const uint32_t tiles = 400;
  const size_t lds = (size_t)(MODE == 3 ? (threads / 64) : 1) * (1u << bits) * 4;
  if (lds > 160 * 1024 / blocks_per_cu) { printf("%-9s bins 2^%u threads %4d x %d/CU: needs %zu KB of LDS, skipped\n", name, bits, threads, blocks_per_cu, lds >> 10); return; }
  CHECK(hipFuncSetAttribute((const void*)k_rank<MODE, ITEMS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int grid = n_cu * blocks_per_cu;
  hipLaunchKernelGGL((k_rank<MODE, ITEMS>), dim3(grid), dim3(threads), lds, 0, d_out, bits, 20u);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_rank<MODE, ITEMS>), dim3(grid), dim3(threads), lds, 0, d_out, bits, tiles);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double keys = (double)grid * threads * ITEMS * tiles;
  printf("%-9s bins 2^%-2u threads %4d x %d/CU items %2d: %7.3f ms  %7.1f keys/ns chip  %6.2f keys/clk/CU (2.4 GHz)  %6.0f ns per 16k keys per CU\n", name, bits,
         threads, blocks_per_cu, ITEMS, ms, keys / (ms * 1e6), keys / (ms * 1e6) / n_cu / 2.4, 16384.0 / (keys / (ms * 1e6) / n_cu));
}

int main(int argc, char** argv) {
  const uint32_t bits = argc > 1 ? atoi(argv[1]) : 10;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  uint32_t* d_out;
  CHECK(hipMalloc(&d_out, (size_t)n_cu * 4 * 1024 * 4));
  for (uint32_t b : {bits, 8u, 3u}) {
    run<0, 16>("add", b, 1024, 1, n_cu, d_out);
    run<1, 16>("rtn", b, 1024, 1, n_cu, d_out);
    run<1, 8>("rtn", b, 512, 2, n_cu, d_out);
    run<1, 8>("rtn", b, 1024, 2, n_cu, d_out);
    run<2, 16>("ballot", b, 1024, 1, n_cu, d_out);
    run<3, 16>("wavepriv", b, 1024, 1, n_cu, d_out);
    run<3, 8>("wavepriv", b, 512, 2, n_cu, d_out);
  }
  return 0;
}
