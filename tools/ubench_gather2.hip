// Does the way the row table is ALLOCATED move the cliff of tools/ubench_gather.hip (scattered 128-byte line reads are
// 2.3x slower once the lanes of one wave instruction spread over more than ~1 GB)?  Same kernel, pattern 0 (every lane
// anywhere) and pattern 1 (lanes of a wave inside one window), over a table obtained by
//   mode 0: hipMalloc
//   mode 1: hipExtMallocWithFlags(hipDeviceMallocContiguous)
//   mode 2: hipMemCreate (one physical handle) mapped at a 1 GB / 2 GB / 32 GB aligned reservation
// (pattern 3 = pattern 1 with the window aligned to its own size)
// usage: ubench_gather2 [GB] [mode] [align_log2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <int PATTERN>
__global__ void __launch_bounds__(256, 2) k_gather(const char* table, uint64_t lines, uint64_t win_lines, int steps, uint32_t* out) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t wave = t >> 6;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 1
  for (int i = 0; i < steps; i++) {
    uint64_t la, lb;
    if (PATTERN == 0) {
      la = mix(((uint64_t)t << 20) + 2 * i) % lines;
      lb = mix(((uint64_t)t << 20) + 2 * i + 1) % lines;
    } else {
      uint64_t base = (mix(((uint64_t)wave << 20) + i) % (lines - win_lines + 1));
      if (PATTERN == 3) base = base / win_lines * win_lines;   // window aligned to its own size
      la = base + mix(((uint64_t)t << 20) + 2 * i) % win_lines;
      lb = base + mix(((uint64_t)t << 20) + 2 * i + 1) % win_lines;
    }
    const uint4* pa = reinterpret_cast<const uint4*>(table + la * 128);
    const uint4* pb = reinterpret_cast<const uint4*>(table + lb * 128);
    uint4 v[12];
#pragma unroll
    for (int j = 0; j < 6; j++) { v[j] = pa[j]; v[6 + j] = pb[j]; }
#pragma unroll
    for (int j = 0; j < 12; j++) { acc.x ^= v[j].x; acc.y += v[j].y; acc.z ^= v[j].z; acc.w += v[j].w; }
  }
  out[t] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const uint64_t table_gb = argc > 1 ? atoll(argv[1]) : 16;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;
  const int align_log2 = argc > 3 ? atoi(argv[3]) : 30;
  const uint64_t bytes = table_gb << 30, lines = bytes / 128;
  char* table = nullptr;
  if (mode == 0) {
    CHK(hipMalloc(&table, bytes));
  } else if (mode == 1) {
    CHK(hipExtMallocWithFlags((void**)&table, bytes, hipDeviceMallocContiguous));
  } else {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CHK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("recommended granularity %zu\n", gran);
    hipMemGenericAllocationHandle_t h;
    CHK(hipMemCreate(&h, bytes, &prop, 0));
    CHK(hipMemAddressReserve((void**)&table, bytes, (size_t)1 << align_log2, nullptr, 0));
    CHK(hipMemMap(table, bytes, 0, h, 0));
    hipMemAccessDesc ad = {};
    ad.location = prop.location;
    ad.flags = hipMemAccessFlagsProtReadWrite;
    CHK(hipMemSetAccess(table, bytes, &ad, 1));
  }
  printf("mode %d  table %llu GB at %p\n", mode, (unsigned long long)table_gb, (void*)table);
  CHK(hipMemset(table, 1, bytes));
  uint32_t* out; CHK(hipMalloc(&out, 1 << 22));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 512, steps = 400;
  const double useful = (double)blocks * 256 * steps * 192;
  auto run = [&](int pattern, uint64_t range_bytes, uint64_t win_bytes) {
    const uint64_t rl = range_bytes / 128, wl = win_bytes / 128;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (pattern == 0) k_gather<0><<<blocks, 256>>>(table, rl, wl, steps, out);
      if (pattern == 1) k_gather<1><<<blocks, 256>>>(table, rl, wl, steps, out);
      if (pattern == 3) k_gather<3><<<blocks, 256>>>(table, rl, wl, steps, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("mode %d pattern %d  range %8.0f MB  wave window %8.0f MB : %7.3f ms  %6.0f GB/s useful  %5.2f G lines/s\n", mode, pattern,
           range_bytes / 1048576.0, win_bytes / 1048576.0, ms, useful / ms / 1e6, (double)blocks * 256 * steps * 2 / ms / 1e6);
  };
  for (uint64_t mb : {256ull, 1024ull, 2048ull, 4096ull, (unsigned long long)(table_gb * 1024)}) run(0, mb << 20, 0);
  for (uint64_t wmb : {512ull, 1024ull, 2048ull, 4096ull}) run(1, bytes, wmb << 20);
  for (uint64_t wmb : {1024ull, 2048ull, 4096ull}) run(3, bytes, wmb << 20);
  return 0;
}
