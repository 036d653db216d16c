// Random 128-byte-line gathers from a big table, as round 1 of the accumulation tree issues them: what does the rate
// depend on -- the size of the range the WHOLE chip touches (cache / TLB reach), the range ONE WAVE touches per
// instruction (translations per instruction), or neither?  (Decides how the sort should order the payloads of a bucket
// and how large a window size c the gather round tolerates.)
//   pattern 0: every lane anywhere in the table
//   pattern 1: the 64 lanes of a wave inside one window of `win` bytes (window position random per wave and step)
//   pattern 2: lane-sorted: lane l of a wave reads from the l-th 1/64 slice of the wave's window (ascending rows)
// Each step loads what the backward sweep loads of two operands (x and y: 6 x 16 bytes from two lines), 2 waves per SIMD.
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

// PIECES: 16-byte pieces read of every line -- 6 = x and y (both 64-byte sectors: the backward sweep), 3 = x alone (the first
// sector: the forward sweep)
template <int PATTERN, int PIECES = 6>
__global__ void __launch_bounds__(256, 2) k_gather(const char* table, uint64_t lines, uint64_t win_lines, int steps, uint32_t* out) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t wave = t >> 6, lane = t & 63;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 1
  for (int i = 0; i < steps; i++) {
    uint64_t la, lb;
    if (PATTERN == 0) {
      la = mix(((uint64_t)t << 20) + 2 * i) % lines;
      lb = mix(((uint64_t)t << 20) + 2 * i + 1) % lines;
    } else {
      const uint64_t base = (mix(((uint64_t)wave << 20) + i) % (lines - win_lines + 1));
      if (PATTERN == 1) {
        la = base + mix(((uint64_t)t << 20) + 2 * i) % win_lines;
        lb = base + mix(((uint64_t)t << 20) + 2 * i + 1) % win_lines;
      } else {
        const uint64_t slice = win_lines / 128;
        la = base + (2 * lane) * slice + mix(((uint64_t)t << 20) + 2 * i) % slice;
        lb = base + (2 * lane + 1) * slice + mix(((uint64_t)t << 20) + 2 * i + 1) % slice;
      }
    }
    const uint4* pa = reinterpret_cast<const uint4*>(table + la * 128);
    const uint4* pb = reinterpret_cast<const uint4*>(table + lb * 128);
    uint4 v[2 * PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; j++) { v[j] = pa[j]; v[PIECES + j] = pb[j]; }
#pragma unroll
    for (int j = 0; j < 2 * PIECES; j++) { acc.x ^= v[j].x; acc.y += v[j].y; acc.z ^= v[j].z; acc.w += v[j].w; }
  }
  out[t] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

int main(int argc, char** argv) {
  const uint64_t table_gb = argc > 1 ? atoll(argv[1]) : 16;
  const uint64_t bytes = table_gb << 30, lines = bytes / 128;
  char* table;
  if (hipMalloc(&table, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(table, 1, bytes);
  uint32_t* out; hipMalloc(&out, 1 << 22);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = argc > 2 ? atoi(argv[2]) : 512, steps = 400 * 512 / blocks;   // 512 = 2 blocks per CU: the tree kernel's occupancy
  const double useful = (double)blocks * 256 * steps * 192;
  auto run = [&](int pattern, uint64_t range_bytes, uint64_t win_bytes) {
    const uint64_t rl = range_bytes / 128, wl = win_bytes / 128;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (pattern == 0) k_gather<0><<<blocks, 256>>>(table, rl, wl, steps, out);
      if (pattern == 1) k_gather<1><<<blocks, 256>>>(table, rl, wl, steps, out);
      if (pattern == 2) k_gather<2><<<blocks, 256>>>(table, rl, wl, steps, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("pattern %d  range %8.0f MB  wave window %8.0f MB : %7.3f ms  %6.0f GB/s useful (192 B per lane-step)  %5.2f G lines/s\n", pattern,
           range_bytes / 1048576.0, win_bytes / 1048576.0, ms, useful / ms / 1e6, (double)blocks * 256 * steps * 2 / ms / 1e6);
  };
  if (argc > 3) {   // x alone: three pieces of the first sector of every line
    for (uint64_t wmb : {256ull, 1024ull}) {
      float ms = 0;
      const uint64_t wl = (wmb << 20) / 128;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        k_gather<1, 3><<<blocks, 256>>>(table, lines, wl, steps, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("pattern 1, first sector only (48 B per line)  wave window %8.0f MB : %7.3f ms  %5.2f G lines/s\n", wmb * 1.0, ms,
             (double)blocks * 256 * steps * 2 / ms / 1e6);
    }
    return 0;
  }
  for (uint64_t mb : {64ull, 256ull, 1024ull, 4096ull, (unsigned long long)(table_gb * 1024)}) run(0, mb << 20, 0);
  for (uint64_t wmb : {2ull, 32ull, 256ull, 1024ull, 4096ull}) run(1, bytes, wmb << 20);
  for (uint64_t wmb : {32ull, 256ull, 1024ull, 4096ull, (unsigned long long)(table_gb * 1024)}) run(2, bytes, wmb << 20);
  return 0;
}
