// Cost of a dependent chain of projective additions as a function of the EXEC mask (which lanes take part).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../montgomery_amd/csrc/curve.h"
using namespace msm;
constexpr int ADDS = 8;
__global__ void __launch_bounds__(64) k_mask(uint32_t* out, uint64_t mask) {
  Fe<Fp377> x, y, z;
  for (int i = 0; i < 13; i++) { x.l[i] = (threadIdx.x * 7 + i * 13 + 5) & LMASK; y.l[i] = (threadIdx.x * 3 + i * 11 + 1) & LMASK; }
  z = y;
  Proj<Fp377> P, Q; P.X = x; P.Y = y; P.Z = z; Q.X = y; Q.Y = x; Q.Z = y;
  if ((mask >> threadIdx.x) & 1) {
#pragma unroll 1
    for (int it = 0; it < ADDS; it++) proj_add<Fp377>(P, P, Q);
  }
  uint32_t s = 0;
  for (int i = 0; i < 13; i++) s ^= P.X.l[i] ^ P.Y.l[i] ^ P.Z.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  uint32_t* out; hipMalloc(&out, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct { const char* name; uint64_t m; } masks[] = {
      {"all 64 lanes", ~0ull}, {"lanes 0-62", ~0ull >> 1}, {"lanes 0-59", ~0ull >> 4}, {"lanes 0-47", ~0ull >> 16}, {"lanes 0-31", ~0ull >> 32},
      {"lanes 0-15", 0xFFFFull}, {"lanes 0-7", 0xFFull}, {"lanes 0-3", 0xFull}, {"lanes 0-1", 3ull}, {"lane 0", 1ull}, {"lane 63", 1ull << 63},
      {"lanes 0,16,32,48", 0x0001000100010001ull}, {"even lanes", 0x5555555555555555ull}, {"lanes 32-63", ~0ull << 32}, {"all but lane 5", ~(1ull << 5)}};
  for (int waves : {256, 832, 2048})
    for (auto& mk : masks) {
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0); k_mask<<<waves, 64>>>(out, mk.m); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("waves=%5d  %-16s %8.1f us  %6.2f us per addition in sequence\n", waves, mk.name, ms * 1e3, ms * 1e3 / ADDS);
    }
  return 0;
}
