// Throughput of chains of mixed projective additions (the k_bucket_finish inner loop) without any memory traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../montgomery_amd/csrc/curve.h"
using namespace msm;
constexpr int ADDS = 6;
template <int MIXED>
__global__ void __launch_bounds__(256) k_chain(uint32_t* out) {
  Fe<Fp377> x, y, z;
  for (int i = 0; i < 13; i++) { x.l[i] = (threadIdx.x * 7 + i * 13 + 5) & LMASK; y.l[i] = (threadIdx.x * 3 + i * 11 + 1) & LMASK; }
  z = y;
  Proj<Fp377> P, Q; P.X = x; P.Y = y; P.Z = z; Q.X = y; Q.Y = x; Q.Z = y;
#pragma unroll 1
  for (int it = 0; it < ADDS; it++) {
    if (MIXED) proj_add_mixed<Fp377>(P, P, Q, false); else proj_add<Fp377>(P, P, Q);
    Q.X.l[0] ^= it;
  }
  uint32_t s = 0;
  for (int i = 0; i < 13; i++) s ^= P.X.l[i] ^ P.Y.l[i] ^ P.Z.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  uint32_t* out; hipMalloc(&out, 1 << 26);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mixed = 0; mixed < 2; mixed++)
    for (int waves : {1024, 2048, 4096, 8192}) {
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (mixed) k_chain<1><<<waves / 4, 256>>>(out); else k_chain<0><<<waves / 4, 256>>>(out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("%-8s waves=%5d  %8.1f us total  %.3e additions/s\n", mixed ? "mixed" : "general", waves, ms * 1e3, (double)waves * 64 * ADDS / (ms * 1e-3));
    }
  return 0;
}
