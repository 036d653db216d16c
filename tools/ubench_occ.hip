// How does the dispatcher place many small workgroups, and what does a dependent chain of projective additions cost
// per step under each placement?  (Decides the launch shape of the bucket-reduction kernels.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../montgomery_amd/csrc/curve.h"
using namespace msm;
constexpr int ADDS = 16;
template <int MODE>
__device__ __forceinline__ void body(uint32_t* out) {
  Fe<Fp377> x, y, z;
  for (int i = 0; i < 13; i++) { x.l[i] = (threadIdx.x * 7 + i * 13 + 5) & LMASK; y.l[i] = (threadIdx.x * 3 + i * 11 + 1) & LMASK; }
  z = y;
  Proj<Fp377> P, Q; P.X = x; P.Y = y; P.Z = z; Q.X = y; Q.Y = x; Q.Z = y;
#pragma unroll 1
  for (int it = 0; it < ADDS; it++) proj_add<Fp377>(P, P, Q);
  uint32_t s = 0;
  for (int i = 0; i < 13; i++) s ^= P.X.l[i] ^ P.Y.l[i] ^ P.Z.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(64) k_w64(uint32_t* out) { body<0>(out); }
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) k_w64_occ1(uint32_t* out) { body<1>(out); }
__global__ void __launch_bounds__(256) k_w256(uint32_t* out) { body<2>(out); }
__global__ void __launch_bounds__(64) k_w64_lds(uint32_t* out) {
  extern __shared__ uint32_t pad[];
  if (threadIdx.x == 999) pad[0] = 1;
  body<3>(out);
}
int main() {
  uint32_t* out; hipMalloc(&out, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k_w64_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int variant = 0; variant < 6; variant++)
    for (int waves : {64, 256, 512, 1024, 2048, 4096}) {
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        switch (variant) {
          case 0: k_w64<<<waves, 64>>>(out); break;
          case 1: k_w64_occ1<<<waves, 64>>>(out); break;
          case 2: k_w256<<<waves / 4, 256>>>(out); break;
          case 3: k_w64_lds<<<waves, 64, 40 * 1024>>>(out); break;      // 4 blocks per CU
          case 4: k_w64_lds<<<waves, 64, 80 * 1024>>>(out); break;      // 2 blocks per CU
          case 5: k_w64_lds<<<waves, 64, 159 * 1024>>>(out); break;     // 1 block per CU
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const char* names[] = {"64-thread blocks", "64-thread, waves_per_eu(1,1)", "256-thread blocks", "64-thread + 40 KB LDS", "64-thread + 80 KB LDS",
                             "64-thread + 159 KB LDS"};
      printf("%-30s waves=%5d  %8.1f us total  %6.2f us per addition in sequence\n", names[variant], waves, ms * 1e3, ms * 1e3 / ADDS);
    }
  return 0;
}
