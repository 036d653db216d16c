// The field multiplier as compiled, at 1 / 2 / 4 resident waves per SIMD (256-thread blocks): SIMD cycles per product,
// and the same with 8 / 16 / 32 products per loop iteration (up to 113 KB of code: does instruction fetch keep up?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../montgomery_amd/csrc/curve.h"
using namespace msm;
template <int MODE, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k(uint32_t* out, int iters) {
  Fe<Fp377> x, y, z;
  for (int i = 0; i < 13; i++) { x.l[i] = (threadIdx.x * 7 + i * 13 + 5) & LMASK; y.l[i] = (threadIdx.x * 3 + i * 11 + 1) & LMASK; }
  z = y;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) { fe_mul<Fp377>(x, x, y); __builtin_amdgcn_sched_barrier(0); fe_mul<Fp377>(z, z, x); __builtin_amdgcn_sched_barrier(0); }
    if (MODE == 1) { fe_sqr<Fp377>(x, x); __builtin_amdgcn_sched_barrier(0); fe_sqr<Fp377>(z, z); __builtin_amdgcn_sched_barrier(0); }
    if (MODE >= 2) {   // a long loop body (MODE products, ~3.5 KB of code each): does the instruction fetch keep up?
#pragma unroll
      for (int u = 0; u < MODE / 2; u++) { fe_mul<Fp377>(x, x, y); __builtin_amdgcn_sched_barrier(0); fe_mul<Fp377>(z, z, x); __builtin_amdgcn_sched_barrier(0); }
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < 13; i++) s ^= x.l[i] ^ z.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int WAVES>
void run(const char* name, uint32_t* out, int n_cu) {
  const int per_iter = MODE >= 2 ? MODE : 2;
  const int iters = 8000 / per_iter / WAVES, blocks = n_cu * WAVES;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    (void)hipEventRecord(e0);
    k<MODE, WAVES><<<blocks, 256>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double per_simd = (double)iters * per_iter * WAVES;   // products per SIMD
  printf("%-8s %d waves/SIMD: %8.3f ms  %7.0f SIMD cycles (2.4 GHz) per product\n", name, WAVES, best, best * 1e-3 * 2.4e9 / per_simd);
}
int main() {
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  uint32_t* out; (void)hipMalloc(&out, 1 << 24);
  run<0, 1>("fe_mul", out, prop.multiProcessorCount); run<0, 2>("fe_mul", out, prop.multiProcessorCount); run<0, 4>("fe_mul", out, prop.multiProcessorCount);
  run<8, 2>("8 mul/loop", out, prop.multiProcessorCount); run<16, 2>("16 mul/loop", out, prop.multiProcessorCount); run<32, 2>("32 mul/loop", out, prop.multiProcessorCount);
  run<1, 1>("fe_sqr", out, prop.multiProcessorCount); run<1, 2>("fe_sqr", out, prop.multiProcessorCount); run<1, 4>("fe_sqr", out, prop.multiProcessorCount);
  return 0;
}
