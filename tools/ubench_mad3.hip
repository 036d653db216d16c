// v_mad_u64_u32 as hipcc itself emits it (no inline asm: two adjacent asm statements get an s_nop between them, which
// the figures of ubench_int2 therefore include).  8 multiply-accumulates per step into 8 / 4 / 2 / 1 accumulators, the
// multiplier renewed by one v_mul_lo + v_add per step.  2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int CHAINS>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t a0, int iters) {
  uint64_t acc[8];
  uint32_t x[8];
  for (int i = 0; i < 8; i++) { acc[i] = threadIdx.x + i; x[i] = a0 * (i + 3) + threadIdx.x; }
  uint32_t y = a0 ^ threadIdx.x;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i % CHAINS] += (uint64_t)x[i] * y;
      y = y * 1664525u + 1013904223u;
    }
  }
  uint64_t s = 0;
  for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
template <int CHAINS>
void run(int waves, uint32_t* out) {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * waves, iters = 400000 / waves;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(e0);
    k<CHAINS><<<blocks, 256>>>(out, 12345 + r, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double wave_instr_per_simd = (double)iters * 4 * 8 * waves;   // MADs only
  printf("compiler MADs, %d accumulators, %d waves/SIMD: %7.3f ms  %5.2f cycles at 2.4 GHz per MAD per SIMD (incl. 2 other VALU per 8)\n", CHAINS,
         waves, best, best * 1e-3 * 2.4e9 / wave_instr_per_simd);
}
int main() {
  uint32_t* out; hipMalloc(&out, 1 << 24);
  for (int w : {2, 4}) { run<8>(w, out); run<4>(w, out); run<2>(w, out); run<1>(w, out); }
  return 0;
}
