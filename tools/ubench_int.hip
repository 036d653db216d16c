// Instruction-rate microbenchmark for gfx950: which integer / fp64 ops can carry a big-int multiply.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_int.hip -o tools/ubench_int
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int ILP = 8;

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t a0, uint32_t b0) {
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
  uint64_t acc[ILP];
  uint32_t lo[ILP];
  double d[ILP];
  for (int i = 0; i < ILP; i++) { acc[i] = i * 77 + threadIdx.x; lo[i] = i + threadIdx.x; d[i] = 1.0 + i + threadIdx.x; }
  double da = 1.0000001, db = 0.5;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(a));
      if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(a));
      if (OP == 3) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(lo[i]) : "v"(a) : "vcc");
      if (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(da), "v"(db));
      if (OP == 5) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo[i]) : "v"(a), "v"(b));
      if (OP == 6) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo[i]) : "v"(a), "v"(b));
      if (OP == 7) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(lo[i]) : "v"(a) : "vcc");
      if (OP == 8) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo[i]) : "v"(a));
      if (OP == 9) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % ILP]));
      if (OP == 11) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da));
      if (OP == 12) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(lo[i]) : "v"(a));
      if (OP == 13) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(lo[i]) : "v"(a), "v"(b));
      if (OP == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo[i]) : "v"(a) : );
      if (OP == 15) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % ILP]));
    }
  }
  uint64_t s = 0; double ds = 0;
  for (int i = 0; i < ILP; i++) { s += acc[i] + lo[i]; ds += d[i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ (uint32_t)ds;
}

template <int OP>
int run(const char* name, int blocks_per_cu, uint32_t* out) {
  int blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<OP><<<blocks, 256>>>(out, 12345, 67890);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    CHECK(hipEventRecord(e0));
    k<OP><<<blocks, 256>>>(out, 12345 + r, 67890);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double ops = (double)blocks * 256 * ITERS * ILP;
  double waveinstr = ops / 64;
  // cycles per wave-instruction per SIMD at assumed 2.4 GHz: time * clk * (1024 SIMDs) / waveinstr
  double cyc = best * 1e-3 * 2.4e9 * 1024 / waveinstr;
  printf("%-22s wg/CU=%d  %8.3f ms  %8.2f Gop/s(lane)  ~%5.2f cyc/wave-instr/SIMD @2.4GHz\n", name, blocks_per_cu, best, ops / best * 1e-6, cyc);
  return 0;
}

int main() {
  uint32_t* out;
  CHECK(hipMalloc(&out, 256 * 8 * 256 * 4 * 4));
  for (int bpc : {4, 8}) {
    run<0>("v_mad_u64_u32", bpc, out);
    run<1>("v_mul_lo_u32", bpc, out);
    run<2>("v_mul_hi_u32", bpc, out);
    run<3>("v_add_co_u32", bpc, out);
    run<7>("v_addc_co_u32", bpc, out);
    run<6>("v_add3_u32", bpc, out);
    run<9>("v_lshl_add_u64", bpc, out);
    run<4>("v_fma_f64", bpc, out);
    run<11>("v_mul_f64", bpc, out);
    run<5>("v_mad_u32_u24", bpc, out);
    run<8>("v_mul_hi_u32_u24", bpc, out);
    run<12>("v_alignbit_b32", bpc, out);
    run<13>("v_and_or_b32", bpc, out);
    run<14>("v_cndmask_b32", bpc, out);
    run<15>("v_pk_fma_f32", bpc, out);
  }
  // clock estimate
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  return 0;
}
