// 12-word add-with-carry chains as hipcc writes them for the packed field arithmetic: carry through an SGPR pair
// (VOP3b, what the compiler picks when VCC is busy) against carry through VCC (VOP2).  2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(256, 2) k(uint32_t* out, uint32_t a0, int iters) {
  uint32_t x[12], y[12];
  for (int i = 0; i < 12; i++) { x[i] = a0 * (i + 1) + threadIdx.x; y[i] = a0 ^ (i * 977 + threadIdx.x); }
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {   // carry in VCC
      asm volatile(
          "v_add_co_u32 %0, vcc, %0, %12\n\tv_addc_co_u32 %1, vcc, %1, %13, vcc\n\tv_addc_co_u32 %2, vcc, %2, %14, vcc\n\t"
          "v_addc_co_u32 %3, vcc, %3, %15, vcc\n\tv_addc_co_u32 %4, vcc, %4, %16, vcc\n\tv_addc_co_u32 %5, vcc, %5, %17, vcc\n\t"
          "v_addc_co_u32 %6, vcc, %6, %18, vcc\n\tv_addc_co_u32 %7, vcc, %7, %19, vcc\n\tv_addc_co_u32 %8, vcc, %8, %20, vcc\n\t"
          "v_addc_co_u32 %9, vcc, %9, %21, vcc\n\tv_addc_co_u32 %10, vcc, %10, %22, vcc\n\tv_addc_co_u32 %11, vcc, %11, %23, vcc"
          : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11])
          : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8]), "v"(y[9]), "v"(y[10]), "v"(y[11])
          : "vcc");
    } else {           // carry in an SGPR pair
      uint64_t c;
      asm volatile(
          "v_add_co_u32_e64 %0, %12, %0, %13\n\tv_addc_co_u32_e64 %1, %12, %1, %14, %12\n\tv_addc_co_u32_e64 %2, %12, %2, %15, %12\n\t"
          "v_addc_co_u32_e64 %3, %12, %3, %16, %12\n\tv_addc_co_u32_e64 %4, %12, %4, %17, %12\n\tv_addc_co_u32_e64 %5, %12, %5, %18, %12\n\t"
          "v_addc_co_u32_e64 %6, %12, %6, %19, %12\n\tv_addc_co_u32_e64 %7, %12, %7, %20, %12\n\tv_addc_co_u32_e64 %8, %12, %8, %21, %12\n\t"
          "v_addc_co_u32_e64 %9, %12, %9, %22, %12\n\tv_addc_co_u32_e64 %10, %12, %10, %23, %12\n\tv_addc_co_u32_e64 %11, %12, %11, %24, %12"
          : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]),
            "=&s"(c)
          : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8]), "v"(y[9]), "v"(y[10]), "v"(y[11]));
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < 12; i++) s ^= x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, uint32_t* out, int n_cu) {
  const int iters = 40000, blocks = n_cu * 2;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 12345 + r, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-24s %8.3f ms  %6.2f SIMD cycles (2.4 GHz) per carry instruction\n", name, best, best * 1e-3 * 2.4e9 / ((double)iters * 12 * 2));
}
int main() {
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  uint32_t* out; (void)hipMalloc(&out, 1 << 24);
  run<0>("carry in VCC", out, prop.multiProcessorCount);
  run<1>("carry in an SGPR pair", out, prop.multiProcessorCount);
  return 0;
}
