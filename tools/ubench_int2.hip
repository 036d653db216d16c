// Instruction-rate microbenchmark, second edition (round 2): validates the integer-MAD roofline denominator.
//   * controls: v_fma_f32, v_pk_fma_f32, v_add_u32, v_and_b32 next to v_mad_u64_u32 and the other ops the big-integer
//     kernels issue;
//   * kernels of >= 5 ms (the round-1 edition timed 0.13 ms launches);
//   * a sweep over 1 / 2 / 4 / 8 resident waves per SIMD (grid = CUs x k workgroups of 256 threads, total work fixed);
//   * the clock actually held during each kernel: delta s_memtime / delta s_memrealtime x 100 MHz, stamped by one lane
//     per workgroup around the loop (MI355X_MICROARCH.md, DVFS item 6), median over workgroups.
// Cycles per wave-instruction per SIMD are quoted in REAL shader cycles (from the stamps); lane-ops/s from wall time.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_int2.hip -o tools/ubench_int2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ILP = 8;

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint64_t* stamps, uint32_t a0, uint32_t b0, int iters) {
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
  uint64_t acc[ILP], acc2[ILP];
  uint32_t lo[ILP];
  float f[ILP];
  double d[ILP];
  for (int i = 0; i < ILP; i++) { acc[i] = i * 77 + threadIdx.x; lo[i] = i + threadIdx.x; d[i] = 1.0 + i + threadIdx.x; f[i] = 1.0f + i; acc2[i] = i; }
  double da = 1.0000001, db = 0.5;
  float fa = 1.0000001f, fb = 0.5f;
  uint64_t sm = __ballot(threadIdx.x & 1);
  uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(a));
      if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fa), "v"(fb));
      if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(a));
      if (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(da), "v"(db));
      if (OP == 5) asm volatile("v_and_b32 %0, %0, %1" : "+v"(lo[i]) : "v"(a));
      if (OP == 6) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo[i]) : "v"(a), "v"(b));
      if (OP == 7) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(lo[i]) : "v"(a) : "vcc");
      if (OP == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % ILP]));
      if (OP == 9) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % ILP]));
      if (OP == 10) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(lo[i]) : "v"(a));
      if (OP == 11) asm volatile("v_lshrrev_b64 %0, 30, %0" : "+v"(acc[i]));
      if (OP == 12) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      if (OP == 13) asm volatile("v_mov_b32 %0, %1" : "+v"(lo[i]) : "v"(a));
      if (OP == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo[i]) : "v"(a));
      if (OP == 15) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(lo[i]) : "v"(a), "v"(b));
      if (OP == 16) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(lo[i]) : "v"(a), "v"(b));
      if (OP == 17) asm volatile("v_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(lo[i]) : "v"(a) : "vcc");
      if (OP == 18) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(lo[i]) : "v"(a));
      if (OP == 19) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(lo[i]));
      if (OP == 20) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(lo[i]) : "v"(a));
      if (OP == 21) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(lo[i]) : "v"(a) : "vcc");
      if (OP == 22) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, %1, %2" : "+v"(acc[i]), "+v"(lo[i]) : "v"(a), "v"(b) : "vcc");
      if (OP == 23) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshl_add_u64 %1, %1, 0, %4" : "+v"(acc[i]), "+v"(acc2[i]) : "v"(a), "v"(b), "v"(acc2[(i + 1) % ILP]) : "vcc");
      if (OP == 24) asm volatile("v_mov_b64 %0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % ILP]));
      if (OP == 25) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(lo[i]) : "v"(a), "s"(sm));
      if (OP == 28) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(lo[i]) : "v"(a));
      if (OP == 29) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(lo[i]) : "v"(a) : "vcc");
      if (OP == 30) asm volatile("v_cmp_lt_u32 %2, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(lo[i]) : "v"(a), "s"(sm));
      if (OP == 31) { lo[i] = (lo[i] < a) ? lo[i] + b : lo[i] ^ a; }
      if (OP == 26) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(a));
      // dependent accumulation chains, as a column of a product-scanning multiplier issues them (1, 2, 4 independent chains)
      if (OP == 40) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
      if (OP == 41) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i & 1]) : "v"(a), "v"(b) : "vcc");
      if (OP == 42) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b) : "vcc");
      if (OP == 43) asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo[0]) : "v"(a));
      if (OP == 27) asm volatile("v_bfe_u32 %0, %0, 3, 30" : "+v"(lo[i]));
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
  uint64_t s = 0; double ds = 0; float fs = 0;
  for (int i = 0; i < ILP; i++) { s += acc[i] + lo[i] + acc2[i]; ds += d[i]; fs += f[i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ (uint32_t)ds ^ (uint32_t)fs;
}

static int n_cu = 256;

template <int OP>
int run(const char* name, int waves_per_simd, uint32_t* out, uint64_t* d_stamps) {
  const int blocks = n_cu * waves_per_simd;
  const int iters = 3200000 / ILP / waves_per_simd;   // fixed work per SIMD: >= 5 ms at 4 cycles per instruction
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<OP><<<blocks, 256>>>(out, d_stamps, 12345, 67890, iters);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  double clk = 0;
  for (int r = 0; r < 3; r++) {
    CHECK(hipEventRecord(e0));
    k<OP><<<blocks, 256>>>(out, d_stamps, 12345 + r, 67890, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) {
      best = ms;
      std::vector<uint64_t> st(2 * blocks);
      CHECK(hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> c(blocks);
      for (int b = 0; b < blocks; b++) c[b] = st[2 * b + 1] ? (double)st[2 * b] / (double)st[2 * b + 1] * 100e6 : 0;
      std::sort(c.begin(), c.end());
      clk = c[blocks / 2];
    }
  }
  const double ops = (double)blocks * 256 * iters * ILP;
  const double waveinstr_per_simd = ops / 64 / (n_cu * 4);
  const double cyc_real = best * 1e-3 * clk / waveinstr_per_simd;
  printf("%-16s waves/SIMD=%d  %8.3f ms  clock %5.0f MHz  %8.2f Tlane-op/s  %5.2f real cyc/wave-instr/SIMD\n", name, waves_per_simd,
         best, clk * 1e-6, ops / best * 1e-9, cyc_real);
  return 0;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  n_cu = prop.multiProcessorCount;
  printf("device %s CUs=%d nominal clock=%d kHz\n", prop.gcnArchName, n_cu, prop.clockRate);
  uint32_t* out; uint64_t* stamps;
  CHECK(hipMalloc(&out, (size_t)n_cu * 8 * 256 * 4));
  CHECK(hipMalloc(&stamps, (size_t)n_cu * 8 * 16));
  for (int w : {2, 4}) {
    run<2>("v_fma_f32", w, out, stamps);
    run<8>("v_pk_fma_f32", w, out, stamps);
    run<3>("v_add_u32", w, out, stamps);
    run<5>("v_and_b32", w, out, stamps);
    run<13>("v_mov_b32", w, out, stamps);
    run<0>("v_mad_u64_u32", w, out, stamps);
    run<40>("mad64 1 chain", w, out, stamps);
    run<41>("mad64 2 chains", w, out, stamps);
    run<42>("mad64 4 chains", w, out, stamps);
    run<43>("add_u32 1 chain", w, out, stamps);
    run<12>("v_mad_i64_i32", w, out, stamps);
    run<1>("v_mul_lo_u32", w, out, stamps);
    run<6>("v_add3_u32", w, out, stamps);
    run<7>("v_addc_co_u32", w, out, stamps);
    run<9>("v_lshl_add_u64", w, out, stamps);
    run<11>("v_lshrrev_b64", w, out, stamps);
    run<10>("v_alignbit_b32", w, out, stamps);
    run<4>("v_fma_f64", w, out, stamps);
    run<14>("v_cndmask(vcc)", w, out, stamps);
    run<25>("v_cndmask(sgpr)", w, out, stamps);
    run<28>("v_cndmask_e64(vcc)", w, out, stamps);
    run<29>("cmp+cndmask_e32 (x2)", w, out, stamps);
    run<30>("cmp+cndmask_e64 (x2)", w, out, stamps);
    run<31>("C select (cmp,add,xor,cnd)", w, out, stamps);
    run<15>("v_bfi_b32", w, out, stamps);
    run<16>("v_or3_b32", w, out, stamps);
    run<17>("v_subb_co_u32", w, out, stamps);
    run<21>("v_add_co_u32", w, out, stamps);
    run<18>("v_lshl_or_b32", w, out, stamps);
    run<19>("v_lshrrev_b32", w, out, stamps);
    run<20>("v_xor_b32", w, out, stamps);
    run<26>("v_sub_u32", w, out, stamps);
    run<27>("v_bfe_u32", w, out, stamps);
    run<24>("v_mov_b64", w, out, stamps);
    run<22>("mad64+and (x2)", w, out, stamps);
    run<23>("mad64+add64 (x2)", w, out, stamps);
  }
  return 0;
}
