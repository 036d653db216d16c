// The four field inversions on the GPU: division steps (fe_inv, the product's), Fermat
// (a^(p-2)), Kaliski's almost-inverse (the reference's algorithm, src/wasm/inverse.ts:136-218) and the reference's
// experimental word-sliced almost-inverse (src/inverse/faster-inverse-wasm.ts:133-343), each as a chain of
// dependent inversions: latency for a lone wave and throughput with every SIMD holding two waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../montgomery_amd/csrc/field.h"
using namespace msm;
constexpr int REPS = 4;
template <class F, int WHICH>
__global__ void __launch_bounds__(256) k_inv(uint32_t* out) {
  Fe<F> x, r;
  for (int i = 0; i < F::NL; i++) x.l[i] = (threadIdx.x * 2654435761u + blockIdx.x * 40503u + i * 97u + 5u) & LMASK;
  x.l[F::NL - 1] &= 0xFFFF;
#pragma unroll 1
  for (int it = 0; it < REPS; it++) {
    if (WHICH == 0) fe_inv<F>(r, x);
    if (WHICH == 1) fe_inv_fermat<F>(r, x);
    if (WHICH == 2) fe_inv_kaliski<F>(r, x);
    if (WHICH == 3) fe_inv_wordsliced<F>(r, x);
    for (int i = 0; i < F::NL; i++) x.l[i] = r.l[i] ^ (uint32_t)it;
    fe_reduce_4p<F>(x);
  }
  uint32_t s = 0;
  for (int i = 0; i < F::NL; i++) s ^= x.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F>
void run(const char* field, uint32_t* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"division steps", "Fermat", "Kaliski", "word-sliced"};
  for (int which = 0; which < 4; which++)
    for (int cfg = 0; cfg < 2; cfg++) {
      const int blocks = cfg ? 2048 : 1, threads = cfg ? 256 : 64;
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (which == 0) k_inv<F, 0><<<blocks, threads>>>(out);
        if (which == 1) k_inv<F, 1><<<blocks, threads>>>(out);
        if (which == 2) k_inv<F, 2><<<blocks, threads>>>(out);
        if (which == 3) k_inv<F, 3><<<blocks, threads>>>(out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("%-6s %-15s %-22s %9.1f us per inversion in sequence   %.3e inversions/s\n", field, names[which],
             cfg ? "2048 x 256 (2 waves/SIMD)" : "one wave", ms * 1e3 / REPS, (double)blocks * threads * REPS / (ms * 1e-3));
    }
}
int main() {
  uint32_t* out; hipMalloc(&out, 1 << 24);
  run<Fp377>("Fp377", out);
  run<Fp253>("Fp253", out);
  return 0;
}
