// Latency / throughput of the field multiplier as compiled: dependent chain on one wave vs full occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../montgomery_amd/csrc/curve.h"
using namespace msm;
constexpr int ITERS = 2000;
__global__ void k_chain(uint32_t* out, int mode) {
  Fe<Fp377> x, y, z;
  for (int i = 0; i < 13; i++) { x.l[i] = (threadIdx.x * 7 + i * 13 + 5) & LMASK; y.l[i] = (threadIdx.x * 3 + i * 11 + 1) & LMASK; }
  z = y;
  if (mode == 0) for (int it = 0; it < ITERS; it++) fe_mul<Fp377>(x, x, y);                       // dependent chain
  if (mode == 1) for (int it = 0; it < ITERS / 2; it++) { fe_mul<Fp377>(x, x, y); fe_mul<Fp377>(z, z, y); }  // two independent chains
  if (mode == 2) for (int it = 0; it < ITERS; it++) fe_sqr<Fp377>(x, x);
  if (mode == 3) {   // projective additions, dependent
    Proj<Fp377> P, Q; P.X = x; P.Y = y; P.Z = z; Q.X = y; Q.Y = x; Q.Z = y;
    for (int it = 0; it < ITERS / 20; it++) proj_add<Fp377>(P, P, Q);
    x = P.X;
  }
  uint32_t s = 0;
  for (int i = 0; i < 13; i++) s ^= x.l[i] ^ z.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  uint32_t* out; hipMalloc(&out, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"mul dependent", "mul 2 chains", "sqr dependent", "proj_add dependent"};
  for (int mode = 0; mode < 4; mode++)
    for (int cfg = 0; cfg < 3; cfg++) {
      int blocks = cfg == 0 ? 1 : (cfg == 1 ? 1024 : 2048), threads = cfg == 0 ? 64 : 256;
      k_chain<<<blocks, threads>>>(out, mode); hipDeviceSynchronize();
      hipEventRecord(e0); k_chain<<<blocks, threads>>>(out, mode); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double ops = mode == 3 ? ITERS / 20 : ITERS;
      printf("%-20s blocks=%4d threads=%3d  %8.3f ms  %7.3f us/op (per lane chain)  %.3e lane-ops/s\n", names[mode], blocks, threads, ms,
             ms * 1e3 / ops, ops * blocks * threads / (ms * 1e-3));
    }
  return 0;
}
