// Bisects the cost of the bit-tree reduction kernel: same structure with parts switched off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../montgomery_amd/csrc/msm_kernels.h"
using namespace msm;
using F = Fp377;

template <int LOADS, int TREE, int OCC1>
__device__ __forceinline__ void bt_body(uint32_t* out, const uint32_t* rows, uint32_t n_in, uint32_t nbits, uint64_t* stamps) {
  __shared__ uint32_t lds[3 * NL * 64];
  const uint32_t blk = blockIdx.x, y = blockIdx.y, kk = blockIdx.z, tid = threadIdx.x, nblk = gridDim.x;
  uint64_t t0 = wall_clock64();
  const uint32_t count = n_in >> 1;
  const uint32_t span = (count + nblk - 1) / nblk;
  const uint32_t beg = min(blk * span, count), end = min(beg + span, count);
  Proj<F> acc;
  proj_set_zero<F>(acc);
  Proj<F> Q;
  for (int l = 0; l < NL; l++) { Q.X.l[l] = (tid * 7 + l * 13 + 5) & LMASK; Q.Y.l[l] = (tid * 3 + l * 11 + 1) & LMASK; Q.Z.l[l] = (tid + l) & LMASK; }
#pragma unroll 1
  for (uint32_t i = beg + tid; i < end; i += 64) {
    const uint32_t j = ((((i >> y) << 1) | 1u) << y) | (i & ((1u << y) - 1u));
    if (LOADS) proj_load_planar(Q, rows + (uint64_t)kk * (3 * NL) * n_in, n_in, j);
    proj_add<F>(acc, acc, Q);
  }
  if (TREE) {
#pragma unroll 1
    for (uint32_t s = 32; s >= 1; s >>= 1) {
      if (TREE >= 2 || (tid >= s && tid < 2 * s)) {
#pragma unroll
        for (int l = 0; l < NL; l++) {
          lds[(l)*64 + tid] = acc.X.l[l]; lds[(NL + l) * 64 + tid] = acc.Y.l[l]; lds[(2 * NL + l) * 64 + tid] = acc.Z.l[l];
        }
      }
      __syncthreads();
      if (TREE >= 2 || tid < s) {   // TREE == 2: every lane adds (the extra lanes' results are never used)
        Proj<F> R;
        const uint32_t src = (tid + s) & 63;
#pragma unroll
        for (int l = 0; l < NL; l++) {
          R.X.l[l] = lds[(l)*64 + src]; R.Y.l[l] = lds[(NL + l) * 64 + src]; R.Z.l[l] = lds[(2 * NL + l) * 64 + src];
        }
        if (TREE == 3 && tid >= s) { proj_set_zero<F>(R); proj_set_zero<F>(acc); }   // all lanes enter, the upper ones leave early inside
        proj_add<F>(acc, acc, R);
      }
      __syncthreads();
    }
  }
  const uint64_t o = ((uint64_t)kk * (nbits + 1) + y) * nblk + blk;
  if (tid == 0 || !TREE) proj_store(out + (o * 64 + (TREE ? 0 : tid)) * (3 * NL), acc);
  if (tid == 0) { stamps[2 * o] = t0; stamps[2 * o + 1] = wall_clock64(); }
}
template <int LOADS, int TREE>
__global__ void __launch_bounds__(64) k_bt(uint32_t* out, const uint32_t* rows, uint32_t n_in, uint32_t nbits, uint64_t* stamps) {
  bt_body<LOADS, TREE, 0>(out, rows, n_in, nbits, stamps);
}

int main() {
  const uint32_t n_in = 8192, nbits = 13, kc = 8, nblk = 8;
  size_t words = (size_t)kc * 39 * n_in;
  std::vector<uint32_t> h(words);
  uint64_t x = 88172645463325252ull;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)x & LMASK; }
  uint32_t *rows, *out; uint64_t* stamps;
  hipMalloc(&rows, words * 4); hipMemcpy(rows, h.data(), words * 4, hipMemcpyHostToDevice);
  hipMalloc(&out, (size_t)kc * 14 * nblk * 64 * 39 * 4);
  const int nwaves = nblk * nbits * kc;
  hipMalloc(&stamps, nwaves * 2 * 8 * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int variant = 0; variant < 6; variant++) {
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      dim3 g(nblk, nbits, kc);
      switch (variant) {
        case 0: k_bt<1, 1><<<g, 64>>>(out, rows, n_in, nbits, stamps); break;
        case 1: k_bt<0, 1><<<g, 64>>>(out, rows, n_in, nbits, stamps); break;
        case 2: k_bt<1, 0><<<g, 64>>>(out, rows, n_in, nbits, stamps); break;
        case 3: k_bt<0, 0><<<g, 64>>>(out, rows, n_in, nbits, stamps); break;
        case 4: k_bt<1, 2><<<g, 64>>>(out, rows, n_in, nbits, stamps); break;
        case 5: k_bt<1, 3><<<g, 64>>>(out, rows, n_in, nbits, stamps); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<uint64_t> st(nwaves * 2);
    hipMemcpy(st.data(), stamps, nwaves * 16, hipMemcpyDeviceToHost);
    uint64_t tmin = ~0ull, tmax = 0; double dur = 0, dmax = 0;
    for (int w = 0; w < nwaves; w++) { tmin = std::min(tmin, st[2 * w]); tmax = std::max(tmax, st[2 * w + 1]); double d = (double)(st[2 * w + 1] - st[2 * w]); dur += d; dmax = std::max(dmax, d); }
    const char* names[] = {"loads + tree", "no loads, tree", "loads, no tree", "no loads, no tree", "loads, all-lane tree", "all enter, upper lanes identity"};
    printf("%-20s waves=%d  %8.1f us kernel;  wave lifetime avg %.1f us max %.1f us; first start -> last end %.1f us (100 MHz clock)\n", names[variant], nwaves, ms * 1e3,
           dur / nwaves / 100.0, dmax / 100.0, (tmax - tmin) / 100.0);
  }
  return 0;
}
