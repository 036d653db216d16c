// Round-5 verdict item 8, priced before built: the reduction half of a Montgomery product is a product with the CONSTANT p --
// could int8 MFMA do the 13-row m * p accumulation of a wave's 64 elements faster than the VALU rows of fe_mul?
//
// Shape of the MFMA form: the 13 quotient digits of an element (30 bits each) as 52 unsigned bytes; p as 48 bytes; the product
// M * p is the contraction  out[column c] = sum_j Mb[j] * Pb[c - j]  over byte positions, i.e. D = T * B with T the
// 128 x 64 Toeplitz matrix of p's bytes (constant) and B = [byte j][element] (64 elements = 2 column blocks of 32):
//   4 row blocks (128 byte columns) x 2 k steps (64 byte positions) x 2 column blocks = 16 v_mfma_i32_32x32x32_i8 per wave,
// with the element on the lane (col = lane & 31) and 16 byte columns of it per accumulator tile in the registers of lanes
// l and l + 32 -- the orientation that needs no LDS transpose.  What the form costs on top of the MFMAs:
//   (a) the quotient digits must exist BEFORE the contraction: the interleaved (CIOS) product gets digit i from row i's
//       running column for one multiply; separated, M = (T mod R) * (-p^-1) mod R is a 13 x 13 low product = 91 multiply-adds;
//   (b) 13 limbs -> 52 bytes per element (unsigned bytes need 9 bits in a signed i8 operand: 7-bit pieces make it 5 pieces
//       per limb and 20 % more MFMAs -- not charged here);
//   (c) 128 i32 byte columns per element back to 13 limbs of 30 bits with carries.
// This benchmark times, per wave and reduction, in shader cycles at one and two waves per SIMD:
//   valu    the reduction rows as fe_mul has them (quotient digit + 12 multiply-adds + shift per row; p = 1 mod 2^30)
//   mfma    the 16 MFMAs alone (operands constant: the instruction's cycles do not depend on data)
//   fold    (c) alone: 128 columns -> 13 limbs, in-lane, no cross-lane traffic (a LOWER bound: half of every element's columns
//           sit in lane l + 32)
//   split   (b) alone
// If mfma + fold + split is not below valu / 1.5 the MFMA form is closed without its remaining costs ((a), the lane-half
// exchange, the signed-byte pieces) having to be built.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ const uint32_t P377[13] = {0x1u, 0x14230000u, 0x8u, 0x2d7510cu, 0x9480017u, 0xd88bee8u, 0x1138f1efu, 0x367cc03du,
                                      0x93b1a22u, 0x1701b285u, 0xeac63b0u, 0x1185f144u, 0x1ae3au};

// the reduction rows of fe_mul for BLS12-377 (p = 1 mod 2^30: the quotient digit is -acc0 mod 2^30, row i adds m * p[1..12])
__global__ void __launch_bounds__(256) k_valu(uint32_t* out, uint32_t iters) {
  uint64_t acc[14];
  for (int i = 0; i < 14; i++) acc[i] = (uint64_t)(threadIdx.x * 2654435761u + i * 40503u) & 0x3FFFFFFFull;
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 13; r++) {
      const uint32_t m = (0u - (uint32_t)acc[0]) & 0x3FFFFFFFu;
      const uint64_t carry = (acc[0] + m) >> 30;
#pragma unroll
      for (int j = 1; j < 13; j++) acc[j - 1] = acc[j] + (uint64_t)m * P377[j] + (j == 1 ? carry : 0);
      acc[12] = acc[13];
      acc[13] = 0;
    }
  }
  uint32_t x = 0;
  for (int i = 0; i < 14; i++) x ^= (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

__global__ void __launch_bounds__(256) k_mfma(uint32_t* out, uint32_t iters) {
  v4i a[2], b[4];
  for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) a[i][j] = (int)(threadIdx.x * 0x01010101u + i + j);
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) b[i][j] = (int)(threadIdx.x * 0x01030507u + i * 7 + j);
  v16i acc[8];
  for (int t = 0; t < 8; t++) for (int j = 0; j < 16; j++) acc[t][j] = 0;
  for (uint32_t it = 0; it < iters; it++) {
    // one reduction of the wave's 64 elements: 4 row blocks x 2 column blocks of accumulators, 2 k steps each
#pragma unroll
    for (int t = 0; t < 8; t++) {
      acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], b[t & 3], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[1], b[(t + 1) & 3], acc[t], 0, 0, 0);
    }
  }
  uint32_t x = 0;
  for (int t = 0; t < 8; t++) for (int j = 0; j < 16; j++) x ^= (uint32_t)acc[t][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

// 128 byte columns (each < 2^22) of one element -> 13 limbs of 30 bits with carries
__global__ void __launch_bounds__(256) k_fold(uint32_t* out, uint32_t iters) {
  uint32_t col[128];
  for (int i = 0; i < 128; i++) col[i] = (threadIdx.x * 2654435761u + i * 97u) & 0x3FFFFFu;
  uint32_t x = 0;
  for (uint32_t it = 0; it < iters; it++) {
    uint64_t carry = 0;
    uint32_t limb[13];
#pragma unroll
    for (int l = 0; l < 13; l++) {
      // limb l covers bits [30 l, 30 l + 30): byte columns floor(30 l / 8) .. floor((30 l + 29) / 8)
      uint64_t v = carry;
      const int b0 = (30 * l) / 8, sh = (30 * l) % 8;
#pragma unroll
      for (int k = 0; k < 5; k++) v += (uint64_t)col[b0 + k] << (8 * k);
      v >>= sh;
      limb[l] = (uint32_t)v & 0x3FFFFFFFu;
      carry = v >> 30;
    }
#pragma unroll
    for (int l = 0; l < 13; l++) { x ^= limb[l]; col[l * 9] ^= limb[l] & 0xFFFFu; }   // (static indices: the columns stay in registers)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

// 13 limbs of 30 bits -> 52 bytes packed four to a dword (the MFMA operand)
__global__ void __launch_bounds__(256) k_split(uint32_t* out, uint32_t iters) {
  uint32_t limb[13];
  for (int i = 0; i < 13; i++) limb[i] = (threadIdx.x * 2654435761u + i * 1013u) & 0x3FFFFFFFu;
  uint32_t x = 0;
  for (uint32_t it = 0; it < iters; it++) {
    uint32_t w[13];
    // 390 bits as 13 dwords of four bytes: dword d = bits [32 d, 32 d + 32) of the limb string
#pragma unroll
    for (int d = 0; d < 13; d++) {
      const int bit = 32 * d, l0 = bit / 30, sh = bit % 30;
      uint64_t v = (uint64_t)limb[l0 % 13] >> sh;
      v |= (uint64_t)limb[(l0 + 1) % 13] << (30 - sh);
      if (60 - sh < 32) v |= (uint64_t)limb[(l0 + 2) % 13] << (60 - sh);
      w[d] = (uint32_t)v;
    }
#pragma unroll
    for (int d = 0; d < 13; d++) { x ^= w[d]; limb[d] = (limb[d] + w[(d + 1) % 13]) & 0x3FFFFFFFu; }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <class K>
static double cycles_per_wave_iter(K kernel, int waves_per_simd, uint32_t iters, uint32_t* d_out, int n_cu, double ghz) {
  const int blocks = n_cu * waves_per_simd;   // 256 threads = 4 waves = one per SIMD and block
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, iters / 8);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  // every SIMD runs `waves_per_simd` waves for `iters` iterations: SIMD cycles per (wave, iteration)
  return ms * 1e-3 * ghz * 1e9 / ((double)iters * waves_per_simd);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;   // nominal; the ratio between the kernels is what counts
  uint32_t* d_out;
  CHECK(hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4));
  printf("cycles per wave and reduction of its 64 elements at the nominal %.2f GHz (SIMD cycles; lower is better)\n", ghz);
  for (int w : {1, 2, 4}) {
    const double v = cycles_per_wave_iter(k_valu, w, 20000, d_out, n_cu, ghz);
    const double m = cycles_per_wave_iter(k_mfma, w, 20000, d_out, n_cu, ghz);
    const double f = cycles_per_wave_iter(k_fold, w, 20000, d_out, n_cu, ghz);
    const double s = cycles_per_wave_iter(k_split, w, 20000, d_out, n_cu, ghz);
    printf("%d wave(s) per SIMD: valu rows %7.0f | 16 x mfma_i32_32x32x32_i8 %7.0f + fold %7.0f + split %7.0f = %7.0f  -> MFMA form / VALU rows = %.2f (needs <= 0.67)\n",
           w, v, m, f, s, m + f + s, (m + f + s) / v);
  }
  return 0;
}
