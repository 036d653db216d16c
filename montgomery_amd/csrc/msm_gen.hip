// Synthetic input generation on the GPU (bench / large-N tests only):
//   generate_points  : P_i = sum_j T_j[idx_ij] with T_j[t] = (t + 1) * B_j, B_j = b_j * G  -- the scheme of
//                      `randomPointsFast` (reference src/curve-random.ts:14-92: 5 basis points, small tables,
//                      batch-normalised), with KNOWN discrete logs a_i = sum_j (idx_ij + 1) * b_j mod q so a
//                      2^26-point MSM can be checked in O(N): sum s_i P_i = (sum s_i a_i) * G
//   generate_scalars : uniform scalars < q by masked rejection sampling (src/curve-random.ts:151-194)
// Randomness is a counter-based splitmix64 stream, identical on host and device, keyed by (seed, index).
//
#include "msm_internal.h"

using namespace msmi;

namespace msm_gen {

static inline Q256 scalar_order(const msm_ctx* ctx) {
  const CurveInfo& ci = curve_info(ctx->curve);
  Q256 q;
  for (int j = 0; j < 4; j++) q.v[j] = (uint64_t)ci.q[2 * j] | ((uint64_t)ci.q[2 * j + 1] << 32);
  q.bits = ci.q_bits;
  return q;
}

static __global__ void __launch_bounds__(256) k_gen_scalars(uint64_t* out, uint64_t n, uint64_t seed, Q256 qq) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t q[4], s[4];
#pragma unroll
  for (int j = 0; j < 4; j++) q[j] = qq.v[j];
  draw_scalar(s, seed, i, q, qq.bits);
#pragma unroll
  for (int j = 0; j < 4; j++) out[i * 4 + j] = s[j];
}

int generate_scalars(msm_ctx* ctx, uint64_t n, uint64_t seed, void* dev_dst, uint8_t* host_out) {
  // dev_dst: caller-owned device buffer of n * 32 bytes; NULL (host copy only): the staging buffer carries them
  void* dst = dev_dst;
  if (!dst) {
    ctx->ensure(ctx->scal, std::max<uint64_t>(n, 1) * 32);
    dst = ctx->scal.p;
  }
  if (n) {
    hipLaunchKernelGGL(k_gen_scalars, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint64_t*)dst, n, seed,
                       scalar_order(ctx));
    if (host_out) HIPCHK(hipMemcpyAsync(host_out, dst, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  return MSM_OK;
}

struct U256 {
  uint64_t v[4];
};

static inline void addmod_q(U256& r, const U256& a, const U256& b, const uint64_t* q) {
  unsigned __int128 c = 0;
  U256 t;
  for (int i = 0; i < 4; i++) {
    c += (unsigned __int128)a.v[i] + b.v[i];
    t.v[i] = (uint64_t)c;
    c >>= 64;
  }
  if (c || ge_q(t.v, q)) {
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; i++) {
      unsigned __int128 d = (unsigned __int128)t.v[i] - q[i] - (uint64_t)br;
      t.v[i] = (uint64_t)d;
      br = (d >> 64) & 1;
    }
  }
  r = t;
}

// a_i = sum_j tbl_scalar[j][idx_ij] mod q for all n points (32-byte little-endian each), on the host cores: at 2^26 the
// plain loop is 3.4e8 modular additions, so it is split over threads
static inline void write_discrete_logs(uint8_t* a_out, uint64_t n, uint64_t seed, const std::vector<U256>& tbl_scalar, const uint64_t* q) {
  const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
  const unsigned nt = (unsigned)std::min<uint64_t>(hw, std::max<uint64_t>(1, n >> 14));
  auto work = [&](uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; i++) {
      U256 a = {{0, 0, 0, 0}};
      for (int j = 0; j < N_BASIS; j++) addmod_q(a, a, tbl_scalar[(size_t)j * TBL + table_index(seed, i, j)], q);
      for (int k = 0; k < 4; k++)
        for (int bb = 0; bb < 8; bb++) a_out[i * 32 + 8 * k + bb] = (uint8_t)(a.v[k] >> (8 * bb));
    }
  };
  if (nt <= 1) { work(0, n); return; }
  std::vector<std::thread> th;
  const uint64_t per = (n + nt - 1) / nt;
  for (unsigned k = 0; k < nt; k++) th.emplace_back(work, std::min<uint64_t>(n, k * per), std::min<uint64_t>(n, (k + 1) * per));
  for (auto& x : th) x.join();
}

int generate_points(msm_ctx* ctx, uint64_t n, uint64_t seed, uint8_t* a_out) {
  using namespace msm_host;
  if (n >= (1ull << 30)) return MSM_ERR_ARG;
  const Curve6& C = ctx->hc;
  const CurveInfo& ci = curve_info(ctx->curve);
  const Q256 qq = scalar_order(ctx);
  const uint64_t* q = qq.v;
  // generator in host Montgomery form
  Proj6 G;
  {
    Fe6 t;
    for (int i = 0; i < 6; i++) t.v[i] = (uint64_t)ci.gx[2 * i] | ((uint64_t)ci.gx[2 * i + 1] << 32);
    C.F.mul(G.X, t, ctx->k_dev_to_host);
    for (int i = 0; i < 6; i++) t.v[i] = (uint64_t)ci.gy[2 * i] | ((uint64_t)ci.gy[2 * i + 1] << 32);
    C.F.mul(G.Y, t, ctx->k_dev_to_host);
    G.Z = C.F.one;
  }
  const size_t cb = ctx->coord_bytes(), pb = 2 * cb;   // wire coordinate / point bytes of this curve
  std::vector<uint8_t> wire((size_t)N_BASIS * TBL * pb);
  std::vector<U256> tbl_scalar((size_t)N_BASIS * TBL);
  Fe6 one_plain = {{1, 0, 0, 0, 0, 0}};
  for (int j = 0; j < N_BASIS; j++) {
    U256 b;
    draw_scalar(b.v, seed ^ 0xba5e5ull, (uint64_t)j, q, qq.bits);
    // B = b * G (MSB-first double and add)
    Proj6 B = C.zero();
    for (int bit = 255; bit >= 0; bit--) {
      B = C.dbl(B);
      if ((b.v[bit / 64] >> (bit % 64)) & 1) B = C.add(B, G);
    }
    // the table's multiples, then ONE inversion for all of them (Montgomery's trick; the reference batch-normalises too,
    // src/curve-random.ts:60-75)
    std::vector<Proj6> mult(TBL);
    std::vector<Fe6> pre(TBL);
    Proj6 acc = C.zero();
    U256 sacc = {{0, 0, 0, 0}};
    Fe6 run = C.F.one;
    for (int t = 0; t < TBL; t++) {
      acc = C.add(acc, B);
      addmod_q(sacc, sacc, b, q);
      tbl_scalar[(size_t)j * TBL + t] = sacc;
      mult[t] = acc;
      pre[t] = run;
      if (!C.is_zero(acc)) C.F.mul(run, run, acc.Z);
    }
    Fe6 inv_run;
    C.F.inv(inv_run, run);
    for (int t = TBL - 1; t >= 0; t--) {
      Fe6 zi, x, y;
      uint8_t* w = &wire[((size_t)j * TBL + t) * pb];
      if (C.is_zero(mult[t])) {
        memset(w, 0, pb);
        continue;
      }
      C.F.mul(zi, inv_run, pre[t]);
      C.F.mul(inv_run, inv_run, mult[t].Z);
      C.F.mul(x, mult[t].X, zi);
      C.F.mul(y, mult[t].Y, zi);
      C.F.mul(x, x, one_plain);
      C.F.mul(y, y, one_plain);
      for (size_t i = 0; i < cb / 8; i++)
        for (int k = 0; k < 8; k++) {
          w[8 * i + k] = (uint8_t)(x.v[i] >> (8 * k));
          w[cb + 8 * i + k] = (uint8_t)(y.v[i] >> (8 * k));
        }
    }
  }
  ScopedDevBuf d_wire, d_tbl;   // freed on every path
  ctx->ensure(d_wire, wire.size());
  ctx->ensure(d_tbl, (size_t)N_BASIS * TBL * msm::ROW_WORDS * 4);
  HIPCHK(hipMemcpyAsync(d_wire.p, wire.data(), wire.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->errflag.p, 0, 4, ctx->stream));
  W_LAUNCH(ctx, msm::k_points_from_wire, dim3((N_BASIS * TBL + 255) / 256), dim3(256), 0, ctx->stream, (uint32_t*)d_tbl.p,
                     (const uint32_t*)d_wire.p, (uint64_t)N_BASIS * TBL, 1, (uint32_t*)ctx->errflag.p);
  ctx->n_points = 0;
  ctx->drop_tables();   // window tables belong to the points they were built from
  ctx->ensure(ctx->rows, std::max<uint64_t>(n, 1) * msm::ROW_WORDS * 4);
  if (n)
    W_LAUNCH(ctx, k_gen_points, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)ctx->rows.p,
                       (const uint32_t*)d_tbl.p, n, seed);
  HIPCHK(hipMemcpyAsync(ctx->h_info, ctx->errflag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  ctx->release(d_wire);
  ctx->release(d_tbl);
  if (ctx->h_info[0]) {
    ctx->err = "msm_generate_points: table point failed validation";
    return MSM_ERR_POINT;
  }
  ctx->n_points = n;
  if (a_out) write_discrete_logs(a_out, n, seed, tbl_scalar, q);
  return MSM_OK;
}

// Edwards curve: the same scheme over the unified extended addition (host: TeCurve6, device: k_te_gen_points)
int generate_points_te(msm_ctx* ctx, uint64_t n, uint64_t seed, uint8_t* a_out) {
  using namespace msm_host;
  if (n >= (1ull << 30)) return MSM_ERR_ARG;
  const TeCurve6& C = ctx->hte;
  const Q256 qq = scalar_order(ctx);
  const uint64_t* q = qq.v;
  auto words8_to_host = [&](const uint32_t* w8) {   // device Montgomery (2^270) words -> host Montgomery
    Fe6 t = {{0, 0, 0, 0, 0, 0}}, r;
    for (int i = 0; i < 4; i++) t.v[i] = (uint64_t)w8[2 * i] | ((uint64_t)w8[2 * i + 1] << 32);
    C.F.mul(r, t, ctx->k_te_to_host);
    return r;
  };
  Ext6 G;
  G.X = words8_to_host(msm::Fp253::GXW);
  G.Y = words8_to_host(msm::Fp253::GYW);
  G.Z = C.F.one;
  C.F.mul(G.T, G.X, G.Y);
  std::vector<uint8_t> wire((size_t)N_BASIS * TBL * 64);
  std::vector<U256> tbl_scalar((size_t)N_BASIS * TBL);
  Fe6 one_plain = {{1, 0, 0, 0, 0, 0}};
  for (int j = 0; j < N_BASIS; j++) {
    U256 b;
    draw_scalar(b.v, seed ^ 0xba5e5ull, (uint64_t)j, q, qq.bits);
    Ext6 B = C.zero();
    for (int bit = 255; bit >= 0; bit--) {   // the unified addition doubles as well
      B = C.add(B, B);
      if ((b.v[bit / 64] >> (bit % 64)) & 1) B = C.add(B, G);
    }
    std::vector<Ext6> mult(TBL);
    std::vector<Fe6> pre(TBL);
    Ext6 acc = C.zero();
    U256 sacc = {{0, 0, 0, 0}};
    Fe6 run = C.F.one;
    for (int t = 0; t < TBL; t++) {   // Z is never zero on the Edwards curve (complete addition law)
      acc = C.add(acc, B);
      addmod_q(sacc, sacc, b, q);
      tbl_scalar[(size_t)j * TBL + t] = sacc;
      mult[t] = acc;
      pre[t] = run;
      C.F.mul(run, run, acc.Z);
    }
    Fe6 inv_run;
    C.F.inv(inv_run, run);
    for (int t = TBL - 1; t >= 0; t--) {
      Fe6 zi, x, y;
      C.F.mul(zi, inv_run, pre[t]);
      C.F.mul(inv_run, inv_run, mult[t].Z);
      C.F.mul(x, mult[t].X, zi);
      C.F.mul(y, mult[t].Y, zi);
      C.F.mul(x, x, one_plain);
      C.F.mul(y, y, one_plain);
      uint8_t* w = &wire[((size_t)j * TBL + t) * 64];
      for (int i = 0; i < 4; i++)
        for (int k = 0; k < 8; k++) {
          w[8 * i + k] = (uint8_t)(x.v[i] >> (8 * k));
          w[32 + 8 * i + k] = (uint8_t)(y.v[i] >> (8 * k));
        }
    }
  }
  ScopedDevBuf d_wire, d_tbl, d_pts;
  ctx->ensure(d_wire, wire.size());
  ctx->ensure(d_tbl, (size_t)N_BASIS * TBL * msm::te::TE_ROW_WORDS * 4);
  ctx->ensure(d_pts, std::max<uint64_t>(n, 1) * 64);
  HIPCHK(hipMemcpyAsync(d_wire.p, wire.data(), wire.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->errflag.p, 0, 4, ctx->stream));
  hipLaunchKernelGGL(msm::te::k_te_points_from_wire, dim3((N_BASIS * TBL + 255) / 256), dim3(256), 0, ctx->stream, (uint32_t*)d_tbl.p,
                     (const uint32_t*)d_wire.p, (uint64_t)N_BASIS * TBL, 1, (uint32_t*)ctx->errflag.p);
  ctx->n_points = 0;
  ctx->drop_tables();   // window tables belong to the points they were built from
  ctx->ensure(ctx->rows, std::max<uint64_t>(n, 1) * msm::te::TE_ROW_WORDS * 4);
  if (n) {
    const uint32_t grid = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(msm::te::k_te_gen_points, dim3(grid), dim3(256), 0, ctx->stream, (uint32_t*)d_pts.p, (const uint32_t*)d_tbl.p,
                       n, seed);
    hipLaunchKernelGGL(msm::te::k_te_points_from_wire, dim3(grid), dim3(256), 0, ctx->stream, (uint32_t*)ctx->rows.p,
                       (const uint32_t*)d_pts.p, n, 0, (uint32_t*)ctx->errflag.p);
  }
  HIPCHK(hipMemcpyAsync(ctx->h_info, ctx->errflag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  ctx->release(d_wire);
  ctx->release(d_tbl);
  ctx->release(d_pts);
  if (ctx->h_info[0]) {
    ctx->err = "msm_generate_points: table point failed validation";
    return MSM_ERR_POINT;
  }
  ctx->n_points = n;
  if (a_out) write_discrete_logs(a_out, n, seed, tbl_scalar, q);
  return MSM_OK;
}

}  // namespace msm_gen
