/* Plain-C host of the C ABI (include/msm_hip.h): no Python, no torch -- what a cgo / JNI / N-API binding would do.
 *   gcc -O2 -Iinclude examples/msm_demo.c -Lmontgomery_amd -lmsm_hip -Wl,-rpath,'$ORIGIN/../montgomery_amd' -o examples/msm_demo
 *   examples/msm_demo [log2_n] [curve id]
 * Generates N points and scalars on the GPU, runs the MSM twice with different window sizes (the result is a
 * group element: it must not depend on c), then once more as K one-window shards recombined with
 * msm_combine_curve, the way the ranks of a multi-GPU run do.  Exit code 0 = all three agree. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "msm_hip.h"

static void die(msm_ctx* ctx, const char* what, int rc) {
  fprintf(stderr, "%s failed: %d (%s)\n", what, rc, ctx ? msm_last_error(ctx) : "no context");
  exit(1);
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 16;
  const int curve = argc > 2 ? atoi(argv[2]) : MSM_CURVE_BLS12_377_G1;
  const uint64_t n = 1ull << lg;
  msm_ctx* ctx = NULL;
  int rc = msm_ctx_create(&ctx, curve, 0);
  if (rc) die(NULL, "msm_ctx_create (no usable GPU? there is no CPU fallback)", rc);
  if ((rc = msm_generate_points(ctx, n, 42, NULL))) die(ctx, "msm_generate_points", rc);
  void* d_scalars = NULL; /* one device buffer per scalar handle, owned by the context until msm_device_free */
  if ((rc = msm_device_alloc(ctx, n * 32, &d_scalars))) die(ctx, "msm_device_alloc", rc);
  if ((rc = msm_generate_scalars(ctx, n, 43, d_scalars, NULL))) die(ctx, "msm_generate_scalars", rc);

  msm_opts opts;
  msm_result a, b, c3;
  memset(&opts, 0, sizeof opts);
  if ((rc = msm_run(ctx, d_scalars, n, 1, &opts, &a))) die(ctx, "msm_run", rc);
  opts.c = 11;
  if ((rc = msm_run(ctx, d_scalars, n, 1, &opts, &b))) die(ctx, "msm_run (c = 11)", rc);
  printf("N = 2^%d  c = %d  K = %d  %.3f ms  (accumulate %.3f ms, %llu pair additions)\n", lg, a.c, a.K, a.phase_ms[MSM_T_TOTAL],
         a.phase_ms[MSM_T_ACCUMULATE], (unsigned long long)a.n_pairs);
  printf("x = 0x");
  for (int i = 47; i >= 0; i--) printf("%02x", a.x[i]);
  printf("\n");
  int ok = memcmp(a.x, b.x, 48) == 0 && memcmp(a.y, b.y, 48) == 0 && a.is_infinity == b.is_infinity;

  { /* window shards: one window at a time, then the host combine */
    int32_t cc = 0, K = 0;
    memset(&opts, 0, sizeof opts);
    if ((rc = msm_plan(ctx, n, &opts, &cc, &K))) die(ctx, "msm_plan", rc);
    uint8_t* parts = (uint8_t*)malloc((size_t)K * 144);
    for (int k = 0; k < K; k++) {
      opts.c = cc; opts.k_lo = k; opts.k_hi = k + 1;
      if ((rc = msm_window_sums(ctx, d_scalars, n, 1, &opts, parts + (size_t)k * 144, NULL))) die(ctx, "msm_window_sums", rc);
    }
    if ((rc = msm_combine_curve(curve, parts, K, cc, &c3))) die(ctx, "msm_combine_curve", rc);
    free(parts);
    ok = ok && memcmp(a.x, c3.x, 48) == 0 && memcmp(a.y, c3.y, 48) == 0;
  }
  msm_ctx_destroy(ctx);
  printf(ok ? "OK: result independent of the window size and of the sharding\n" : "MISMATCH\n");
  return ok ? 0 : 2;
}
