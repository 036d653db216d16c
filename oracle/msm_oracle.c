/* CPU oracle, C port (TEST INFRASTRUCTURE ONLY -- never linked into the product).
 *
 * Plain-C restatement of the reference's batched-affine Pippenger MSM for BLS12-377 G1, used
 *   (1) by tests/ as a fast checker at sizes the pure-Python oracle (oracle/msm_oracle.py) cannot reach,
 *   (2) by bench.py's `cpu_baseline` leg (kind "port").
 * Parity status: pinned through tests/test_oracle_kat.py, which checks this port against the Python
 * oracle and against the reference's known-answer material (see the header of msm_oracle.py).
 *
 * Follows, phase by phase, the reference at /root/reference (paths relative to it):
 *   field: Montgomery multiplication        src/wasm/multiply-montgomery.ts:58-136 (here 6 x 64-bit limbs, R = 2^384)
 *   inverse                                 src/wasm/inverse.ts:191-218 (here a^(p-2))
 *   GLV decompose                           src/wasm/glv.ts:68-169, constants :35-63, lattice src/glv/glv.ts:21-50
 *   preparePointsAndScalars                 src/msm-batched-affine.ts:350-421
 *   slice + count / integrate / sortPoints  src/msm-batched-affine.ts:175-203, :423-447, :456-502
 *   accumulation rounds + batchAddNew       src/msm-batched-affine.ts:243-282, src/curve-affine.ts:376-458
 *   reduceBucketsColumnProjective           src/msm-batched-affine.ts:556-583
 *   final sum                               src/msm-batched-affine.ts:312-333
 *   projective add / double                 src/curve-projective.ts:51-160, :202-253
 * Threads: every phase of a window runs on all host cores, entries split across threads for slicing and sorting,
 * buckets split across threads for the accumulation rounds and the reduction (the reference's SPMD layout).
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -shared).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t v[6]; } fe;

/* p, src/concrete/bls12-377.params.ts:11-12 */
static const fe P = {{0x8508c00000000001ull, 0x170b5d4430000000ull, 0x1ef3622fba094800ull,
                      0x1a22d9f300f5138full, 0xc63b05c06ca1493bull, 0x01ae3a4617c510eaull}};
static uint64_t PINV;   /* -p^-1 mod 2^64 */
static fe ONE, R2, BETA_M, B_M;
static int inited = 0;

static int fe_ge(const fe* a, const fe* b) {
  for (int i = 5; i >= 0; i--) {
    if (a->v[i] > b->v[i]) return 1;
    if (a->v[i] < b->v[i]) return 0;
  }
  return 1;
}
static int fe_is_zero(const fe* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3] | a->v[4] | a->v[5]) == 0; }
static int fe_eq(const fe* a, const fe* b) { return memcmp(a, b, sizeof(fe)) == 0; }
static void fe_sub_raw(fe* r, const fe* a, const fe* b) {
  uint64_t br = 0;
  for (int i = 0; i < 6; i++) {
    u128 d = (u128)a->v[i] - b->v[i] - br;
    r->v[i] = (uint64_t)d;
    br = (uint64_t)(d >> 64) & 1;
  }
}
static void fe_add(fe* r, const fe* a, const fe* b) {
  u128 c = 0;
  fe t;
  for (int i = 0; i < 6; i++) { c += (u128)a->v[i] + b->v[i]; t.v[i] = (uint64_t)c; c >>= 64; }
  if (c || fe_ge(&t, &P)) fe_sub_raw(&t, &t, &P);
  *r = t;
}
static void fe_sub(fe* r, const fe* a, const fe* b) {
  fe t;
  if (fe_ge(a, b)) fe_sub_raw(&t, a, b);
  else { fe u; fe_sub_raw(&u, b, a); fe_sub_raw(&t, &P, &u); }
  *r = t;
}
static void fe_neg(fe* r, const fe* a) { if (fe_is_zero(a)) *r = *a; else fe_sub_raw(r, &P, a); }
static void fe_mul(fe* r, const fe* a, const fe* b) {
  uint64_t t[8] = {0};
  for (int i = 0; i < 6; i++) {
    u128 c = 0;
    for (int j = 0; j < 6; j++) { c += (u128)a->v[i] * b->v[j] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[6] = (uint64_t)c; t[7] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * PINV;
    c = ((u128)m * P.v[0] + t[0]) >> 64;
    for (int j = 1; j < 6; j++) { c += (u128)m * P.v[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[5] = (uint64_t)c; t[6] = t[7] + (uint64_t)(c >> 64); t[7] = 0;
  }
  fe o;
  memcpy(o.v, t, 48);
  if (t[6] || fe_ge(&o, &P)) fe_sub_raw(&o, &o, &P);
  *r = o;
}
static void fe_inv(fe* r, const fe* a) {
  fe e, two = {{2, 0, 0, 0, 0, 0}}, acc = ONE;
  fe_sub_raw(&e, &P, &two);
  for (int bit = 376; bit >= 0; bit--) {
    fe_mul(&acc, &acc, &acc);
    if ((e.v[bit / 64] >> (bit % 64)) & 1) fe_mul(&acc, &acc, a);
  }
  *r = acc;
}
static void fe_from_bytes(fe* r, const uint8_t* b) {
  for (int i = 0; i < 6; i++) { uint64_t v = 0; for (int j = 0; j < 8; j++) v |= (uint64_t)b[8 * i + j] << (8 * j); r->v[i] = v; }
}
static void fe_to_bytes(uint8_t* b, const fe* a) {
  for (int i = 0; i < 6; i++) for (int j = 0; j < 8; j++) b[8 * i + j] = (uint8_t)(a->v[i] >> (8 * j));
}

static void init_consts(void) {
  if (inited) return;
  uint64_t x = P.v[0];
  for (int i = 0; i < 6; i++) x *= 2 - P.v[0] * x;
  PINV = (uint64_t)0 - x;
  fe t = {{1, 0, 0, 0, 0, 0}};
  for (int i = 0; i < 384; i++) fe_add(&t, &t, &t);
  ONE = t;
  for (int i = 0; i < 384; i++) fe_add(&t, &t, &t);
  R2 = t;
  /* beta, src/concrete/bls12-377.params.ts:33-34 */
  static const uint8_t beta_be[48] = {0x01,0xae,0x3a,0x46,0x17,0xc5,0x10,0xea,0xbc,0x87,0x56,0xba,0x8f,0x8c,0x52,0x4e,
    0xb8,0x88,0x2a,0x75,0xcc,0x9b,0xc8,0xe3,0x59,0x06,0x4e,0xe8,0x22,0xfb,0x5b,0xff,0xd1,0xe9,0x45,0x77,0x9f,0xff,0xff,0xff,
    0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff};
  uint8_t le[48];
  for (int i = 0; i < 48; i++) le[i] = beta_be[47 - i];
  fe bt;
  fe_from_bytes(&bt, le);
  fe_mul(&BETA_M, &bt, &R2);
  B_M = ONE; /* b = 1 */
  inited = 1;
}

/* ---------------------------------------------------------------- curve */
typedef struct { fe x, y; int inf; } aff;
typedef struct { fe X, Y, Z; } proj;

static void proj_zero(proj* P_) { memset(P_, 0, sizeof(*P_)); P_->Y = ONE; }
static int proj_is_zero(const proj* P_) { return fe_is_zero(&P_->Z); }
static void proj_dbl(proj* R, const proj* Q) {
  if (proj_is_zero(Q)) { proj_zero(R); return; }
  fe w, s, ss, sss, Rr, B, h, t, u, B2, B4, B8;
  fe_mul(&t, &Q->X, &Q->X); fe_add(&w, &t, &t); fe_add(&w, &w, &t);
  fe_mul(&s, &Q->Y, &Q->Z); fe_mul(&ss, &s, &s); fe_mul(&sss, &s, &ss);
  fe_mul(&Rr, &Q->Y, &s); fe_mul(&B, &Q->X, &Rr);
  fe_add(&B2, &B, &B); fe_add(&B4, &B2, &B2); fe_add(&B8, &B4, &B4);
  fe_mul(&h, &w, &w); fe_sub(&h, &h, &B8);
  proj O;
  fe_mul(&t, &h, &s); fe_add(&O.X, &t, &t);
  fe_sub(&u, &B4, &h); fe_mul(&u, &w, &u);
  fe_mul(&t, &Rr, &Rr); fe_add(&t, &t, &t); fe_add(&t, &t, &t); fe_add(&t, &t, &t);
  fe_sub(&O.Y, &u, &t);
  fe_add(&t, &sss, &sss); fe_add(&t, &t, &t); fe_add(&O.Z, &t, &t);
  *R = O;
}
static void proj_add(proj* R, const proj* A, const proj* Bq) {
  if (proj_is_zero(A)) { *R = *Bq; return; }
  if (proj_is_zero(Bq)) { *R = *A; return; }
  fe Y1Z2, X1Z2, Z1Z2, u, v, t, uu, vv, vvv, Rr, Aa, t2;
  fe_mul(&Y1Z2, &A->Y, &Bq->Z); fe_mul(&X1Z2, &A->X, &Bq->Z); fe_mul(&Z1Z2, &A->Z, &Bq->Z);
  fe_mul(&t, &Bq->Y, &A->Z); fe_sub(&u, &t, &Y1Z2);
  fe_mul(&t, &Bq->X, &A->Z); fe_sub(&v, &t, &X1Z2);
  if (fe_is_zero(&v)) { if (fe_is_zero(&u)) { proj_dbl(R, A); return; } proj_zero(R); return; }
  fe_mul(&uu, &u, &u); fe_mul(&vv, &v, &v); fe_mul(&vvv, &v, &vv); fe_mul(&Rr, &vv, &X1Z2);
  fe_mul(&Aa, &uu, &Z1Z2); fe_sub(&Aa, &Aa, &vvv); fe_add(&t, &Rr, &Rr); fe_sub(&Aa, &Aa, &t);
  proj O;
  fe_mul(&O.X, &v, &Aa);
  fe_sub(&t, &Rr, &Aa); fe_mul(&t, &u, &t); fe_mul(&t2, &vvv, &Y1Z2); fe_sub(&O.Y, &t, &t2);
  fe_mul(&O.Z, &vvv, &Z1Z2);
  *R = O;
}

/* batch affine addition S_i = G_i + H_i into G_i's storage, one inversion per call
 * (batchAddNew, src/curve-affine.ts:376-458).  scratch: n field elements. */
static void batch_add(aff** G, aff** H, size_t n, fe* pre, uint8_t* kind) {
  fe acc = ONE;
  for (size_t i = 0; i < n; i++) {
    aff *g = G[i], *h = H[i];
    fe den = ONE;
    kind[i] = 0;
    if (g->inf) kind[i] = 2;               /* result = h */
    else if (h->inf) kind[i] = 3;          /* result = g */
    else if (fe_eq(&g->x, &h->x)) {
      if (fe_eq(&g->y, &h->y) && !fe_is_zero(&g->y)) { kind[i] = 1; fe_add(&den, &g->y, &g->y); }
      else kind[i] = 4;                    /* zero */
    } else fe_sub(&den, &h->x, &g->x);
    pre[i] = acc;
    fe_mul(&acc, &acc, &den);
    /* stash the denominator in-place of nothing: recomputed below */
  }
  fe inv;
  fe_inv(&inv, &acc);
  for (size_t ii = n; ii-- > 0;) {
    aff *g = G[ii], *h = H[ii];
    fe den = ONE, d, m, mm, x3, y3, t;
    if (kind[ii] == 0) fe_sub(&den, &h->x, &g->x);
    else if (kind[ii] == 1) fe_add(&den, &g->y, &g->y);
    fe_mul(&d, &inv, &pre[ii]);
    fe_mul(&inv, &inv, &den);
    if (kind[ii] == 2) { *g = *h; continue; }
    if (kind[ii] == 3) continue;
    if (kind[ii] == 4) { g->inf = 1; continue; }
    if (kind[ii] == 1) { fe_mul(&t, &g->x, &g->x); fe_add(&m, &t, &t); fe_add(&m, &m, &t); fe_mul(&m, &m, &d); }
    else { fe_sub(&t, &h->y, &g->y); fe_mul(&m, &t, &d); }
    fe_mul(&mm, &m, &m);
    fe_sub(&x3, &mm, &g->x); fe_sub(&x3, &x3, &h->x);
    fe_sub(&t, &g->x, &x3); fe_mul(&y3, &m, &t); fe_sub(&y3, &y3, &g->y);
    g->x = x3; g->y = y3;
  }
}

/* ---------------------------------------------------------------- GLV (src/wasm/glv.ts:68-169) */
/* lattice for BLS12-377 (egcdStopEarly(lambda, q)): v00 = 1, v01 = t + 1, v10 = -t, v11 = 1,
 * t = 0x452217cc900000010a11800000000000; m0 = -438, m1 = trunc(-2^261 * t / q); m = 145, k = 116 */
static const uint64_t T_LO = 0x0a11800000000000ull, T_HI = 0x452217cc90000001ull;
static const uint64_t Q4[4] = {0x0a11800000000001ull, 0x59aa76fed0000001ull, 0x60b44d1e5c37b001ull, 0x12ab655e9a2ca556ull};
static uint64_t M1ABS[3]; /* |m1|, 135 bits */
static int glv_inited = 0;

/* generic little helpers on little-endian 64-bit limb arrays */
static void bn_mul(uint64_t* r, const uint64_t* a, int na, const uint64_t* b, int nb) {
  memset(r, 0, 8 * (size_t)(na + nb));
  for (int i = 0; i < na; i++) {
    u128 c = 0;
    for (int j = 0; j < nb; j++) { c += (u128)a[i] * b[j] + r[i + j]; r[i + j] = (uint64_t)c; c >>= 64; }
    r[i + nb] = (uint64_t)c;
  }
}
static void bn_shr(uint64_t* r, int nr, const uint64_t* x, int nx, int sh) {
  int ws = sh / 64, bs = sh % 64;
  for (int i = 0; i < nr; i++) {
    uint64_t lo = i + ws < nx ? x[i + ws] : 0, hi = i + ws + 1 < nx ? x[i + ws + 1] : 0;
    r[i] = bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
  }
}
static void bn_addsub(uint64_t* r, int n, const uint64_t* a, int na, int sub) {
  u128 c = sub ? 1 : 0;
  for (int i = 0; i < n; i++) {
    uint64_t ai = i < na ? a[i] : 0;
    if (sub) ai = ~ai;
    c += (u128)r[i] + ai; r[i] = (uint64_t)c; c >>= 64;
  }
}
static void glv_init(void) {
  if (glv_inited) return;
  /* |m1| = floor(2^261 * t / q) by long division, bit by bit (one-time) */
  uint64_t num[7] = {0}, rem[5] = {0}, quo[7] = {0};
  /* num = t << 261 : t is 127 bits -> 388 bits */
  uint64_t tt[2] = {T_LO, T_HI};
  for (int i = 0; i < 2; i++) {
    int pos = 261 + 64 * i, w = pos / 64, b = pos % 64;
    num[w] |= tt[i] << b;
    if (b) num[w + 1] |= tt[i] >> (64 - b);
  }
  for (int bit = 7 * 64 - 1; bit >= 0; bit--) {
    /* rem = rem * 2 + bit */
    for (int i = 4; i > 0; i--) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
    rem[0] = (rem[0] << 1) | ((num[bit / 64] >> (bit % 64)) & 1);
    /* if rem >= q: rem -= q */
    int ge = 1;
    if (rem[4] == 0) { for (int i = 3; i >= 0; i--) { if (rem[i] > Q4[i]) break; if (rem[i] < Q4[i]) { ge = 0; break; } } }
    if (ge) { bn_addsub(rem, 5, Q4, 4, 1); quo[bit / 64] |= 1ull << (bit % 64); }
  }
  M1ABS[0] = quo[0]; M1ABS[1] = quo[1]; M1ABS[2] = quo[2];
  glv_inited = 1;
}
/* s (4 limbs, < q) -> magnitudes a0, a1 (2 limbs each) and sign flags */
static void glv_decompose(const uint64_t* s, uint64_t* a0, int* n0, uint64_t* a1, int* n1) {
  uint64_t shi[3], prod[8], X0[3], X1[3], one[1];
  bn_shr(shi, 3, s, 4, 116);
  uint64_t m0abs[1] = {438};
  bn_mul(prod, shi, 3, m0abs, 1);
  bn_shr(X0, 3, prod, 4, 145);
  one[0] = (prod[144 / 64] >> (144 % 64)) & 1;
  bn_addsub(X0, 3, one, 1, 0);
  bn_mul(prod, shi, 3, M1ABS, 3);
  bn_shr(X1, 3, prod, 6, 145);
  one[0] = (prod[144 / 64] >> (144 % 64)) & 1;
  bn_addsub(X1, 3, one, 1, 0);
  /* x0 = -X0, x1 = -X1 (m0, m1 < 0).  s0 = s + v00 x0 + v01 x1 = s - X0 - (t + 1) X1; s1 = v10 x0 + v11 x1 = t X0 - X1 */
  uint64_t acc[6] = {s[0], s[1], s[2], s[3], 0, 0}, term[6], v01[2] = {T_LO + 1, T_HI}, tt[2] = {T_LO, T_HI};
  bn_addsub(acc, 6, X0, 3, 1);
  bn_mul(term, X1, 3, v01, 2); term[5] = 0;
  bn_addsub(acc, 6, term, 5, 1);
  *n0 = (int)(acc[5] >> 63);
  if (*n0) { for (int i = 0; i < 6; i++) acc[i] = ~acc[i]; one[0] = 1; bn_addsub(acc, 6, one, 1, 0); }
  a0[0] = acc[0]; a0[1] = acc[1];
  memset(acc, 0, sizeof acc);
  bn_mul(term, X0, 3, tt, 2); term[5] = 0;
  bn_addsub(acc, 6, term, 5, 0);
  bn_addsub(acc, 6, X1, 3, 1);
  *n1 = (int)(acc[5] >> 63);
  if (*n1) { for (int i = 0; i < 6; i++) acc[i] = ~acc[i]; one[0] = 1; bn_addsub(acc, 6, one, 1, 0); }
  a1[0] = acc[0]; a1[1] = acc[1];
}

void oracle_glv_decompose(const uint8_t* scalar32, uint8_t* out40) {
  glv_init();
  uint64_t s[4], a0[2], a1[2];
  int n0, n1;
  for (int i = 0; i < 4; i++) { uint64_t v = 0; for (int j = 0; j < 8; j++) v |= (uint64_t)scalar32[8 * i + j] << (8 * j); s[i] = v; }
  glv_decompose(s, a0, &n0, a1, &n1);
  for (int i = 0; i < 2; i++) for (int j = 0; j < 8; j++) { out40[8 * i + j] = (uint8_t)(a0[i] >> (8 * j)); out40[16 + 8 * i + j] = (uint8_t)(a1[i] >> (8 * j)); }
  memset(out40 + 32, 0, 8);
  out40[32] = (uint8_t)n0; out40[36] = (uint8_t)n1;
}

static uint32_t bits128(const uint64_t* a, int start, int len) {
  if (start >= 128) return 0;
  int w = start / 64, b = start % 64;
  u128 v = a[w];
  if (w + 1 < 2) v |= (u128)a[w + 1] << 64;
  return (uint32_t)(v >> b) & ((1u << len) - 1);
}

/* default window size: the reference's table (src/msm-common.ts:8-41) */
/* CPUs this process may really use: the cgroup CPU quota (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us) when there is
 * one -- the MI355X boxes show 256 logical CPUs and grant 16 CPUs' worth of time; threads beyond the quota only spin in
 * barriers and burn it.  0 = no quota. */
static int cpu_quota(void) {
  long q = -1, per = 100000;
  FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f) {
    char a[64];
    if (fscanf(f, "%63s %ld", a, &per) == 2 && strcmp(a, "max") != 0) q = atol(a);
    fclose(f);
  } else {
    f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
    if (f) { if (fscanf(f, "%ld", &q) != 1) q = -1; fclose(f); }
    f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
    if (f) { if (fscanf(f, "%ld", &per) != 1) per = 100000; fclose(f); }
  }
  if (q <= 0 || per <= 0) return 0;
  return (int)((q + per - 1) / per);
}
int oracle_cpu_quota(void) { return cpu_quota(); }

int oracle_window_size(int n_log) {
  switch (n_log) { case 14: return 13; case 15: case 16: case 17: case 18: return 14; case 19: case 20: return 18; }
  return n_log - 1 > 1 ? n_log - 1 : 1;
}

/* ---------------------------------------------------------------- the MSM */
/* points: n x 96 B (x || y LE canonical, all-zero = identity); scalars: n x 32 B LE.
 * c <= 0: reference window table.  out: 96 B (x || y), *is_inf.  returns 0, or -1 on bad input. */
int oracle_msm_bls377(const uint8_t* points, const uint8_t* scalars, uint64_t n, int c, uint8_t* out, int* is_inf, int* threads_used) {
  init_consts();
  glv_init();
  if (threads_used) *threads_used = 1;
  memset(out, 0, 96);
  if (n == 0) { *is_inf = 1; return 0; }
  int nlog = 0;
  while ((1ull << nlog) < n) nlog++;
  if (c <= 0) c = oracle_window_size(nlog);
  const int b = 126;
  const int K = (b + 1 + c - 1) / c;
  const uint64_t L = 1ull << (c - 1);
  const uint64_t n2 = 2 * n;

  /* prep 1: Montgomery form, GLV, 2N (point, magnitude) entries with the GLV sign folded into y */
  aff* pts = (aff*)malloc(sizeof(aff) * n2);
  uint64_t* mags = (uint64_t*)malloc(16 * n2);
  if (!pts || !mags) return -1;
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
  /* one thread per physical core: with both SMT siblings of every core spinning in libgomp's barriers the 2^18 case ran
   * 40x slower than on half as many threads (256 logical CPUs, 128 cores on the MI355X host) */
  { int half = omp_get_num_procs() / 2; if (half >= 8 && nthreads > half) nthreads = half; }
  { int quota = cpu_quota(); if (quota >= 1 && nthreads > quota) nthreads = quota; }
  /* small inputs: a parallel region over 128 threads costs more than the work it splits (K windows x 4 regions each) */
  { uint64_t useful = n2 / 2048 + 1; if ((uint64_t)nthreads > useful) nthreads = (int)useful; }
#endif
  if (threads_used) *threads_used = nthreads;
  int bad = 0;
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (int64_t i = 0; i < (int64_t)n; i++) {
    fe x, y;
    fe_from_bytes(&x, points + 96 * i);
    fe_from_bytes(&y, points + 96 * i + 48);
    aff A, Bp;
    A.inf = fe_is_zero(&x) && fe_is_zero(&y);
    if (fe_ge(&x, &P) || fe_ge(&y, &P)) bad = 1;
    fe_mul(&A.x, &x, &R2);
    fe_mul(&A.y, &y, &R2);
    Bp = A;
    fe_mul(&Bp.x, &A.x, &BETA_M);   /* endomorphism, src/wasm/curve.ts:90-103 */
    uint64_t s[4];
    for (int k = 0; k < 4; k++) { uint64_t v = 0; for (int j = 0; j < 8; j++) v |= (uint64_t)scalars[32 * i + 8 * k + j] << (8 * j); s[k] = v; }
    /* reduce mod q (inputs are specified < q) */
    for (int it = 0; it < 16; it++) {
      int ge = 1;
      for (int k = 3; k >= 0; k--) { if (s[k] > Q4[k]) break; if (s[k] < Q4[k]) { ge = 0; break; } }
      if (!ge) break;
      bn_addsub(s, 4, Q4, 4, 1);
    }
    int n0, n1;
    glv_decompose(s, &mags[4 * i], &n0, &mags[4 * i + 2], &n1);
    if (n0) fe_neg(&A.y, &A.y);
    if (n1) fe_neg(&Bp.y, &Bp.y);
    pts[2 * i] = A;
    pts[2 * i + 1] = Bp;
  }
  if (bad) { free(pts); free(mags); return -1; }

  proj* part = (proj*)malloc(sizeof(proj) * K);
  int oom = 0;
  /* Every phase of a window runs on a TEAM of threads -- the reference's SPMD layout: entries split across the team for
   * slicing / counting / sorting (:175-203, :456-502), buckets split across it for the accumulation rounds and the
   * reduction (`computeBucketsSplit`, :626-667, src/msm-common.ts:72-172).  With many cores the windows themselves run side
   * by side, one team each (they are independent until the final sum, :312-333): 128 threads meeting in four barriers per
   * window scaled badly (round 2: 7.6e3 points/s per thread at 2^24 against 5.8e4 on 17 threads). */
  int wpar = 1, team = nthreads, team_seen = 0;
#ifdef _OPENMP
  if (nthreads >= 16) {
    wpar = nthreads / 2 < K ? nthreads / 2 : K;   /* measured under a 16-CPU quota at 2^20: 8 windows x 2 threads 0.97 s, 2 x 8 1.03-1.10 s, 1 x 16 1.23 s */
    { const char* e = getenv("ORACLE_WPAR"); if (e && atoi(e) >= 1) wpar = atoi(e) < K ? atoi(e) : K; }   /* tools/cpu_teams.py */
    team = nthreads / wpar;
    if (team < 1) team = 1;
    omp_set_max_active_levels(2);
  }
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(wpar)
  for (int k = 0; k < K; k++) {
    const int nthreads = team;   /* inside the window: the team */
    uint32_t* dig = (uint32_t*)malloc(4 * n2);
    uint64_t* start = (uint64_t*)malloc((L + 2) * 8);
    uint64_t* cursor = (uint64_t*)malloc((L + 2) * 8);
    aff* sorted = (aff*)malloc(sizeof(aff) * (n2 ? n2 : 1));
    proj* tsum = (proj*)malloc(sizeof(proj) * nthreads);
    if (!dig || !start || !cursor || !sorted || !tsum) {
#pragma omp atomic write
      oom = 1;
      free(dig); free(start); free(cursor); free(sorted); free(tsum);
      continue;
    }
    for (int i = 0; i < nthreads; i++) proj_zero(&tsum[i]);   /* the runtime may grant an inner team fewer threads than asked for */
    /* slice + count (:175-203) and scatter (:456-502).  NOTE: digits need the carry of all lower windows.
     * Each thread counts its own contiguous share of the entries in a PRIVATE histogram and later scatters the same
     * share from private cursors (a shared histogram bounces its cache lines between 256 threads on two sockets);
     * windows whose private histograms would not fit fall back to atomic counters, which is fine there because with
     * that many buckets two threads rarely meet on one. */
    memset(start, 0, (L + 2) * 8);
    const int use_local = (uint64_t)nthreads * (L + 2) * 4 <= (1ull << 30);
    uint32_t* th = use_local ? (uint32_t*)calloc((size_t)nthreads * (L + 2), 4) : NULL;
    const int local = th != NULL;
#pragma omp parallel num_threads(nthreads)
    {
      int tid = 0, nt = 1;
#ifdef _OPENMP
      tid = omp_get_thread_num(); nt = omp_get_num_threads();
#endif
      const uint64_t jlo = n2 / nt * tid + (tid < (int)(n2 % nt) ? tid : n2 % nt);
      const uint64_t jhi = jlo + n2 / nt + (tid < (int)(n2 % nt) ? 1 : 0);
      uint32_t* mine = local ? th + (size_t)tid * (L + 2) : NULL;
      for (uint64_t j = jlo; j < jhi; j++) {
        uint32_t carry = 0, l = 0;
        for (int kk = 0; kk <= k; kk++) {
          l = bits128(&mags[2 * j], kk * c, c) + carry;
          if (l > L) { l = (uint32_t)(2 * L - l); carry = 1; } else carry = 0;
        }
        dig[j] = l | (carry << 31);
        if (l) { if (local) mine[l]++; else __atomic_fetch_add(&start[l + 1], 1, __ATOMIC_RELAXED); }
      }
#pragma omp barrier
      if (local) {   /* bucket totals, bucket by bucket in parallel */
#pragma omp for schedule(static)
        for (int64_t l = 1; l <= (int64_t)L; l++) {
          uint64_t tot = 0;
          for (int t2 = 0; t2 < nt; t2++) tot += th[(size_t)t2 * (L + 2) + l];
          start[l + 1] = tot;
        }
      }
#pragma omp single
      {
        /* integrate (:423-447): start[l] = first slot of bucket l, start[L+1] = total */
        for (uint64_t l = 1; l <= L + 1; l++) start[l] += start[l - 1];
        memcpy(cursor, start, 8 * (L + 2));
      }
      if (local) {   /* private cursors: bucket start + what the threads before this one put there */
#pragma omp for schedule(static)
        for (int64_t l = 1; l <= (int64_t)L; l++) {
          uint32_t run = 0;
          for (int t2 = 0; t2 < nt; t2++) { uint32_t v = th[(size_t)t2 * (L + 2) + l]; th[(size_t)t2 * (L + 2) + l] = run; run += v; }
        }
      }
      for (uint64_t j = jlo; j < jhi; j++) {
        uint32_t l = dig[j] & 0x7fffffffu;
        if (!l) continue;
        aff a = pts[j];
        if (dig[j] >> 31) fe_neg(&a.y, &a.y);
        uint64_t pos = local ? start[l] + mine[l]++ : __atomic_fetch_add(&cursor[l], 1, __ATOMIC_RELAXED);
        sorted[pos] = a;
      }
    }
    free(th);
    const uint64_t total = start[L + 1];
    /* accumulation rounds (:243-282) and bucket reduction (:556-583): buckets [lo, hi] per thread, equal shares of the
     * sorted entries; every thread builds its own pair lists and shares one inversion per round among them */
#pragma omp parallel num_threads(nthreads)
    {
      int tid = 0, nt = 1;
#ifdef _OPENMP
      tid = omp_get_thread_num(); nt = omp_get_num_threads();
      if (tid == 0) {
#pragma omp atomic write
        team_seen = nt;
      }
#endif
      /* bucket range of this thread: boundaries where the cumulative entry count crosses tid / nt of the total */
      uint64_t lo = 1, hi = 0;
      {
        const uint64_t want_lo = total / nt * tid, want_hi = (tid == nt - 1) ? total : total / nt * (tid + 1);
        uint64_t a = 1, b = L + 1;   /* first bucket l with start[l] >= want_lo */
        while (a < b) { uint64_t m = (a + b) / 2; if (start[m] >= want_lo) b = m; else a = m + 1; }
        lo = a;
        a = 1; b = L + 1;
        while (a < b) { uint64_t m = (a + b) / 2; if (start[m] >= want_hi) b = m; else a = m + 1; }
        hi = a - 1;                 /* buckets lo .. hi */
        if (tid == nt - 1) hi = L;
      }
      proj contrib;
      proj_zero(&contrib);
      if (lo <= hi) {
        const uint64_t ebeg = start[lo], eend = start[hi + 1];
        uint64_t maxb = 0;
        for (uint64_t l = lo; l <= hi; l++) { uint64_t sz = start[l + 1] - start[l]; if (sz > maxb) maxb = sz; }
        size_t maxpairs = (eend - ebeg) / 2 + 1;
        aff** Gp = (aff**)malloc(sizeof(aff*) * maxpairs);
        aff** Hp = (aff**)malloc(sizeof(aff*) * maxpairs);
        fe* pre = (fe*)malloc(sizeof(fe) * maxpairs);
        uint8_t* kind = (uint8_t*)malloc(maxpairs);
        if (!Gp || !Hp || !pre || !kind) {
#pragma omp atomic write
          oom = 1;
        } else {
          for (uint64_t m = 1; m < maxb; m *= 2) {
            size_t np = 0;
            for (uint64_t l = lo; l <= hi; l++) {
              uint64_t bs = start[l], be = start[l + 1];
              for (uint64_t q = bs; q + m < be; q += 2 * m) { Gp[np] = &sorted[q]; Hp[np] = &sorted[q + m]; np++; }
            }
            if (np) batch_add(Gp, Hp, np, pre, kind);
          }
          /* reduction of this thread's bucket range: sum_l l * B_l = tri + (lo - 1) * row (:574-580) */
          proj row, tri;
          proj_zero(&row); proj_zero(&tri);
          for (uint64_t l = hi; l >= lo; l--) {
            if (start[l + 1] > start[l] && !sorted[start[l]].inf) {
              proj Bk; Bk.X = sorted[start[l]].x; Bk.Y = sorted[start[l]].y; Bk.Z = ONE;
              proj_add(&row, &row, &Bk);
            }
            proj_add(&tri, &tri, &row);
          }
          uint64_t w = lo - 1;
          contrib = tri;
          while (w) {              /* double-and-add of the row sum */
            if (w & 1) proj_add(&contrib, &contrib, &row);
            w >>= 1;
            if (w) proj_dbl(&row, &row);
          }
        }
        free(Gp); free(Hp); free(pre); free(kind);
      }
      tsum[tid] = contrib;
    }
    proj acc;
    proj_zero(&acc);
    for (int i = 0; i < nthreads; i++) proj_add(&acc, &acc, &tsum[i]);
    part[k] = acc;
    free(tsum);
    free(dig); free(start); free(cursor); free(sorted);
  }
  free(pts); free(mags);
  if (threads_used && team_seen) *threads_used = wpar * team_seen;   /* what the runtime really granted */
  if (oom) { free(part); return -2; }
  /* final sum (:322-333) */
  proj acc = part[K - 1];
  for (int k = K - 2; k >= 0; k--) {
    for (int j = 0; j < c; j++) proj_dbl(&acc, &acc);
    proj_add(&acc, &acc, &part[k]);
  }
  free(part);
  if (proj_is_zero(&acc)) { *is_inf = 1; return 0; }
  *is_inf = 0;
  fe zi, x, y, one = {{1, 0, 0, 0, 0, 0}};
  fe_inv(&zi, &acc.Z);
  fe_mul(&x, &acc.X, &zi); fe_mul(&y, &acc.Y, &zi);
  fe_mul(&x, &x, &one); fe_mul(&y, &y, &one);
  fe_to_bytes(out, &x); fe_to_bytes(out + 48, &y);
  return 0;
}

/* sum_i a_i * s_i over n pairs of 32-byte little-endian integers, as ONE 640-bit integer (10 x 64-bit words, no modular
 * reduction: the caller reduces mod q).  Checker of the known-discrete-log identity sum s_i P_i = (sum s_i a_i) G at sizes
 * Python integers are too slow for (2^26: 6.7e7 products). */
void oracle_dot_u256(const uint8_t* a, const uint8_t* s, uint64_t n, uint64_t* out10) {
  uint64_t total[10] = {0};
#pragma omp parallel
  {
    uint64_t acc[10] = {0};
#pragma omp for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
      uint64_t x[4], y[4];
      memcpy(x, a + 32 * i, 32);
      memcpy(y, s + 32 * i, 32);
      uint64_t prod[8] = {0};
      for (int u = 0; u < 4; u++) {
        u128 c = 0;
        for (int v = 0; v < 4; v++) { c += (u128)x[u] * y[v] + prod[u + v]; prod[u + v] = (uint64_t)c; c >>= 64; }
        prod[u + 4] = (uint64_t)c;
      }
      u128 c = 0;
      for (int j = 0; j < 10; j++) { c += (u128)acc[j] + (j < 8 ? prod[j] : 0); acc[j] = (uint64_t)c; c >>= 64; }
    }
#pragma omp critical
    {
      u128 c = 0;
      for (int j = 0; j < 10; j++) { c += (u128)total[j] + acc[j]; total[j] = (uint64_t)c; c >>= 64; }
    }
  }
  memcpy(out10, total, sizeof total);
}

/* field operator for KATs: op 0 = a*b mod p, 1 = a^-1 mod p, 2 = a+b, 3 = a-b (plain canonical integers) */
void oracle_fp_op(int op, const uint8_t* a48, const uint8_t* b48, uint8_t* out48) {
  init_consts();
  fe a, b, r, one = {{1, 0, 0, 0, 0, 0}};
  fe_from_bytes(&a, a48); fe_from_bytes(&b, b48);
  fe_mul(&a, &a, &R2); fe_mul(&b, &b, &R2);
  if (op == 0) fe_mul(&r, &a, &b);
  else if (op == 1) fe_inv(&r, &a);
  else if (op == 2) fe_add(&r, &a, &b);
  else fe_sub(&r, &a, &b);
  fe_mul(&r, &r, &one);
  fe_to_bytes(out48, &r);
}
