/* C ABI of the MI355X MSM engine (libmsm_hip.so).
 *
 * This is the drop-in boundary for the reference's MSM path: the coarse `Curve.Parallel.*` surface
 * that callers use (reference src/parallel.ts:135-145, 251-259) expressed over plain pointers and
 * sizes, plus a few fine-grained operators of the wasm export table (src/field-msm.ts:86-123,
 * src/scalar-glv.ts:41-51) exposed as GPU test kernels.  INTEGRATION.md shows the N-API / ctypes
 * stubs that bind it.
 *
 * Wire formats are the reference's (`pointsFromBytes` / `scalarsFromBytes`, src/parallel.ts:97-133):
 *   BLS12-377 G1 point : 96 bytes  = x || y, each 48-byte little-endian canonical integer (non-Montgomery)
 *   Ed-on-BLS12-377    : 64 bytes  = x || y, each 32-byte little-endian
 *   scalar             : 32 bytes little-endian
 * The all-zero point encoding denotes the identity (the reference's byte format cannot express it;
 * its object form has `isZero`, scripts/zprize23/submission-bls377.ts:83).
 *
 * Ownership: the caller owns every buffer it passes; the library owns all device memory.  Calls on
 * one context are serialised by the caller; distinct contexts are independent.  Every function
 * returns MSM_OK or an error code; msm_last_error() gives the message of the last failure.
 */
#ifndef MSM_HIP_H
#define MSM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  MSM_OK = 0,
  MSM_ERR_ARG = 1,         /* null pointer, bad length, bad window size, unknown curve */
  MSM_ERR_HIP = 2,         /* HIP runtime failure (message has the HIP error string) */
  MSM_ERR_POINT = 3,       /* coordinate >= p, or point not on the curve (when validation is requested) */
  MSM_ERR_NO_POINTS = 4,   /* msm called before msm_set_points, or with n > resident points */
  MSM_ERR_NO_DEVICE = 5,   /* no usable GPU: there is no CPU fallback */
  MSM_ERR_SCALAR = 6,      /* a scalar >= the group order q and msm_opts.strict was set (default: reduced mod q) */
  MSM_ERR_INTERNAL = 7     /* host allocation failure or another unexpected condition; no C++ exception crosses this ABI */
};

enum {
  MSM_CURVE_BLS12_377_G1 = 0,     /* Weierstrass + GLV, batched-affine path: src/msm-batched-affine.ts */
  MSM_CURVE_ED_ON_BLS12_377 = 1,  /* twisted Edwards, generic path: src/msm-basic.ts */
  MSM_CURVE_BLS12_381_G1 = 2,     /* Weierstrass + GLV, batched-affine path; src/concrete/bls12-381.params.ts */
  MSM_CURVE_PALLAS = 3            /* same path on 9 limbs / 8 packed words, src/concrete/pasta.params.ts (the reference sizes
                                   * limbs per field, src/parallel.ts:53-57): coordinates at this ABI -- wire points, test
                                   * operands, the used part of msm_result.x / .y -- are 32-byte little-endian integers, as for
                                   * the Edwards curve; window sums stay 144-byte (X, Y, Z) records for every curve */
};

/* Layout version of this header.  The symbol names do not change when a struct grows or an argument changes meaning, so a
 * binding built against another version of the header would misread memory silently: msm_abi_version() returns the version the
 * LIBRARY was built with, msm_abi_struct_bytes(0 / 1) its sizeof(msm_opts) / sizeof(msm_result).  The Python loader and the N-API
 * addon compare both at load time and refuse a mismatch.  History: 3 = round 3 (msm_generate_scalars writes to a caller-owned
 * buffer; msm_opts.point_lo / by_window); 4 = msm_result.n_pairs_algo; 5 = window tables (msm_opts.no_tables, msm_result.tables,
 * msm_precompute / msm_tables_info / msm_set_tables_limit), msm_reserve, msm_opts.bucket_shard / bucket_shards; 6 = window tables
 * over a range of the points (msm_opts.merged_sums, msm_precompute with point_lo, msm_tables_range). */
#define MSM_ABI_VERSION 6
uint32_t msm_abi_version(void);
uint32_t msm_abi_struct_bytes(int which);

typedef struct msm_ctx msm_ctx;

/* Options of one msm call; mirrors `{c, useSafeAdditions}` of src/msm-batched-affine.ts:74-77.
 * Zero-initialise for defaults. */
typedef struct msm_opts {
  int32_t c;            /* window size in bits, 0 = pick from N (windowSize, src/msm-common.ts:8-41, retuned).  The number of
                           windows is the reference's K = ceil((b + 1) / c) (src/msm-batched-affine.ts:90) with one exception:
                           for c >= 18, a top window that would hold the carry bit of the signed recoding alone
                           (b + 1 = (K - 1) c + 1; BLS12-377: c = 18, 21) is folded into the window below -- K - 1 windows,
                           window k still weighs 2^(c k).  msm_plan / msm_result.K report the K in use; every rank of a
                           sharded run gets the same one from the same c */
  int32_t unsafe;       /* accepted for API parity with msmUnsafe (src/curve-affine.ts:463-522) and ignored: there is one tree
                           kernel and it always handles the edge cases -- the equal-x detection the unsafe variant would drop is
                           19 of its 3 171 instructions per pair addition (0.6 %), and the identity operands it also ignores are
                           structural here (bucket padding) */
  int32_t k_lo, k_hi;   /* window shard [k_lo, k_hi) for msm_window_sums; 0,0 = all windows */
  int32_t serial;       /* != 0: run the window groups one after the other on one stream (no overlap): phase_ms then
                           hold exclusive kernel times -- used for roofline measurements.  Every tree launch then has the
                           chip to itself and takes the launch geometry of a lone window (128 instead of 512 pairs per
                           lane and batch), which is what such a launch would ship with */
  int32_t no_glv;       /* != 0 (Weierstrass curves): no endomorphism split -- digits of the full scalar, K = ceil((b + 1) / c)
                           with b = bit length of q: the window structure of msmProjective / msmBasic
                           (src/parallel.ts:69-87, src/msm-basic.ts:56-91).  Same group element, 2x the additions */
  int32_t strict;       /* != 0: a scalar >= q fails the call with MSM_ERR_SCALAR.  Default: such scalars are reduced mod q
                           (the reference specifies inputs < q, src/curve-random.ts:151-194, and does not check) */
  uint32_t point_lo;    /* the call covers the resident points [point_lo, point_lo + n): scalars[i] belongs to point
                           point_lo + i.  The points-split shard of a multi-GPU run: rank g runs all K windows
                           on its n / G points, msm_combine_groups adds the G partial sums of every window */
  int32_t by_window;    /* multi-device contexts (msm_ctx_create_multi): != 0 shards an MSM by scalar window, every device then
                           needs all n scalars; default 0: by points, device d gets the scalars of its n / G points only */
  int32_t no_tables;    /* != 0: do not use (and do not build) window tables for this call -- the plain path over the resident
                           rows, K windows of buckets and a Horner step, as in rounds 1-4 (see msm_precompute) */
  int32_t bucket_shard, bucket_shards;   /* bucket_shards = G > 1: the call covers only the buckets [L g / G, L (g + 1) / G) of every
                           window, g = bucket_shard (L = buckets per window): the bucket-range shard of a multi-GPU run -- rank g
                           slices ALL scalars into all K windows but sorts and adds an eighth of the entries, K stays the
                           single-GPU plan's (the reference splits every window's buckets across its threads the same way,
                           src/msm-common.ts:72-172).  The partial sums keep the buckets' true weights: the G results of
                           msm_window_sums add up per window (msm_combine_groups), those of msm_run as points */
  int32_t merged_sums;  /* msm_window_sums, != 0: the caller only COMBINES the sums (msm_combine / msm_combine_groups), so the call may
                           hand them back merged -- slot 0 of its window range then carries sum_k 2^(c (k - k_lo)) P_k and the
                           other slots the identity, which the Horner step of either combine takes like one P_k per slot -- and
                           with that run on window tables: those of the whole point set, or of the range of the points the call
                           covers (the share of one rank of a points-split run; see msm_precompute).  Default 0: one P_k per
                           slot, the plain path */
  int32_t reserved_;
} msm_opts;

#define MSM_N_PHASES 8
/* phase_ms indices (HIP-event timings on the context's stream; the reference returns a tic/toc log,
 * src/msm-common.ts:176-214) */
enum {
  MSM_T_TOTAL = 0, MSM_T_UPLOAD = 1, MSM_T_DIGITS = 2, MSM_T_SORT = 3,
  MSM_T_ACCUMULATE = 4, MSM_T_REDUCE = 5, MSM_T_FINAL = 6, MSM_T_ACC_ROUND1 = 7
};

typedef struct msm_result {
  uint8_t x[48];        /* canonical affine result, little-endian (first 32 bytes used for Ed-on-BLS12-377) */
  uint8_t y[48];
  int32_t is_infinity;  /* Weierstrass only; twisted Edwards returns (0, 1) for the identity */
  int32_t c;            /* window size used */
  int32_t K;            /* number of windows */
  int32_t rounds;       /* accumulation tree rounds (k_batch_add launches) summed over all window groups */
  float phase_ms[MSM_N_PHASES];
  uint64_t n_pairs;     /* affine pair additions issued (all rounds, all windows), padding lanes of the tree included */
  uint64_t max_bucket;  /* largest bucket population seen */
  uint64_t n_pairs_algo; /* pair additions the bucket sums NEED: sum over the non-empty buckets of (population - 1); the
                            basis of roofline figures (n_pairs is ~2.5 % above it at 2^26) */
  int32_t tables;        /* != 0: the call ran on window tables (K tables of the point set, one set of buckets per window group) */
  int32_t reserved_;
} msm_result;

/* Context: binds one curve to one GPU (device index as seen by HIP). Replaces
 * `Weierstraß.create(params)` / `TwistedEdwards.create(params)` (src/parallel.ts:40-66, 179-200). */
int msm_ctx_create(msm_ctx** out, int curve, int device);
/* The same over a device list ("create(curve id, device list)").  Every device holds the whole
 * point set; msm_run / msm_window_sums give every device a share of the POINTS (all windows over n / G points, the
 * default) or, with msm_opts.by_window, one contiguous shard of the window range (windows are independent until the
 * final sum, src/msm-batched-affine.ts:312-333), run the shards from one host thread per device and combine the
 * K x 144 bytes of partition sums per device on the host -- so a C or JS host can use a whole node without
 * torch.distributed.  Device scalars must live on devices[0]; the other devices copy their part peer-to-peer.
 * (bench.py --gpus N keeps the one-process-per-GPU RCCL form of the north star.) */
int msm_ctx_create_multi(msm_ctx** out, int curve, const int32_t* devices, int32_t n_devices);
int msm_ctx_device_count(const msm_ctx* ctx);
void msm_ctx_destroy(msm_ctx* ctx);
const char* msm_last_error(const msm_ctx* ctx);

/* Upload + convert the base points and keep them resident (pointsFromBytes, src/parallel.ts:97-116:
 * fromPackedBytes + toMontgomery; the endomorphism image beta*x is precomputed here once instead of
 * per call as in preparePointsAndScalars, src/msm-batched-affine.ts:350-421).
 * on_device != 0: `points` is a device pointer.  check_curve != 0: verify the curve equation. */
int msm_set_points(msm_ctx* ctx, const void* points, uint64_t n, int on_device, int check_curve);

/* Window tables.  For a fixed point set (the bases of a prover: the reference's callers load their points once and run many
 * MSMs over them, scripts/msm-weierstrass.ts:19-35) the library can keep K tables instead of one: table k holds 2^(c k) P_i for
 * every point, so the digit of window k addresses a point that already carries the window's weight and all K windows share ONE
 * set of 2^(c-1) buckets -- K times fewer buckets to finish and reduce, no Horner step, same group element.  The reference has
 * no counterpart (its heap is 4 GiB); six tables of 2^26 BLS12-377 points are 103 GB of the 288 GB of HBM.
 * msm_run builds them by itself on its first call over the WHOLE current point set with the default window (opts->c == 0)
 * when they fit the limit (default: 10 % of the device memory -- 28 GB: point sets of up to 2^24 BLS12-377 points; what they
 * buy shrinks from 6-10 % below 2^24 points to 1.5 % at 2^26, where they would take 96 GB), and uses them whenever the call's plan is the one they were built
 * for; any other call -- another window size, a prefix or another range of the points, msm_window_sums without merged_sums, a bucket-range shard, a device-list context --
 * takes the plain path over table 0, which is always the plain row table.  msm_precompute builds them ahead of the first call
 * (also for an explicit opts->c); it is not an error if they do not fit: the plain path stays.  msm_set_points drops them.
 * Tables over a RANGE of the points (round 6): the rank of a points-split run works on its share [point_lo, point_lo + n) of
 * the resident points in every step; msm_precompute with opts->point_lo builds the tables of exactly that range (2^23 points x 7
 * tables = 15 GB, in a buffer of their own next to the plain rows), and msm_run / msm_window_sums (with msm_opts.merged_sums)
 * over that range run on them.  Without msm_precompute they are built when a call comes back for the same range a second
 * time in a row -- a caller that walks over several ranges on one GPU is spared a build per call.  A point set holds the tables
 * of ONE range (or of the whole set, which a range never replaces by itself).
 * msm_tables_info: window size and number of tables present (0, 0: none) and their bytes; msm_tables_range: the points they cover. */
int msm_precompute(msm_ctx* ctx, uint64_t n, const msm_opts* opts);
int msm_tables_info(const msm_ctx* ctx, int32_t* c_out, int32_t* K_out, uint64_t* bytes_out);
int msm_tables_range(const msm_ctx* ctx, uint64_t* point_lo_out, uint64_t* n_out);
int msm_set_tables_limit(msm_ctx* ctx, uint64_t bytes);   /* 0: never build tables */

/* Everything a later msm_run(ctx, <device scalars>, n, opts) allocates or builds -- the per-call workspace (device memory costs
 * ~40 ms per GB to get on this system: 3 s before the first 2^26 MSM) and the window tables -- taken out of that call: runs one
 * MSM over internally generated scalars and discards the result. */
int msm_reserve(msm_ctx* ctx, uint64_t n, const msm_opts* opts);

/* Point-set handles (the reference's `pointPtr`s are independent allocations, src/parallel.ts:97-116): a context starts
 * with point set 0; msm_pointset_create adds an empty one and makes it current, msm_pointset_select switches.
 * msm_set_points / msm_generate_points / msm_run / msm_get_points always act on the current set. */
int msm_pointset_create(msm_ctx* ctx, int32_t* id_out);
int msm_pointset_select(msm_ctx* ctx, int32_t id);
int msm_pointset_destroy(msm_ctx* ctx, int32_t id);

/* Device buffers for scalar handles (`scalarPtr`s are independent allocations too): owned by the context, freed by
 * msm_device_free or with the context.  On a multi-device context they live on devices[0]. */
int msm_device_alloc(msm_ctx* ctx, uint64_t bytes, void** dev_ptr_out);
int msm_device_free(msm_ctx* ctx, void* dev_ptr);
int msm_device_upload(msm_ctx* ctx, void* dev_ptr, const void* host, uint64_t bytes);

/* Device memory the working buffers of one call may take (digits, sort records, tree nodes: the reference sizes them per call
 * in wasm memory, src/msm-batched-affine.ts:96-130).  0 = automatic: 85 % of what the device has free when a big call starts.
 * Windows run in as many groups as fit, and a window whose buffers would not fit at all runs over ranges of the points, one
 * range after the other, its sums added on the host -- so a limit (a GPU shared with other work) or an input of 2^29 points
 * costs time, not an error.  Multi-device contexts apply the limit per device. */
int msm_set_workspace_limit(msm_ctx* ctx, uint64_t bytes);

/* sum_i scalars[i] * points[i] over the first n resident points
 * (msm / msmUnsafe, src/msm-batched-affine.ts:69-340, 587-598; for the Edwards curve msmBasic,
 * src/msm-basic.ts:45-164).  on_device != 0: `scalars` already sits in HBM. */
int msm_run(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const msm_opts* opts, msm_result* out);

/* Window-sharded form for multi-GPU runs: computes the partition sums P_k for k in [k_lo, k_hi)
 * only (src/msm-batched-affine.ts:42 "P_k = sum_l l * B_(k,l)") and writes them as
 * (k_hi - k_lo) x 144 bytes: X || Y || Z homogeneous projective, 48-byte little-endian canonical
 * integers (twisted Edwards: the extended point without T, which msm_combine rebuilds from T Z = X Y).
 * Ranks exchange these with one all-gather; msm_combine finishes.  With msm_opts.merged_sums the slots may come back merged
 * (first slot: sum_k 2^(c (k - k_lo)) P_k, the others the identity), which either combine takes unchanged, and the call may run
 * on window tables (msm_result.tables says whether it did). */
int msm_window_sums(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const msm_opts* opts,
                    uint8_t* partials_out, msm_result* stats);

/* msm_run on a multi-device context with the scalars already PLACED: dev_scalars[d] is a device pointer on devices[d]
 * holding the n_d x 32 bytes of that device's points [n d / G, n (d + 1) / G) (points split), so no peer copy happens
 * inside the call.  On a single-device context dev_scalars[0] is the whole scalar array. */
int msm_run_placed(msm_ctx* ctx, const void* const* dev_scalars, uint64_t n, const msm_opts* opts, msm_result* out);

/* S = sum_k 2^(c k) P_k over all K windows, then to affine (src/msm-batched-affine.ts:322-333,
 * src/curve-projective.ts:335-349).  Host arithmetic only: `ctx` may be NULL. */
int msm_combine(msm_ctx* ctx, const uint8_t* partials, int32_t K, int32_t c, msm_result* out);
/* The same without a context: `curve` names the constants.  No GPU is touched. */
int msm_combine_curve(int curve, const uint8_t* partials, int32_t K, int32_t c, msm_result* out);
/* Points-split runs: `partials` holds G groups of K window sums (group g = the sums of the g-th share of the points, as
 * msm_window_sums wrote them); P_k = sum over the groups, then as msm_combine_curve. */
int msm_combine_groups(int curve, const uint8_t* partials, int32_t G, int32_t K, int32_t c, msm_result* out);

/* Window plan for n points: the c the library would pick (opts->c forces one) and the resulting K (see msm_opts.c).  It is the
 * plan of msm_run over DEVICE-RESIDENT scalars: over the whole current point set that is the plan on window tables where they
 * exist or would be built (opts->no_tables: the plain plan, which msm_window_sums without merged_sums and bucket-range shards run;
 * opts->merged_sums: the plan of msm_window_sums(merged_sums) over the range [point_lo, point_lo + n), on its tables where they fit).
 * Host scalars of 2^24 points and more cross PCIe behind the computation: msm_run then runs whole MSMs over growing ranges of
 * the points, each under the plan of ITS size, and msm_result reports the plan of the last, biggest range. */
int msm_plan(const msm_ctx* ctx, uint64_t n, const msm_opts* opts, int32_t* c_out, int32_t* K_out);

/* Synthetic inputs generated on the GPU (randomPointsFast / randomScalars, src/curve-random.ts):
 * n resident points P_i = a_i * G and, if `a_out` is non-null, the n scalars a_i (32-byte LE) so a
 * caller can verify sum s_i P_i = (sum s_i a_i) G in O(n). */
int msm_generate_points(msm_ctx* ctx, uint64_t n, uint64_t seed, uint8_t* a_out);
/* n uniformly random scalars < q (`randomScalars`, src/curve-random.ts:151-194) into `dev_dst`, a caller-owned device buffer
 * of n * 32 bytes (msm_device_alloc, or any device allocation of the caller), and / or into `host_out` (n * 32 bytes).
 * Either may be NULL, not both. */
int msm_generate_scalars(msm_ctx* ctx, uint64_t n, uint64_t seed, void* dev_dst, uint8_t* host_out);

/* Read resident points [first, first + count) back in wire format (tests, CPU-baseline sampling). */
int msm_get_points(msm_ctx* ctx, uint64_t first, uint64_t count, uint8_t* out_xy);

/* ---- fine-grained GPU operators for parity tests (debug surface) ----
 * Coordinates are the curve's ABI width: 48-byte little-endian integers for BLS12-377 / BLS12-381, 32-byte ones for Pallas and
 * Ed-on-BLS12-377 ("48-byte" / "96-byte" below stand for one / two coordinates of that width). */
enum { MSM_OP_MUL = 0, MSM_OP_SQR = 1, MSM_OP_ADD = 2, MSM_OP_SUB = 3, MSM_OP_INV = 4,
       MSM_OP_TO_MONT = 5, MSM_OP_FROM_MONT = 6,
       MSM_OP_INV_FERMAT = 7,   /* a^(p-2): the cross-check of MSM_OP_INV (division steps) */
       MSM_OP_INV_KALISKI = 8,  /* the reference's almost-inverse, src/wasm/inverse.ts:136-218 */
       MSM_OP_INV_WORDSLICED = 9 /* the reference's experimental word-sliced almost-inverse, src/inverse/faster-inverse-wasm.ts:133-343 */ };
/* element-wise base-field op on n operands, each a 48-byte (32 for Ed) little-endian word string;
 * MUL/SQR/ADD/SUB/INV act on Montgomery-form operands (radix 2^390 / 2^270) and return canonical
 * Montgomery-form values, i.e. `multiply`, `square`, `add`, `subtract`, `inverse` of
 * src/field-msm.ts:86-123. */
int msm_test_fp(msm_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, uint64_t n);
/* `batchInverse` (src/wasm/inverse.ts:220-271): n Montgomery-form elements (48-byte words, non-zero) inverted with
 * one field inversion per `per_lane` consecutive elements (Montgomery's trick). */
int msm_test_batch_inverse(msm_ctx* ctx, const uint8_t* xs, uint8_t* out, uint64_t n, uint32_t per_lane);
/* GLV `decompose` (src/wasm/glv.ts:68-169) of n 32-byte scalars: out = n x 40 bytes
 * |s0| (16 B LE) || |s1| (16 B LE) || neg0 (u32) || neg1 (u32). */
int msm_test_glv(msm_ctx* ctx, const uint8_t* scalars, uint8_t* out, uint64_t n);
/* affine pair additions G_i + H_i through the batched-affine kernel (batchAddNew,
 * src/curve-affine.ts:376-458): inputs n x 96-byte wire points, output n x 96 bytes.  Ed-on-BLS12-377: the
 * unified extended addition of the gather round (src/curve-twisted-edwards.ts:84-165), n x 64 bytes each way. */
int msm_test_batch_add(msm_ctx* ctx, const uint8_t* g, const uint8_t* h, uint8_t* out, uint64_t n);

/* ---- worst-case / operator-level surface (round 2) ---- */
/* fe_mul / fe_sqr on RAW register-form operands: n x NL 30-bit limbs as uint32 (NL = 13; Ed-on-BLS12-377: 9), exactly as
 * given -- unreduced values up to the documented 2^6 p, all-ones limbs -- and the result limbs as they leave the
 * multiplier (not reduced).  op = MSM_OP_MUL or MSM_OP_SQR.  Mirrors src/field.test.ts:27-155 on [0, 2p) and beyond. */
int msm_test_fp_raw(msm_ctx* ctx, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, uint64_t n);
/* curve operators, one per element: Weierstrass curves take and return homogeneous projective points n x (X || Y || Z),
 * 48-byte little-endian integers < p (any representative; Z = 0 is the identity); Ed-on-BLS12-377 extended points
 * n x (X || Y || Z || T) of 32 bytes.  op 0: general addition with every edge case (proj_add / te_add, 9M form),
 * 1: doubling of P, 2 (Weierstrass): mixed addition, Q affine with (0, 0) the identity.
 * src/curve-projective.test.ts:77-208, src/curve-twisted-edwards.test.ts:55-158. */
enum { MSM_CURVE_OP_ADD = 0, MSM_CURVE_OP_DOUBLE = 1, MSM_CURVE_OP_ADD_MIXED = 2 };
int msm_test_curve_op(msm_ctx* ctx, int op, const uint8_t* p, const uint8_t* q, uint8_t* out, uint64_t n);
/* msm_test_batch_add through the plane-reading modes of the tree kernel with a chosen number of pairs per lane (= per
 * shared inversion): mode 1 = regular rounds, 2 = tail rounds (operand descriptors; an all-zero H_e is passed as
 * "no second operand").  steps >= 1. */
int msm_test_batch_add_mode(msm_ctx* ctx, const uint8_t* g, const uint8_t* h, uint8_t* out, uint64_t n, int mode, uint32_t steps);
/* Bucket reduction alone (Weierstrass curves): buckets = K x L affine points (x || y, 48-byte LE; (0, 0) = empty bucket),
 * bucket l of window k at index k L + l - 1; L a power of two.  Writes P_k = sum_l l B_(k,l) as K x 144 bytes (X || Y || Z)
 * and the device time of the reduction in *ms_out.
 *   mode 0: as the MSM does it -- projective row / triangle sums per chunk of buckets and bit-sliced weights
 *           (reduceBucketsColumnProjective, src/msm-batched-affine.ts:556-583);
 *   mode 1: the all-affine reduction of the reference's single-thread MSM (reduceBucketsAffine,
 *           src/msm-batched-affine-single-thread.ts:522-667, doc/zprize22.md:317-358) out of in-place batched-affine
 *           additions and doublings; 2^c0 = buckets per chunk of its linear part.  SURVEY section 8(f)-3. */
int msm_test_bucket_reduce(msm_ctx* ctx, const uint8_t* buckets, int32_t K, uint32_t L, int mode, int c0, uint8_t* partials_out,
                           float* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* MSM_HIP_H */
