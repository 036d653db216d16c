// Node N-API shim over the C ABI of libmsm_hip.so (include/msm_hip.h).
//
// This is the "thin Node N-API C-ABI shim into HIP" that replaces the reference's wasm backend for
// the MSM path: the JS facade in js/montgomery-hip.js rebuilds the reference's
// `Curve.Parallel.{pointsFromBytes, scalarsFromBytes, msm, msmUnsafe}` (src/parallel.ts:135-145) and
// `compute_msm` (scripts/zprize23/submission-bls377.ts:20-65) on top of the functions exported here.
// Plain N-API (C), version 8 as shipped with the image's node 12; no node-addon-api / node-gyp.
//
// Exports: createContext(curve, device | [devices]) -> external handle, destroyContext(h), setPoints(h, Buffer, pointBytes, check),
//          pointsetCreate(h) -> id, pointsetSelect(h, id), pointsetDestroy(h, id),
//          msm(h, Buffer scalars, c, coordBytes, noGlv, unsafe) -> {x: Buffer, y: Buffer, isZero, c, K, phaseMs: number[8], nPairs},
//          deviceAlloc(h, bytes) -> device buffer handle, deviceUpload(h, dbuf, Buffer), deviceFree(h, dbuf),
//          msmDevice(h, dbuf, n, c, noGlv, unsafe) -> as msm: the scalars already sit in HBM (the reference keeps them in the
//          memory its kernels compute in, src/parallel.ts:119-133, scripts/msm-weierstrass.ts:29-32),
//          plan(h, n, c) -> {c, K}, generatePoints(h, n, seed) -> n, generateScalars(h, n, seed[, dbuf]) -> Buffer | n
//          the fine operator table of the reference's wasm exports (src/field-msm.ts:86-123,190-243, src/scalar-glv.ts:41-51,105-128)
//          over Buffers instead of wasm pointers: fieldOp(h, op, a, b) -> Buffer (msm_test_fp: multiply / square / add / subtract /
//          inverse / toMontgomery / fromMontgomery on n = a.length / coordBytes elements), batchInverse(h, xs, perLane) -> Buffer,
//          glvDecompose(h, scalars) -> Buffer of n x 40 bytes (|s0|, |s1|: 16 bytes each, neg0, neg1: 4 bytes each),
//          batchAdd(h, G, H) -> Buffer of n affine sums (wire points, (0, 0) = identity)
// The addon is built against include/msm_hip.h and checks at load that the library it found was too (msm_abi_version).
#include <node_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "msm_hip.h"

#define NAPI_OK(call)                                                      \
  do {                                                                     \
    if ((call) != napi_ok) {                                               \
      napi_throw_error(env, NULL, "N-API call failed: " #call);            \
      return NULL;                                                         \
    }                                                                      \
  } while (0)

static napi_value throw_msm(napi_env env, msm_ctx* ctx, int rc, const char* what) {
  char buf[640];
  snprintf(buf, sizeof buf, "%s: msm error %d: %s", what, rc, ctx ? msm_last_error(ctx) : "no context (no usable GPU? there is no CPU fallback)");
  napi_throw_error(env, NULL, buf);
  return NULL;
}

// What a JS context handle points at: the sizes of the curve's wire format are derived here, never taken from JS
// (a wrong size from the caller would make the library read past the Buffer or this shim read past res.x).
typedef struct {
  msm_ctx* ctx;      /* NULL after destroyContext: a second destroy or any later call throws instead of double-freeing */
  int32_t curve;
  size_t coord_bytes, point_bytes;
  uint32_t refs;     /* the JS handle + every live device-buffer handle: the struct outlives whichever is collected last */
} js_ctx;

/* a device buffer of the context (msm_device_alloc): one per scalar pointer of the JS facade */
typedef struct {
  js_ctx* owner;
  void* dev;         /* NULL after deviceFree */
  uint64_t bytes;
} js_dbuf;

static void ctx_unref(js_ctx* h) {
  if (--h->refs == 0) free(h);
}

static js_ctx* get_handle(napi_env env, napi_value v) {
  void* p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
    napi_throw_type_error(env, NULL, "expected a context handle from createContext()");
    return NULL;
  }
  js_ctx* h = (js_ctx*)p;
  if (!h->ctx) {
    napi_throw_error(env, NULL, "this context has been destroyed");
    return NULL;
  }
  return h;
}
static msm_ctx* get_ctx(napi_env env, napi_value v) {
  js_ctx* h = get_handle(env, v);
  return h ? h->ctx : NULL;
}
static void finalize_handle(napi_env env, void* data, void* hint) {
  (void)env; (void)hint;
  js_ctx* h = (js_ctx*)data;
  if (h->ctx) msm_ctx_destroy(h->ctx);   /* frees every device buffer the context still holds */
  h->ctx = NULL;
  ctx_unref(h);
}
/* a collected scalar pointer gives its device memory back (if its context is still alive and nobody freed it by hand) */
static void finalize_dbuf(napi_env env, void* data, void* hint) {
  (void)hint;
  js_dbuf* b = (js_dbuf*)data;
  if (b->dev && b->owner->ctx) {
    msm_device_free(b->owner->ctx, b->dev);
    int64_t adj;
    napi_adjust_external_memory(env, -(int64_t)b->bytes, &adj);
  }
  ctx_unref(b->owner);
  free(b);
}
static js_dbuf* get_dbuf(napi_env env, js_ctx* h, napi_value v) {
  void* p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
    napi_throw_type_error(env, NULL, "expected a device buffer from deviceAlloc()");
    return NULL;
  }
  js_dbuf* b = (js_dbuf*)p;
  if (b->owner != h || !b->dev) {
    napi_throw_error(env, NULL, b->owner != h ? "this device buffer belongs to another context" : "this device buffer has been freed");
    return NULL;
  }
  return b;
}

static napi_value CreateContext(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  int32_t curve = 0, device = 0;
  if (argc > 0) napi_get_value_int32(env, argv[0], &curve);
  msm_ctx* ctx = NULL;
  int rc;
  bool is_array = false;
  if (argc > 1) napi_is_array(env, argv[1], &is_array);
  if (is_array) {   // a device list: one context over several GPUs of the node (msm_ctx_create_multi)
    uint32_t nd = 0;
    NAPI_OK(napi_get_array_length(env, argv[1], &nd));
    if (nd == 0 || nd > 64) {
      napi_throw_range_error(env, NULL, "device list must hold 1..64 device indices");
      return NULL;
    }
    int32_t devs[64];
    for (uint32_t i = 0; i < nd; i++) {
      napi_value e;
      NAPI_OK(napi_get_element(env, argv[1], i, &e));
      NAPI_OK(napi_get_value_int32(env, e, &devs[i]));
    }
    rc = msm_ctx_create_multi(&ctx, curve, devs, (int32_t)nd);
  } else {
    if (argc > 1) napi_get_value_int32(env, argv[1], &device);
    rc = msm_ctx_create(&ctx, curve, device);
  }
  if (rc != MSM_OK) return throw_msm(env, NULL, rc, "createContext");
  js_ctx* h = (js_ctx*)calloc(1, sizeof(js_ctx));
  if (!h) {
    msm_ctx_destroy(ctx);
    napi_throw_error(env, NULL, "out of memory");
    return NULL;
  }
  h->ctx = ctx;
  h->refs = 1;
  h->curve = curve;
  h->coord_bytes = (curve == MSM_CURVE_ED_ON_BLS12_377 || curve == MSM_CURVE_PALLAS) ? 32 : 48;   /* per field */
  h->point_bytes = 2 * h->coord_bytes;
  napi_value out;
  NAPI_OK(napi_create_external(env, h, finalize_handle, NULL, &out));
  return out;
}

static napi_value DestroyContext(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  js_ctx* h = get_handle(env, argv[0]);   // throws on a handle that was destroyed before
  if (h) {
    msm_ctx_destroy(h->ctx);
    h->ctx = NULL;
  }
  return NULL;
}

static napi_value SetPoints(napi_env env, napi_callback_info info) {  // pointsFromBytes, src/parallel.ts:97-116
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  js_ctx* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  msm_ctx* ctx = h->ctx;
  void* data;
  size_t len;
  NAPI_OK(napi_get_buffer_info(env, argv[1], &data, &len));
  int32_t point_bytes = (int32_t)h->point_bytes, check = 0;
  if (argc > 2) napi_get_value_int32(env, argv[2], &point_bytes);
  if (argc > 3) napi_get_value_int32(env, argv[3], &check);
  if ((size_t)point_bytes != h->point_bytes || len % h->point_bytes) {   // the C ABI reads n x point_bytes of this curve
    napi_throw_range_error(env, NULL, "point buffer: expected a multiple of the curve's point size (96 bytes; Ed-on-BLS12-377 and Pallas: 64)");
    return NULL;
  }
  int rc = msm_set_points(ctx, data, len / point_bytes, 0, check);
  if (rc != MSM_OK) return throw_msm(env, ctx, rc, "setPoints");
  napi_value n;
  NAPI_OK(napi_create_uint32(env, (uint32_t)(len / point_bytes), &n));
  return n;
}

static napi_value result_object(napi_env env, const js_ctx* h, const msm_result* res) {
  const size_t coord = h->coord_bytes;   // of the context's curve, never taken from JS
  napi_value out, x, y, v, ph;
  NAPI_OK(napi_create_object(env, &out));
  NAPI_OK(napi_create_buffer_copy(env, coord, res->x, NULL, &x));
  NAPI_OK(napi_create_buffer_copy(env, coord, res->y, NULL, &y));
  NAPI_OK(napi_set_named_property(env, out, "x", x));
  NAPI_OK(napi_set_named_property(env, out, "y", y));
  NAPI_OK(napi_get_boolean(env, res->is_infinity != 0, &v));
  NAPI_OK(napi_set_named_property(env, out, "isZero", v));
  NAPI_OK(napi_create_int32(env, res->c, &v));
  NAPI_OK(napi_set_named_property(env, out, "c", v));
  NAPI_OK(napi_create_int32(env, res->K, &v));
  NAPI_OK(napi_set_named_property(env, out, "K", v));
  NAPI_OK(napi_create_double(env, (double)res->n_pairs_algo, &v));
  NAPI_OK(napi_set_named_property(env, out, "nPairs", v));
  NAPI_OK(napi_create_array_with_length(env, MSM_N_PHASES, &ph));
  for (uint32_t i = 0; i < MSM_N_PHASES; i++) {
    NAPI_OK(napi_create_double(env, res->phase_ms[i], &v));
    NAPI_OK(napi_set_element(env, ph, i, v));
  }
  NAPI_OK(napi_set_named_property(env, out, "phaseMs", ph));
  return out;
}

static napi_value Msm(napi_env env, napi_callback_info info) {  // msm / msmUnsafe, src/msm-batched-affine.ts:69-340
  size_t argc = 6;   // (ctx, scalars, c, coordBytes, noGlv, unsafe)
  napi_value argv[6];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  js_ctx* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  msm_ctx* ctx = h->ctx;
  void* data;
  size_t len;
  NAPI_OK(napi_get_buffer_info(env, argv[1], &data, &len));
  if (len % 32) {
    napi_throw_range_error(env, NULL, "scalar buffer length is not a multiple of 32");
    return NULL;
  }
  msm_opts opts;
  memset(&opts, 0, sizeof opts);
  if (argc > 2) napi_get_value_int32(env, argv[2], &opts.c);   // (the fourth argument, coordBytes, is accepted and ignored)
  if (argc > 4) napi_get_value_int32(env, argv[4], &opts.no_glv);   // msmProjective, src/parallel.ts:69-87
  if (argc > 5) napi_get_value_int32(env, argv[5], &opts.unsafe);   // msmUnsafe, src/msm-batched-affine.ts:587-598
  msm_result res;
  int rc = msm_run(ctx, data, len / 32, 0, &opts, &res);
  if (rc != MSM_OK) return throw_msm(env, ctx, rc, "msm");
  return result_object(env, h, &res);
}

// device buffers: the scalars of a scalar pointer live in HBM from scalarsFromBytes / randomScalars on, as the reference's
// live in wasm memory (src/parallel.ts:119-133); msm() then crosses no PCIe
static napi_value DeviceAlloc(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  js_ctx* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  double bytes = 0;
  NAPI_OK(napi_get_value_double(env, argv[1], &bytes));
  if (!(bytes >= 0) || bytes > 9.0e15) {
    napi_throw_range_error(env, NULL, "deviceAlloc: bad size");
    return NULL;
  }
  js_dbuf* b = (js_dbuf*)calloc(1, sizeof(js_dbuf));
  if (!b) {
    napi_throw_error(env, NULL, "out of memory");
    return NULL;
  }
  b->bytes = (uint64_t)bytes < 32 ? 32 : (uint64_t)bytes;
  int rc = msm_device_alloc(h->ctx, b->bytes, &b->dev);
  if (rc != MSM_OK) {
    free(b);
    return throw_msm(env, h->ctx, rc, "deviceAlloc");
  }
  b->owner = h;
  h->refs++;
  napi_value out;
  if (napi_create_external(env, b, finalize_dbuf, NULL, &out) != napi_ok) {
    msm_device_free(h->ctx, b->dev);
    h->refs--;
    free(b);
    napi_throw_error(env, NULL, "N-API call failed: napi_create_external");
    return NULL;
  }
  int64_t adj;
  napi_adjust_external_memory(env, (int64_t)b->bytes, &adj);   // lets the collector see what a dropped handle holds
  return out;
}

static napi_value DeviceUpload(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  js_ctx* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  js_dbuf* b = get_dbuf(env, h, argv[1]);
  if (!b) return NULL;
  void* data;
  size_t len;
  NAPI_OK(napi_get_buffer_info(env, argv[2], &data, &len));
  if (len > b->bytes) {
    napi_throw_range_error(env, NULL, "deviceUpload: the Buffer is larger than the device buffer");
    return NULL;
  }
  int rc = msm_device_upload(h->ctx, b->dev, data, len);
  if (rc != MSM_OK) return throw_msm(env, h->ctx, rc, "deviceUpload");
  return NULL;
}

static napi_value DeviceFree(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  js_ctx* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  js_dbuf* b = get_dbuf(env, h, argv[1]);   // throws on a buffer that was freed before
  if (!b) return NULL;
  int rc = msm_device_free(h->ctx, b->dev);
  b->dev = NULL;
  int64_t adj;
  napi_adjust_external_memory(env, -(int64_t)b->bytes, &adj);
  if (rc != MSM_OK) return throw_msm(env, h->ctx, rc, "deviceFree");
  return NULL;
}

static napi_value MsmDevice(napi_env env, napi_callback_info info) {  // msm over scalars resident in HBM
  size_t argc = 6;   // (ctx, dbuf, n, c, noGlv, unsafe)
  napi_value argv[6];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  js_ctx* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  js_dbuf* b = get_dbuf(env, h, argv[1]);
  if (!b) return NULL;
  uint32_t n = 0;
  NAPI_OK(napi_get_value_uint32(env, argv[2], &n));
  if ((uint64_t)n * 32 > b->bytes) {
    napi_throw_range_error(env, NULL, "msmDevice: more scalars than the device buffer holds");
    return NULL;
  }
  msm_opts opts;
  memset(&opts, 0, sizeof opts);
  if (argc > 3) napi_get_value_int32(env, argv[3], &opts.c);
  if (argc > 4) napi_get_value_int32(env, argv[4], &opts.no_glv);
  if (argc > 5) napi_get_value_int32(env, argv[5], &opts.unsafe);
  msm_result res;
  int rc = msm_run(h->ctx, b->dev, n, 1, &opts, &res);
  if (rc != MSM_OK) return throw_msm(env, h->ctx, rc, "msmDevice");
  return result_object(env, h, &res);
}

static napi_value Plan(napi_env env, napi_callback_info info) {  // windowSize, src/msm-common.ts:8-41
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  msm_ctx* ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  uint32_t n = 0;
  msm_opts opts;
  memset(&opts, 0, sizeof opts);
  napi_get_value_uint32(env, argv[1], &n);
  if (argc > 2) napi_get_value_int32(env, argv[2], &opts.c);
  int32_t c = 0, K = 0;
  int rc = msm_plan(ctx, n, &opts, &c, &K);
  if (rc != MSM_OK) return throw_msm(env, ctx, rc, "plan");
  napi_value out, v;
  NAPI_OK(napi_create_object(env, &out));
  NAPI_OK(napi_create_int32(env, c, &v));
  NAPI_OK(napi_set_named_property(env, out, "c", v));
  NAPI_OK(napi_create_int32(env, K, &v));
  NAPI_OK(napi_set_named_property(env, out, "K", v));
  return out;
}

// randomPointsFast, src/curve-random.ts:14-92: n resident points generated on the GPU
static napi_value GeneratePoints(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  msm_ctx* ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  uint32_t n = 0, seed = 1;
  napi_get_value_uint32(env, argv[1], &n);
  if (argc > 2) napi_get_value_uint32(env, argv[2], &seed);
  int rc = msm_generate_points(ctx, n, seed, NULL);
  if (rc != MSM_OK) return throw_msm(env, ctx, rc, "generatePoints");
  napi_value out;
  NAPI_OK(napi_create_uint32(env, n, &out));
  return out;
}

// randomScalars, src/curve-random.ts:151-194: n uniform scalars < q.  (h, n, seed) -> n x 32 little-endian bytes in a Buffer;
// (h, n, seed, dbuf) -> written to the device buffer, nothing crosses PCIe, returns n
static napi_value GenerateScalars(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  js_ctx* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  msm_ctx* ctx = h->ctx;
  uint32_t n = 0, seed = 1;
  napi_get_value_uint32(env, argv[1], &n);
  if (argc > 2) napi_get_value_uint32(env, argv[2], &seed);
  napi_valuetype t3 = napi_undefined;
  if (argc > 3) napi_typeof(env, argv[3], &t3);
  if (t3 == napi_external) {
    js_dbuf* b = get_dbuf(env, h, argv[3]);
    if (!b) return NULL;
    if ((uint64_t)n * 32 > b->bytes) {
      napi_throw_range_error(env, NULL, "generateScalars: the device buffer is too small");
      return NULL;
    }
    int rc = msm_generate_scalars(ctx, n, seed, b->dev, NULL);
    if (rc != MSM_OK) return throw_msm(env, ctx, rc, "generateScalars");
    napi_value out;
    NAPI_OK(napi_create_uint32(env, n, &out));
    return out;
  }
  void* data = NULL;
  napi_value buf;
  NAPI_OK(napi_create_buffer(env, (size_t)n * 32, &data, &buf));
  int rc = msm_generate_scalars(ctx, n, seed, NULL, (uint8_t*)data);   /* host copy only */
  if (rc != MSM_OK) return throw_msm(env, ctx, rc, "generateScalars");
  return buf;
}

// point-set handles: pointsetCreate(h) -> id (becomes current), pointsetSelect(h, id), pointsetDestroy(h, id)
static napi_value PointsetCreate(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  msm_ctx* ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  int32_t id = 0;
  int rc = msm_pointset_create(ctx, &id);
  if (rc != MSM_OK) return throw_msm(env, ctx, rc, "pointsetCreate");
  napi_value out;
  NAPI_OK(napi_create_int32(env, id, &out));
  return out;
}
static napi_value PointsetOp(napi_env env, napi_callback_info info, int destroy) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  msm_ctx* ctx = get_ctx(env, argv[0]);
  if (!ctx) return NULL;
  int32_t id = 0;
  NAPI_OK(napi_get_value_int32(env, argv[1], &id));
  int rc = destroy ? msm_pointset_destroy(ctx, id) : msm_pointset_select(ctx, id);
  if (rc != MSM_OK) return throw_msm(env, ctx, rc, destroy ? "pointsetDestroy" : "pointsetSelect");
  return NULL;
}
static napi_value PointsetSelect(napi_env env, napi_callback_info info) { return PointsetOp(env, info, 0); }
static napi_value PointsetDestroy(napi_env env, napi_callback_info info) { return PointsetOp(env, info, 1); }

/* ---- the fine operator table: element-wise field / GLV / curve operators over Buffers -------------------------------- */

static int buffer_arg(napi_env env, napi_value v, uint8_t** data, size_t* len) {
  bool is_buf = false;
  if (napi_is_buffer(env, v, &is_buf) != napi_ok || !is_buf) {
    napi_throw_type_error(env, NULL, "expected a Buffer");
    return 0;
  }
  void* p = NULL;
  if (napi_get_buffer_info(env, v, &p, len) != napi_ok) return 0;
  *data = (uint8_t*)p;
  return 1;
}

static napi_value FieldOp(napi_env env, napi_callback_info info) {  // Field.multiply / square / add / subtract / inverse ..., src/field-msm.ts:86-123
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  js_ctx* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  int32_t op = 0;
  NAPI_OK(napi_get_value_int32(env, argv[1], &op));
  uint8_t *a = NULL, *b = NULL;
  size_t la = 0, lb = 0;
  if (!buffer_arg(env, argv[2], &a, &la)) return NULL;
  if (argc > 3) {
    napi_valuetype t;
    NAPI_OK(napi_typeof(env, argv[3], &t));
    if (t != napi_undefined && t != napi_null && !buffer_arg(env, argv[3], &b, &lb)) return NULL;
  }
  if (!b) { b = a; lb = la; }   /* one-operand operators read the second array too: hand them the first */
  if (la % h->coord_bytes || la != lb) {
    napi_throw_error(env, NULL, "fieldOp: operands must be equally long arrays of whole field elements");
    return NULL;
  }
  if (op < MSM_OP_MUL || op > MSM_OP_INV_WORDSLICED) {
    napi_throw_error(env, NULL, "fieldOp: unknown operator");
    return NULL;
  }
  const uint64_t n = la / h->coord_bytes;
  napi_value out;
  void* po = NULL;
  NAPI_OK(napi_create_buffer(env, la, &po, &out));
  if (n) {
    int rc = msm_test_fp(h->ctx, op, a, b, (uint8_t*)po, n);
    if (rc != MSM_OK) return throw_msm(env, h->ctx, rc, "fieldOp");
  }
  return out;
}

static napi_value BatchInverse(napi_env env, napi_callback_info info) {  // batchInverse, src/wasm/inverse.ts:220-271
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  js_ctx* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  uint8_t* xs = NULL;
  size_t len = 0;
  if (!buffer_arg(env, argv[1], &xs, &len)) return NULL;
  uint32_t per_lane = 0;
  NAPI_OK(napi_get_value_uint32(env, argv[2], &per_lane));
  if (len % h->coord_bytes || per_lane == 0) {
    napi_throw_error(env, NULL, "batchInverse: an array of whole field elements and a batch length >= 1");
    return NULL;
  }
  napi_value out;
  void* po = NULL;
  NAPI_OK(napi_create_buffer(env, len, &po, &out));
  if (len) {
    int rc = msm_test_batch_inverse(h->ctx, xs, (uint8_t*)po, len / h->coord_bytes, per_lane);
    if (rc != MSM_OK) return throw_msm(env, h->ctx, rc, "batchInverse");
  }
  return out;
}

static napi_value GlvDecompose(napi_env env, napi_callback_info info) {  // Scalar.decompose, src/scalar-glv.ts:105-128
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  js_ctx* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  uint8_t* sc = NULL;
  size_t len = 0;
  if (!buffer_arg(env, argv[1], &sc, &len)) return NULL;
  if (len % 32) {
    napi_throw_error(env, NULL, "glvDecompose: scalars are 32 bytes each");
    return NULL;
  }
  const uint64_t n = len / 32;
  napi_value out;
  void* po = NULL;
  NAPI_OK(napi_create_buffer(env, n * 40, &po, &out));
  if (n) {
    int rc = msm_test_glv(h->ctx, sc, (uint8_t*)po, n);
    if (rc != MSM_OK) return throw_msm(env, h->ctx, rc, "glvDecompose");
  }
  return out;
}

static napi_value BatchAdd(napi_env env, napi_callback_info info) {  // Affine.batchAdd, src/curve-affine.ts:376-522
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  js_ctx* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  uint8_t *g = NULL, *hh = NULL;
  size_t lg = 0, lh = 0;
  if (!buffer_arg(env, argv[1], &g, &lg) || !buffer_arg(env, argv[2], &hh, &lh)) return NULL;
  if (lg % h->point_bytes || lg != lh || lg == 0) {
    napi_throw_error(env, NULL, "batchAdd: two equally long, non-empty arrays of whole wire points");
    return NULL;
  }
  napi_value out;
  void* po = NULL;
  NAPI_OK(napi_create_buffer(env, lg, &po, &out));
  int rc = msm_test_batch_add(h->ctx, g, hh, (uint8_t*)po, lg / h->point_bytes);
  if (rc != MSM_OK) return throw_msm(env, h->ctx, rc, "batchAdd");
  return out;
}

NAPI_MODULE_INIT() {
  // the library found at run time must come from the header this addon was compiled against: same symbol names, other
  // struct layouts would otherwise be read wrongly without a word
  if (msm_abi_version() != MSM_ABI_VERSION || msm_abi_struct_bytes(0) != sizeof(msm_opts) || msm_abi_struct_bytes(1) != sizeof(msm_result)) {
    napi_throw_error(env, NULL, "msm_hip.node: libmsm_hip.so was built from another version of include/msm_hip.h (msm_abi_version); rebuild both");
    return NULL;
  }
  struct { const char* name; napi_callback fn; } fns[] = {
      {"createContext", CreateContext}, {"destroyContext", DestroyContext}, {"setPoints", SetPoints}, {"msm", Msm}, {"plan", Plan},
      {"generatePoints", GeneratePoints}, {"generateScalars", GenerateScalars},
      {"deviceAlloc", DeviceAlloc}, {"deviceUpload", DeviceUpload}, {"deviceFree", DeviceFree}, {"msmDevice", MsmDevice},
      {"pointsetCreate", PointsetCreate}, {"pointsetSelect", PointsetSelect}, {"pointsetDestroy", PointsetDestroy},
      {"fieldOp", FieldOp}, {"batchInverse", BatchInverse}, {"glvDecompose", GlvDecompose}, {"batchAdd", BatchAdd}};
  for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
    napi_value f;
    if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
    if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
  }
  napi_value v;
  struct { const char* name; int32_t val; } ops[] = {{"OP_MUL", MSM_OP_MUL}, {"OP_SQR", MSM_OP_SQR}, {"OP_ADD", MSM_OP_ADD}, {"OP_SUB", MSM_OP_SUB},
      {"OP_INV", MSM_OP_INV}, {"OP_TO_MONT", MSM_OP_TO_MONT}, {"OP_FROM_MONT", MSM_OP_FROM_MONT}, {"OP_INV_FERMAT", MSM_OP_INV_FERMAT},
      {"OP_INV_KALISKI", MSM_OP_INV_KALISKI}, {"OP_INV_WORDSLICED", MSM_OP_INV_WORDSLICED}};
  for (size_t i = 0; i < sizeof ops / sizeof ops[0]; i++) {
    napi_create_int32(env, ops[i].val, &v);
    napi_set_named_property(env, exports, ops[i].name, v);
  }
  napi_create_int32(env, MSM_ABI_VERSION, &v);
  napi_set_named_property(env, exports, "ABI_VERSION", v);
  napi_create_int32(env, MSM_CURVE_BLS12_377_G1, &v);
  napi_set_named_property(env, exports, "CURVE_BLS12_377_G1", v);
  napi_create_int32(env, MSM_CURVE_ED_ON_BLS12_377, &v);
  napi_set_named_property(env, exports, "CURVE_ED_ON_BLS12_377", v);
  napi_create_int32(env, MSM_CURVE_BLS12_381_G1, &v);
  napi_set_named_property(env, exports, "CURVE_BLS12_381_G1", v);
  napi_create_int32(env, MSM_CURVE_PALLAS, &v);
  napi_set_named_property(env, exports, "CURVE_PALLAS", v);
  return exports;
}
