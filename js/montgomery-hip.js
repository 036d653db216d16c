// JS facade over the N-API shim: the reference's curve-module surface for the MSM path.
//   Weierstraß.create(params) -> { params, Parallel }            src/parallel.ts:40-177
//   Parallel.{getPointer, getScalarPointer, pointsFromBytes, scalarsFromBytes, msm, msmUnsafe}
//                                                                src/parallel.ts:135-145
//   compute_msm(points, scalars) -> {x: bigint, y: bigint}       scripts/zprize23/submission-bls377.ts:20-65
// Plain CommonJS without top-level await so the image's node 12 can load it (the reference's own
// sources need node >= 20).  In the reference "pointers" are offsets into wasm memory; here they are
// small handle objects, the data lives in buffers owned by libmsm_hip.so.
"use strict";
const path = require("path");
const hip = require(path.join(__dirname, "..", "montgomery_amd", "msm_hip.node"));

const bls12377Params = {
  label: "bls12-377",
  modulus: BigInt("0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001"),
  order: BigInt("0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001"),
};
const bls12381Params = {
  label: "bls12-381",
  modulus: BigInt("0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab"),
  order: BigInt("0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001"),
};
const pallasParams = {
  label: "pallas",
  modulus: BigInt("0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001"),
  order: BigInt("0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001"),
};
const edOnBls12377Params = {
  label: "ed-on-bls12-377",
  modulus: BigInt("0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001"),
  order: BigInt("0x4aad957a68b2955982d1347970dec005293a3afc43c8afeb95aee9ac33fd9ff"),
};

function leBytesToBigint(buf) {
  let x = BigInt(0);
  for (let i = buf.length - 1; i >= 0; i--) x = (x << BigInt(8)) | BigInt(buf[i]);
  return x;
}
function bigintToLeBytes(x, n) {
  const out = Buffer.alloc(n);
  for (let i = 0; i < n; i++) { out[i] = Number(x & BigInt(255)); x >>= BigInt(8); }
  return out;
}

function createCurve(params, curveId, coordBytes, device, wireBytes) {
  wireBytes = wireBytes || coordBytes;   // the reference's packed coordinate size = the C ABI's: 48, or 32 for the 255-bit fields
  const ctx = hip.createContext(curveId, device || 0);   // device: an index, or a list of indices (one context over several GPUs)
  const pointBytes = 2 * coordBytes;
  // `let [pointPtr] = await Parallel.randomPointsFast(N)` -- the reference hands back one pointer per point / scalar and its
  // callers keep the first (scripts/msm-weierstrass.ts:18,21,29); a handle here stands for the whole array, so it unpacks to itself
  const firstOf = (ptr) => { ptr[Symbol.iterator] = function* () { yield ptr; }; return ptr; };
  const Parallel = {
    // every point pointer is its own resident point set of the context, as every pointer of the reference is its own
    // allocation; msm() selects the set of the pointer it is given
    getPointer(size) { return { size, n: 0, set: hip.pointsetCreate(ctx), free() { hip.pointsetDestroy(ctx, this.set); } }; },
    // A scalar pointer owns ONE device buffer from scalarsFromBytes / randomScalars on: the reference's scalars live in the
    // memory its kernels compute in (src/parallel.ts:119-133), so msm() crosses no PCIe.  free() gives the buffer back at
    // once; a pointer that is simply dropped gives it back when it is collected (the addon registers a finalizer).
    getScalarPointer(size) { return newScalarPtr(size); },
    async pointsFromBytes(pointPtr, input, n) {
      let b = Buffer.from(input.buffer, input.byteOffset, n * 2 * wireBytes);
      if (wireBytes !== coordBytes) {
        const padded = Buffer.alloc(n * pointBytes);
        for (let i = 0; i < 2 * n; i++) b.copy(padded, i * coordBytes, i * wireBytes, (i + 1) * wireBytes);
        b = padded;
      }
      hip.pointsetSelect(ctx, pointPtr.set);
      pointPtr.n = hip.setPoints(ctx, b, pointBytes, 0);
    },
    async scalarsFromBytes(scalarPtr, input, n) {   // src/parallel.ts:119-133: one upload, then resident
      scalarPtr.free();
      scalarPtr.dev = hip.deviceAlloc(ctx, Math.max(32 * n, 32));
      hip.deviceUpload(ctx, scalarPtr.dev, Buffer.from(input.buffer, input.byteOffset, n * 32));
      scalarPtr.n = n;
    },
    async randomPointsFast(n, options) {   // src/curve-random.ts:14-92; generated on the GPU, explicit seed
      const pointPtr = Parallel.getPointer(n * 2 * wireBytes);
      pointPtr.n = hip.generatePoints(ctx, n, (options && options.seed) || 1);
      return firstOf(pointPtr);
    },
    async randomScalars(n, options) {      // src/curve-random.ts:151-194: generated in HBM, nothing crosses PCIe
      const ptr = newScalarPtr(n * 32);
      const seed = (options && options.seed) || 1;
      ptr.dev = hip.deviceAlloc(ctx, Math.max(32 * n, 32));
      hip.generateScalars(ctx, n, seed, ptr.dev);
      ptr.n = n;
      ptr.toBytes = () => hip.generateScalars(ctx, n, seed);   // the same stream read back (tests)
      return firstOf(ptr);
    },
    async msm(scalarPtr, pointPtr, N, verboseTiming, options) {
      const c = (options && options.c) || 0;
      if (N > pointPtr.n) throw new Error(`msm: ${N} scalars but ${pointPtr.n} points behind this pointer`);
      if (!scalarPtr.dev || N > scalarPtr.n) throw new Error(`msm: ${N} scalars requested but the scalar pointer holds ${scalarPtr.dev ? scalarPtr.n : 0}`);
      hip.pointsetSelect(ctx, pointPtr.set);
      const unsafe = options && options.useSafeAdditions === false ? 1 : 0;
      const r = hip.msmDevice(ctx, scalarPtr.dev, N, c, options && options.noGlv ? 1 : 0, unsafe);
      const result = { x: leBytesToBigint(r.x), y: leBytesToBigint(r.y), isZero: r.isZero };
      return { result, log: verboseTiming ? buildLog(N, r) : [] };
    },
    msmProjective(scalarPtr, pointPtr, N, options) {
      // src/parallel.ts:69-87: signed windows of the whole scalar, no endomorphism split (same group element)
      return Parallel.msm(scalarPtr, pointPtr, N, false, Object.assign({}, options, { noGlv: true }));
    },
    msmUnsafe(scalarPtr, pointPtr, N, verboseTiming, options) {   // src/msm-batched-affine.ts:587-598
      return Parallel.msm(scalarPtr, pointPtr, N, verboseTiming, Object.assign({}, options, { useSafeAdditions: false }));
    },
  };
  function newScalarPtr(size) {
    return { size, n: 0, dev: null, free() { if (this.dev) { const d = this.dev; this.dev = null; this.n = 0; hip.deviceFree(ctx, d); } } };
  }
  // `log` in the reference's shape (createLog, src/msm-common.ts:176-214; filled at src/msm-batched-affine.ts:79-338): one
  // entry with the parameters, then one "label... x.xms" line per phase, "msm total" last.  The phases are the library's
  // eight device timings (msm_result.phase_ms) under the reference's labels where a counterpart exists.
  function buildLog(N, r) {
    const t = r.phaseMs, line = (label, ms) => [`${label}... ${ms.toFixed(1)}ms`];
    return [
      [{ n: Math.ceil(Math.log2(Math.max(N, 1))), K: r.K, c: r.c }],   // log({ n, K, c }), src/msm-batched-affine.ts:93
      line("scalars to device", t[1]),
      line("slice scalars & count buckets", t[2]),
      line("sort points", t[3]),
      line("bucket accumulation (first round)", t[7]),
      line("bucket accumulation", t[4]),
      line("bucket reduction (local)", t[5]),
      line("final sum", t[6]),
      line("msm total", t[0]),
    ];
  }
  // What the reference's callers do with `result` (scripts/msm-weierstrass.ts:89-91):
  //     let sAffinePtr = Curve.Field.getPointer(Curve.Affine.size);
  //     Curve.Projective.toAffine(scratch, sAffinePtr, result);
  //     let s = Curve.Affine.toBigint(sAffinePtr);
  // There `result` points at a projective point in wasm memory; here the library has already normalised it, so the three
  // calls only hand the value through: pointers are small objects, toAffine stores, toBigint returns {x, y, isZero}.
  const newPtr = (size) => ({ size, value: null });
  // The fine operator table (the reference's wasm exports, src/field-msm.ts:86-123,190-243): element-wise over Buffers of
  // n field elements (coordBytes each, little-endian, MONTGOMERY form like the reference's field elements in wasm memory)
  // instead of pointers into wasm memory; every call is one kernel launch over all n elements.  fromBigints / toBigints are
  // fromPackedBytes + toMontgomery and their inverse (src/field-msm.ts:108-115).
  const fieldOp = (op) => (a, b) => hip.fieldOp(ctx, op, a, b);
  const Field = {
    getPointer: newPtr, getPointers(n, size) { return Array.from({ length: n }, () => newPtr(size)); },
    sizeInBytes: coordBytes,
    multiply: fieldOp(hip.OP_MUL), square: fieldOp(hip.OP_SQR), add: fieldOp(hip.OP_ADD), subtract: fieldOp(hip.OP_SUB),
    inverse: fieldOp(hip.OP_INV), toMontgomery: fieldOp(hip.OP_TO_MONT), fromMontgomery: fieldOp(hip.OP_FROM_MONT),
    batchInverse(xs, perLane) { return hip.batchInverse(ctx, xs, perLane || 64); },   // src/wasm/inverse.ts:220-271
    fromBigints(vals) { return hip.fieldOp(ctx, hip.OP_TO_MONT, Buffer.concat(vals.map((v) => bigintToLeBytes(BigInt(v), coordBytes)))); },
    toBigints(buf) {
      const plain = hip.fieldOp(ctx, hip.OP_FROM_MONT, buf), out = [];
      for (let i = 0; i < plain.length; i += coordBytes) out.push(leBytesToBigint(plain.slice(i, i + coordBytes)));
      return out;
    },
  };
  // Scalar.decompose (src/scalar-glv.ts:105-128, src/wasm/glv.ts:68-169): s = (-1)^neg0 s0 + (-1)^neg1 s1 lambda mod q
  const Scalar = {
    decompose(scalars) {
      const sb = Buffer.isBuffer(scalars) ? scalars : Buffer.concat(scalars.map((v) => bigintToLeBytes(BigInt(v), 32)));
      const raw = hip.glvDecompose(ctx, sb), out = [];
      for (let i = 0; i < raw.length; i += 40)
        out.push({ s0: leBytesToBigint(raw.slice(i, i + 16)), s1: leBytesToBigint(raw.slice(i + 16, i + 32)),
                   neg0: raw.readUInt32LE(i + 32) !== 0, neg1: raw.readUInt32LE(i + 36) !== 0 });
      return out;
    },
  };
  const toBigint = (ptr) => { const r = (ptr && ptr.value) || ptr; return { x: r.x, y: r.y, isZero: !!r.isZero }; };
  // Affine.batchAdd (batchAddNew, src/curve-affine.ts:376-522): n pairs of affine points {x, y} | null (identity) -> n sums,
  // all through one launch of the tree kernel with its shared inversions; every kind of pair (P + P, P - P, identities) is handled
  const encPoint = (P) => (P ? Buffer.concat([bigintToLeBytes(BigInt(P.x), coordBytes), bigintToLeBytes(BigInt(P.y), coordBytes)]) : Buffer.alloc(pointBytes));
  const batchAdd = (G, H) => {
    const out = hip.batchAdd(ctx, Buffer.concat(G.map(encPoint)), Buffer.concat(H.map(encPoint))), sums = [];
    for (let i = 0; i < out.length; i += pointBytes) {
      const x = leBytesToBigint(out.slice(i, i + coordBytes)), y = leBytesToBigint(out.slice(i + coordBytes, i + pointBytes));
      sums.push(x === BigInt(0) && y === BigInt(0) ? null : { x, y });
    }
    return sums;
  };
  const Affine = { size: 2 * wireBytes + 4, toBigint, batchAdd };                         // src/curve-affine.ts:77, 220-233, 376-522
  const Projective = { size: 3 * wireBytes + 4, toAffine(_scratch, affinePtr, result) { affinePtr.value = toBigint(result); } };   // src/curve-projective.ts:335-349
  // twisted Edwards callers: `Curve.Curve.toBigint(result)` -> extended bigint point, `Curve.Bigint.toAffine(P)` -> {x, y}
  // (scripts/msm-twisted-edwards.ts:87, scripts/zprize23/submission.ts:33-34)
  const P_MOD = params.modulus;
  const modInv = (a) => { let [r0, r1, s0, s1] = [((a % P_MOD) + P_MOD) % P_MOD, P_MOD, BigInt(1), BigInt(0)];
    while (r1 !== BigInt(0)) { const q = r0 / r1; [r0, r1] = [r1, r0 - q * r1]; [s0, s1] = [s1, s0 - q * s1]; }
    return ((s0 % P_MOD) + P_MOD) % P_MOD; };
  const Curve = { toBigint(result) { const r = toBigint(result); return { X: r.x, Y: r.y, Z: BigInt(1), T: (r.x * r.y) % P_MOD }; } };
  const Bigint = { toAffine(P) { const zi = modInv(P.Z); return { x: (P.X * zi) % P_MOD, y: (P.Y * zi) % P_MOD }; } };
  return { params, Parallel, Field, Scalar, Affine, Projective, Curve, Bigint, close() { hip.destroyContext(ctx); } };
}

const weierstrassIds = { "bls12-377": hip.CURVE_BLS12_377_G1, "bls12-381": hip.CURVE_BLS12_381_G1, "pallas": hip.CURVE_PALLAS };
const Weierstrass = {
  create(params, device) {
    if (!(params.label in weierstrassIds)) throw new Error(`curve ${params.label} has no device constants`);
    return createCurve(params, weierstrassIds[params.label], params.label === "pallas" ? 32 : 48, device);
  },
};
const TwistedEdwards = { create(params, device) { return createCurve(params, hip.CURVE_ED_ON_BLS12_377, 32, device); } };

// The ZPrize entry with the reference's own signature -- compute_msm(points, scalars) -- is exported per curve by
// js/submission-bls377.js and js/submission.js (scripts/zprize23/submission-bls377.ts:20-23, submission.ts:19-22);
// this is the shared body.  points: {x, y, isZero}[] | Buffer, scalars: bigint[] | Buffer -> {x, y}
async function compute_msm_on(curve, coordBytes, inputPoints, inputScalars) {
  const pointBytes = 2 * coordBytes;
  let sbytes, pbytes;
  if (Buffer.isBuffer(inputScalars) || inputScalars instanceof Uint8Array) sbytes = Buffer.from(inputScalars);
  else sbytes = Buffer.concat(inputScalars.map((s) => bigintToLeBytes(BigInt(s), 32)));
  const n = sbytes.length / 32;
  if (Buffer.isBuffer(inputPoints) || inputPoints instanceof Uint8Array) pbytes = Buffer.from(inputPoints);
  else pbytes = Buffer.concat(inputPoints.map((P) => (P.isZero ? Buffer.alloc(pointBytes) : Buffer.concat([bigintToLeBytes(BigInt(P.x), coordBytes), bigintToLeBytes(BigInt(P.y), coordBytes)]))));
  const pp = curve.Parallel.getPointer(pbytes.length);
  const sp = curve.Parallel.getScalarPointer(sbytes.length);
  try {   // point set and scalar buffer are freed whatever the conversion or the MSM throws (bad point, HIP error): `curve` outlives the call
    await curve.Parallel.pointsFromBytes(pp, pbytes, n);
    await curve.Parallel.scalarsFromBytes(sp, sbytes, n);
    const same = n > 1 && pbytes.slice(0, pointBytes).equals(pbytes.slice(pointBytes, 2 * pointBytes));
    const { result } = same ? await curve.Parallel.msm(sp, pp, n) : await curve.Parallel.msmUnsafe(sp, pp, n);
    return { x: result.x, y: result.y, isZero: result.isZero };
  } finally {
    pp.free();
    sp.free();
  }
}

// `startThreads(n)` / `stopThreads()` of src/parallel.ts:291-320 -- the reference's callers bracket every MSM with them
// (scripts/msm-weierstrass.ts:14,50, src/msm.test.ts:23,33).  The worker pool they manage is replaced by the GPU grid, which
// needs no start-up: both resolve at once.  `n` is accepted and ignored (it sized the pool and the memory segmentation).
async function startThreads(_n) {}
async function stopThreads() {}

module.exports = { hip, startThreads, stopThreads, Weierstrass, Weierstraß: Weierstrass /* the reference's spelling, src/parallel.ts:40 */, TwistedEdwards, bls12377Params, bls12381Params, pallasParams, edOnBls12377Params, compute_msm_on, compute_msm: compute_msm_on, leBytesToBigint, bigintToLeBytes };
