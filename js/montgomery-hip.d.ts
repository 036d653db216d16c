// Type declarations of js/montgomery-hip.js in the vocabulary of the reference's TypeScript sources
// (src/parallel.ts:135-160, 251-289; scripts/zprize23/submission-bls377.ts:20-23).
export type CurveParams = { label: string; modulus: bigint; order: bigint };

/** one resident point set of the context (its own allocation, like every pointer of the reference) */
export type PointPtr = { size: number; n: number; set: number; free(): void };
/** one device buffer of the context from scalarsFromBytes / randomScalars on (the reference's scalars live in wasm memory);
 * free() returns it at once, a dropped pointer returns it when collected */
export type ScalarPtr = { size: number; n: number; dev: unknown | null; free(): void; toBytes?: () => Buffer };

/** canonical affine result: what `Affine.toBigint(Projective.toAffine(result))` yields in the reference */
export type AffineResult = { x: bigint; y: bigint; isZero: boolean };
export type MsmOptions = { c?: number; useSafeAdditions?: boolean; noGlv?: boolean };
/** log: the reference's shape (src/msm-common.ts:176-214) -- [{n, K, c}], then ["label... x.xms"] per phase, "msm total" last */
export type MsmOutput = { result: AffineResult; log: unknown[][] };

export interface Parallel {
  getPointer(size: number): PointPtr;
  getScalarPointer(size: number): ScalarPtr;
  /** src/parallel.ts:97-116 (Weierstrass: x || y, packed field size each) / :215-229 (twisted Edwards) */
  pointsFromBytes(pointPtr: PointPtr, input: Uint8Array, n: number): Promise<void>;
  /** src/parallel.ts:119-133: n x 32 bytes little-endian */
  scalarsFromBytes(scalarPtr: ScalarPtr, input: Uint8Array, n: number): Promise<void>;
  /** src/curve-random.ts:14-92, generated on the GPU (explicit seed; the reference is unseeded) */
  randomPointsFast(n: number, options?: { seed?: number }): Promise<PointPtr>;
  /** src/curve-random.ts:151-194 */
  randomScalars(n: number, options?: { seed?: number }): Promise<ScalarPtr>;
  /** src/msm-batched-affine.ts:69-340 */
  msm(scalarPtr: ScalarPtr, pointPtr: PointPtr, N: number, verboseTiming?: boolean, options?: MsmOptions): Promise<MsmOutput>;
  /** src/msm-batched-affine.ts:587-598: msm with useSafeAdditions = false (msm_opts.unsafe) */
  msmUnsafe(scalarPtr: ScalarPtr, pointPtr: PointPtr, N: number, verboseTiming?: boolean, options?: MsmOptions): Promise<MsmOutput>;
  /** src/parallel.ts:69-87: window structure of msmBasic, no endomorphism split */
  msmProjective(scalarPtr: ScalarPtr, pointPtr: PointPtr, N: number, options?: MsmOptions): Promise<MsmOutput>;
}

/** opaque stand-in for a wasm pointer of the reference (`Field.getPointer(size)`) */
export type ValuePtr = { size: number; value: AffineResult | null };
/** The fine operator table of the reference's wasm exports (src/field-msm.ts:86-123, 190-243) over Buffers of n field elements
 * (sizeInBytes each, little-endian, Montgomery form) instead of pointers into wasm memory: one kernel launch per call. */
export interface FieldOps {
  getPointer(size: number): ValuePtr;
  getPointers(n: number, size: number): ValuePtr[];
  sizeInBytes: number;
  multiply(a: Buffer, b: Buffer): Buffer;
  square(a: Buffer): Buffer;
  add(a: Buffer, b: Buffer): Buffer;
  subtract(a: Buffer, b: Buffer): Buffer;
  inverse(a: Buffer): Buffer;
  /** src/wasm/inverse.ts:220-271: Montgomery's trick, perLane elements per inversion */
  batchInverse(xs: Buffer, perLane?: number): Buffer;
  toMontgomery(a: Buffer): Buffer;
  fromMontgomery(a: Buffer): Buffer;
  fromBigints(vals: bigint[]): Buffer;
  toBigints(buf: Buffer): bigint[];
}
export interface Curve {
  params: CurveParams;
  Parallel: Parallel;
  /** the reference's way from `result` to bigints (scripts/msm-weierstrass.ts:89-91); values are already affine here */
  Field: FieldOps;
  /** src/scalar-glv.ts:105-128: s = (-1)^neg0 s0 + (-1)^neg1 s1 lambda mod q (Weierstrass curves with an endomorphism) */
  Scalar: { decompose(scalars: bigint[] | Buffer): { s0: bigint; s1: bigint; neg0: boolean; neg1: boolean }[] };
  Affine: {
    size: number;
    toBigint(ptr: ValuePtr | AffineResult): AffineResult;
    /** batchAddNew, src/curve-affine.ts:376-522: n pairs (null = identity) -> n sums through one launch of the tree kernel */
    batchAdd(G: ({ x: bigint; y: bigint } | null)[], H: ({ x: bigint; y: bigint } | null)[]): ({ x: bigint; y: bigint } | null)[];
  };
  Projective: { size: number; toAffine(scratch: unknown, affinePtr: ValuePtr, result: AffineResult): void };
  /** twisted Edwards call sites (scripts/msm-twisted-edwards.ts:87, scripts/zprize23/submission.ts:33-34) */
  Curve: { toBigint(result: AffineResult | ValuePtr): { X: bigint; Y: bigint; Z: bigint; T: bigint } };
  Bigint: { toAffine(P: { X: bigint; Y: bigint; Z: bigint }): { x: bigint; y: bigint } };
  close(): void;
}

/** device: a GPU index, or a list of indices: one curve object over several GPUs of the node (windows sharded inside the library) */
export const Weierstrass: { create(params: CurveParams, device?: number | number[]): Curve };
export const TwistedEdwards: { create(params: CurveParams, device?: number | number[]): Curve };
/** the reference's spelling (src/parallel.ts:40) */
export const Weierstraß: typeof Weierstrass;
export const bls12377Params: CurveParams;
export const bls12381Params: CurveParams;
export const pallasParams: CurveParams;
export const edOnBls12377Params: CurveParams;

/** shared body of the per-curve entries js/submission-bls377.js and js/submission.js, which export the reference's exact
 * `compute_msm(inputPoints, inputScalars)` (scripts/zprize23/submission-bls377.ts:20-65, submission.ts:19-60) */
export function compute_msm_on(
  curve: Curve,
  coordBytes: number,
  inputPoints: { x: bigint; y: bigint; isZero?: boolean }[] | Uint8Array,
  inputScalars: bigint[] | Uint8Array
): Promise<{ x: bigint; y: bigint }>;
export function compute_msm(
  curve: Curve,
  coordBytes: number,
  inputPoints: { x: bigint; y: bigint; isZero?: boolean }[] | Uint8Array,
  inputScalars: bigint[] | Uint8Array
): Promise<{ x: bigint; y: bigint }>;
/** src/parallel.ts:291-320: no-ops here (the GPU grid is the worker pool); kept so that the reference's call sites run unchanged */
export function startThreads(n?: number): Promise<void>;
export function stopThreads(): Promise<void>;
export function leBytesToBigint(buf: Uint8Array): bigint;
export function bigintToLeBytes(x: bigint, n: number): Buffer;
