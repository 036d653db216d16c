// js/submission-bls377.js and js/submission.js: the reference's ZPrize entry points
// (scripts/zprize23/submission-bls377.ts:20-23, submission.ts:19-22)
export function compute_msm(
  inputPoints: { x: bigint; y: bigint; isZero?: boolean }[] | Uint8Array,
  inputScalars: bigint[] | Uint8Array
): Promise<{ x: bigint; y: bigint }>;
