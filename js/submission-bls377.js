// ZPrize entry point for BLS12-377 G1, with the reference's exact signature
// (scripts/zprize23/submission-bls377.ts:20-23):
//     compute_msm(inputPoints: BigIntPoint[] | U32ArrayPoint[] | Buffer, inputScalars: bigint[] | Uint32Array[] | Buffer)
//       -> Promise<{x: bigint, y: bigint}>
// The curve object (context on GPU 0, points and scalars uploaded per call like the reference does) is created on
// first use and kept for the life of the process.
"use strict";
const m = require("./montgomery-hip.js");
let curve = null;
async function compute_msm(inputPoints, inputScalars) {
  if (!curve) curve = m.Weierstrass.create(m.bls12377Params);
  return m.compute_msm_on(curve, 48, inputPoints, inputScalars);
}
module.exports = { compute_msm };
