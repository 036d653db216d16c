// ZPrize entry point for the twisted Edwards curve Ed-on-BLS12-377, with the reference's exact signature
// (scripts/zprize23/submission.ts:19-22): compute_msm(inputPoints, inputScalars) -> Promise<{x: bigint, y: bigint}>.
"use strict";
const m = require("./montgomery-hip.js");
let curve = null;
async function compute_msm(inputPoints, inputScalars) {
  if (!curve) curve = m.TwistedEdwards.create(m.edOnBls12377Params);
  return m.compute_msm_on(curve, 32, inputPoints, inputScalars);
}
module.exports = { compute_msm };
