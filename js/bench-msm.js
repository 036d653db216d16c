// The reference's benchmark of one curve, through this facade with the reference's own calls
// (scripts/msm-weierstrass.ts:12-51: start the workers, random points once, fresh random scalars per run, a warm-up, 15 timed
// runs of which the first 5 are dropped, median and sample standard deviation).  randomScalars generates into a device
// buffer of the context (the reference: into wasm memory), so the timed call is the resident-scalar form bench.py times.
//     node js/bench-msm.js [log2 n = 16] [curve = bls12-377 | bls12-381 | pallas]
"use strict";
const { Weierstraß, startThreads, stopThreads, bls12377Params, bls12381Params, pallasParams } = require("./montgomery-hip.js");

function median(xs) {
  const s = xs.slice().sort((a, b) => a - b), h = s.length >> 1;
  return s.length % 2 ? s[h] : (s[h - 1] + s[h]) / 2;
}
function sampleStd(xs) {
  const mean = xs.reduce((a, b) => a + b, 0) / xs.length;
  return Math.sqrt(xs.reduce((a, b) => a + (b - mean) * (b - mean), 0) / (xs.length - 1));
}
const ms = (t0) => Number(process.hrtime.bigint() - t0) / 1e6;

async function benchmarkMsm(params, n) {
  const N = 1 << n;
  await startThreads();
  const Curve = Weierstraß.create(params);
  const { Parallel } = Curve;
  const [pointPtr] = await Parallel.randomPointsFast(N);
  let [scalarPtr] = await Parallel.randomScalars(N);
  await Parallel.msmUnsafe(scalarPtr, pointPtr, Math.min(N, 1 << 15), true);   // warm-up (workspace allocation)
  const times = [];
  for (let i = 0; i < 15; i++) {
    scalarPtr.free();   // (the reference's scalars sit in a scoped arena; here the handle owns a device buffer)
    [scalarPtr] = await Parallel.randomScalars(N, { seed: 100 + i });
    const t0 = process.hrtime.bigint();
    await Parallel.msmUnsafe(scalarPtr, pointPtr, N, true);
    const t = ms(t0);
    if (i > 4) times.push(t);
  }
  const { result, log } = await Parallel.msmUnsafe(scalarPtr, pointPtr, N, true);
  // the reference's way to bigints, and its consistency check against a second implementation (msmProjective here)
  const sAffinePtr = Curve.Field.getPointer(Curve.Affine.size);
  Curve.Projective.toAffine(null, sAffinePtr, result);
  const s = Curve.Affine.toBigint(sAffinePtr);
  const other = Curve.Affine.toBigint((await Parallel.msmProjective(scalarPtr, pointPtr, N)).result);
  if (s.isZero !== other.isZero || s.x !== other.x || s.y !== other.y) throw new Error("msm and msmProjective disagree");
  log.forEach((l) => console.log(...l));
  console.log(times.map((t) => +t.toFixed(2)));
  const out = { curve: params.label, n, median_ms: median(times), std_ms: sampleStd(times), points_per_s: N / (median(times) * 1e-3), runs: times.length, consistent: true };
  console.log(`msm (n=${n})... ${out.median_ms.toFixed(2)}ms ± ${out.std_ms.toFixed(2)}ms`);
  console.log(JSON.stringify(out));
  pointPtr.free();
  scalarPtr.free();
  Curve.close();
  await stopThreads();
  return out;
}

const curves = { "bls12-377": bls12377Params, "bls12-381": bls12381Params, pallas: pallasParams };
if (require.main === module) {
  const n = parseInt(process.argv[2] || "16", 10), params = curves[process.argv[3] || "bls12-377"];
  benchmarkMsm(params, n).catch((e) => { console.error(e); process.exit(1); });
}
module.exports = { benchmarkMsm };
