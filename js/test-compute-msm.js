// Mirrors the reference's self-tests scripts/zprize23/submission-test-bls377.ts and submission-test.ts through
// the N-API shim, plus the committed golden vectors (tests/golden/*.json).  Run on a GPU box: node js/test-compute-msm.js
"use strict";
const fs = require("fs");
const path = require("path");
const M = require("./montgomery-hip.js");
const subBls = require("./submission-bls377.js");   // compute_msm(points, scalars): the reference's signature
const subEd = require("./submission.js");

function assert(c, msg) { if (!c) { console.error("FAILED: " + msg); process.exit(1); } }

async function main() {
  await M.startThreads(16);   // the reference's call sites bracket their MSMs with these (scripts/msm-weierstrass.ts:14,50)
  // --- BLS12-377 G1 (submission-test-bls377.ts:6-45)
  const bls = M.Weierstrass.create(M.bls12377Params);
  const point = {
    x: BigInt("111871295567327857271108656266735188604298176728428155068227918632083036401841336689521497731900230387779623820740"),
    y: BigInt("76860045326390600098227152997486448974650822224305058012700629806287380625419427989664237630603922765089083164740"),
    isZero: false,
  };
  const q = M.bls12377Params.order;
  let r = await subBls.compute_msm([point, point], [BigInt(2), q - BigInt(1)]);
  assert(r.x === point.x && r.y === point.y, "2P + (q-1)P = P");
  r = await M.compute_msm(bls, 48, [point, point], [BigInt(2), q - BigInt(1)]);
  assert(r.x === point.x && r.y === point.y, "2P + (q-1)P = P (explicit curve object)");
  console.log("2 points ok");
  const n = 1000;
  let scalars = [], sum = BigInt(0);
  let seed = BigInt(12345);
  for (let i = 0; i < n; i++) {
    seed = (seed * BigInt("6364136223846793005") + BigInt("1442695040888963407")) % (BigInt(1) << BigInt(250));
    scalars.push(seed % q);
    sum = (sum + scalars[i]) % q;
  }
  const r2 = await subBls.compute_msm(Array(n).fill(point), scalars);
  const r3 = await subBls.compute_msm([point], [sum]);
  assert(r2.x === r3.x && r2.y === r3.y, "same points: msm = (sum s) P");
  console.log("same points ok");
  const gold = JSON.parse(fs.readFileSync(path.join(__dirname, "..", "tests", "golden", "msm377.json"), "utf8"));
  for (const c of gold.cases) {
    const pts = Buffer.from(c.points, "hex"), sc = Buffer.from(c.scalars, "hex");
    const pp = bls.Parallel.getPointer(pts.length), sp = bls.Parallel.getScalarPointer(sc.length);
    await bls.Parallel.pointsFromBytes(pp, pts, c.n);
    await bls.Parallel.scalarsFromBytes(sp, sc, c.n);
    const { result, log } = await bls.Parallel.msmUnsafe(sp, pp, c.n, true, { c: c.c || 0 });
    if (c.result === null) assert(result.isZero, c.name + " should be the identity");
    else assert(!result.isZero && result.x === BigInt(c.result[0]) && result.y === BigInt(c.result[1]), "golden " + c.name);
    assert(log.length > 0, "log");
    // the reference's own way from `result` to bigints (scripts/msm-weierstrass.ts:89-91)
    const scratch = bls.Field.getPointers(20, 48), sAffinePtr = bls.Field.getPointer(bls.Affine.size);
    bls.Projective.toAffine(scratch, sAffinePtr, result);
    const s = bls.Affine.toBigint(sAffinePtr);
    assert(s.isZero === result.isZero && s.x === result.x && s.y === result.y, "Projective.toAffine / Affine.toBigint " + c.name);
  }
  console.log("golden bls12-377 ok:", gold.cases.length, "cases");
  {
    // two point pointers side by side: the second pointsFromBytes must not replace the points behind the first
    const a = gold.cases[0], b = gold.cases[1];
    const pa = bls.Parallel.getPointer(a.points.length / 2), pb = bls.Parallel.getPointer(b.points.length / 2);
    const sa = bls.Parallel.getScalarPointer(a.scalars.length / 2), sb = bls.Parallel.getScalarPointer(b.scalars.length / 2);
    await bls.Parallel.pointsFromBytes(pa, Buffer.from(a.points, "hex"), a.n);
    await bls.Parallel.pointsFromBytes(pb, Buffer.from(b.points, "hex"), b.n);
    await bls.Parallel.scalarsFromBytes(sa, Buffer.from(a.scalars, "hex"), a.n);
    await bls.Parallel.scalarsFromBytes(sb, Buffer.from(b.scalars, "hex"), b.n);
    for (const [c, sp, pp] of [[a, sa, pa], [b, sb, pb], [a, sa, pa]]) {
      const { result } = await bls.Parallel.msm(sp, pp, c.n, false, { c: c.c || 0 });
      if (c.result === null) assert(result.isZero, "pointer " + c.name);
      else assert(result.x === BigInt(c.result[0]) && result.y === BigInt(c.result[1]), "pointer " + c.name);
    }
    console.log("two point pointers ok");
  }
  bls.close();
  let threw = false;
  try { bls.close(); } catch (e) { threw = true; }
  assert(threw, "a second close() of the same context must throw, not double-free");
  threw = false;
  try { M.hip.setPoints(M.hip.createContext(M.hip.CURVE_BLS12_377_G1, 0), Buffer.alloc(96), 64, 0); } catch (e) { threw = true; }
  assert(threw, "setPoints with a point size that is not the curve's must throw");

  // --- Ed-on-BLS12-377 (submission-test.ts:5-21)
  const ed = M.TwistedEdwards.create(M.edOnBls12377Params);
  const ep = {
    x: BigInt("2796670805570508460920584878396618987767121022598342527208237783066948667246"),
    y: BigInt("8134280397689638111748378379571739274369602049665521098046934931245960532166"),
  };
  r = await subEd.compute_msm([ep, ep], [BigInt(2), M.edOnBls12377Params.order - BigInt(1)]);
  assert(r.x === ep.x && r.y === ep.y, "ed: 2P + (q-1)P = P");
  const goldEd = JSON.parse(fs.readFileSync(path.join(__dirname, "..", "tests", "golden", "msm_ed377.json"), "utf8"));
  for (const c of goldEd.cases) {
    const rr = await M.compute_msm(ed, 32, Buffer.from(c.points, "hex"), Buffer.from(c.scalars, "hex"));
    assert(rr.x === BigInt(c.result[0]) && rr.y === BigInt(c.result[1]), "golden ed " + c.name);
  }
  {
    // the reference's way from `result` to affine bigints on this curve (scripts/zprize23/submission.ts:33-34)
    const c = goldEd.cases[0], pts = Buffer.from(c.points, "hex"), sc = Buffer.from(c.scalars, "hex");
    const pp = ed.Parallel.getPointer(pts.length), sp = ed.Parallel.getScalarPointer(sc.length);
    await ed.Parallel.pointsFromBytes(pp, pts, c.n);
    await ed.Parallel.scalarsFromBytes(sp, sc, c.n);
    const { result } = await ed.Parallel.msm(sp, pp, c.n);
    const big = ed.Curve.toBigint(result), aff = ed.Bigint.toAffine(big);
    assert(aff.x === BigInt(c.result[0]) && aff.y === BigInt(c.result[1]) && big.T === (big.X * big.Y) % M.edOnBls12377Params.modulus, "Curve.toBigint / Bigint.toAffine");
    pp.free();
  }
  console.log("ed-on-bls12-377 ok");
  ed.close();

  // --- the other Weierstrass curves of src/msm.test.ts:29-31 through their golden vectors (48-byte coordinates in the
  //     fixture; Pallas takes 32-byte coordinates on the wire, src/concrete/pasta.params.ts:17)
  for (const [params, file, wire] of [[M.bls12381Params, "bls381.json", 48], [M.pallasParams, "pallas.json", 32]]) {
    const cv = M.Weierstrass.create(params);
    const g = JSON.parse(fs.readFileSync(path.join(__dirname, "..", "tests", "golden", file), "utf8"));
    for (const c of g.msm) {
      const full = Buffer.from(c.points, "hex"), sc = Buffer.from(c.scalars, "hex");
      const pts = Buffer.alloc(c.n * 2 * wire);
      for (let i = 0; i < 2 * c.n; i++) full.copy(pts, i * wire, i * 48, i * 48 + wire);
      const pp = cv.Parallel.getPointer(pts.length), sp = cv.Parallel.getScalarPointer(sc.length);
      await cv.Parallel.pointsFromBytes(pp, pts, c.n);
      await cv.Parallel.scalarsFromBytes(sp, sc, c.n);
      const a = (await cv.Parallel.msm(sp, pp, c.n, false, { c: c.c || 0 })).result;
      const b = (await cv.Parallel.msmProjective(sp, pp, c.n, { c: 8 })).result;   // src/parallel.ts:69-87
      for (const res of [a, b]) {
        if (c.result === null) assert(res.isZero, params.label + " " + c.name + " should be the identity");
        else assert(!res.isZero && res.x === BigInt(c.result[0]) && res.y === BigInt(c.result[1]), params.label + " golden " + c.name);
      }
    }
    // randomPointsFast / randomScalars (src/curve-random.ts), as src/msm.test.ts:49-50 uses them: msm == msmProjective
    const n = 1 << 12;
    const rp = await cv.Parallel.randomPointsFast(n, { seed: 5 }), rs = await cv.Parallel.randomScalars(n, { seed: 6 });
    assert(rp.n === n && rs.n === n && rs.dev && rs.toBytes().length === 32 * n, "generators");
    const out1 = await cv.Parallel.msmUnsafe(rs, rp, n, true);
    const m1 = out1.result, m2 = (await cv.Parallel.msmProjective(rs, rp, n)).result;
    assert(!m1.isZero && m1.x === m2.x && m1.y === m2.y, params.label + ": msm == msmProjective on generated inputs");
    // the scalars are resident: the same values uploaded from host bytes (scalarsFromBytes: one upload, then resident) and
    // the safe entry give the same element; the log has the reference's shape (src/msm-common.ts:176-214)
    const sp2 = cv.Parallel.getScalarPointer(32 * n);
    await cv.Parallel.scalarsFromBytes(sp2, rs.toBytes(), n);
    const m3 = (await cv.Parallel.msm(sp2, rp, n)).result;
    assert(m3.x === m1.x && m3.y === m1.y, params.label + ": uploaded scalars == generated scalars, safe == unsafe");
    const log = out1.log;
    assert(log.length >= 7 && log[0][0].K > 0 && log[0][0].c > 0 && typeof log[0][0].n === "number", "log: parameters first");
    assert(log.slice(1).every((l) => /^[a-z &()]+\.\.\. \d+\.\dms$/.test(l[0])), "log: 'label... x.xms' lines");
    assert(log[log.length - 1][0].startsWith("msm total..."), "log: msm total last");
    assert((await cv.Parallel.msm(rs, rp, n)).log.length === 0, "log: empty unless verboseTiming");
    sp2.free(); rs.free();
    let threw = false;
    try { await cv.Parallel.msm(rs, rp, n); } catch (e) { threw = true; }
    assert(threw, "a freed scalar pointer must be refused");
    console.log(params.label, "ok:", g.msm.length, "golden cases + generated 2^12");
    cv.close();
  }
  await M.stopThreads();
  console.log("ALL OK");
}
main().catch((e) => { console.error(e); process.exit(1); });
