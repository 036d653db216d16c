// The fine operator table through the N-API boundary (Field.* / Scalar.decompose / Affine.batchAdd of js/montgomery-hip.js, the
// counterparts of the reference's wasm exports, src/field-msm.ts:190-243, src/scalar-glv.ts:105-128) replayed against the
// committed golden vectors tests/golden/fp377.json, glv377.json and point_add377.json (made by tests/golden/make_golden.py from
// the oracle).  usage: node js/test-field.js   (prints one JSON line; exit code 1 on any mismatch)
"use strict";
const fs = require("fs");
const path = require("path");
const { Weierstrass, bls12377Params } = require("./montgomery-hip.js");
const gold = (f) => JSON.parse(fs.readFileSync(path.join(__dirname, "..", "tests", "golden", f)));
const H = (s) => BigInt(s);
const P = bls12377Params.modulus;
let failures = [];
const check = (what, got, want) => { if (got !== want) failures.push(`${what}: got ${got}, want ${want}`); };

const curve = Weierstrass.create(bls12377Params);
const { Field, Scalar, Affine } = curve;

// field operators: Montgomery form in, Montgomery form out
const fp = gold("fp377.json").cases;
const a = Field.fromBigints(fp.map((c) => H(c.a))), b = Field.fromBigints(fp.map((c) => H(c.b)));
const ops = { mul: Field.multiply(a, b), add: Field.add(a, b), sub: Field.subtract(a, b), sqr: Field.square(a) };
for (const key of Object.keys(ops)) {
  const got = Field.toBigints(ops[key]);
  fp.forEach((c, i) => check(`fp ${key}[${i}]`, got[i], H(c[key])));
}
const nz = fp.filter((c) => c.inv !== null);
const am = Field.fromBigints(nz.map((c) => H(c.a)));
const inv = Field.toBigints(Field.inverse(am));
nz.forEach((c, i) => check(`fp inv[${i}]`, inv[i], H(c.inv)));
for (const perLane of [1, 3, 64]) {
  const bi = Field.toBigints(Field.batchInverse(am, perLane));
  nz.forEach((c, i) => check(`fp batchInverse(${perLane})[${i}]`, bi[i], H(c.inv)));
}
// a * a^-1 = 1, through the boundary alone
Field.toBigints(Field.multiply(am, Field.inverse(am))).forEach((v, i) => check(`a * a^-1 [${i}]`, v, BigInt(1)));

// GLV decomposition
const glv = gold("glv377.json").cases;
Scalar.decompose(glv.map((c) => H(c.s))).forEach((r, i) => {
  const c = glv[i];
  check(`glv s0[${i}]`, r.s0, H(c.s0)); check(`glv s1[${i}]`, r.s1, H(c.s1));
  check(`glv neg0[${i}]`, r.neg0, c.neg0); check(`glv neg1[${i}]`, r.neg1, c.neg1);
});

// batched affine addition, every kind of pair
const pa = gold("point_add377.json").cases;
const dec = (Q) => (Q === null ? null : { x: H(Q[0]), y: H(Q[1]) });
Affine.batchAdd(pa.map((c) => dec(c.g)), pa.map((c) => dec(c.h))).forEach((S, i) => {
  const want = dec(pa[i].sum);
  check(`batchAdd[${i}]`, S === null ? "inf" : `${S.x},${S.y}`, want === null ? "inf" : `${want.x},${want.y}`);
});

curve.close();
console.log(JSON.stringify({ ok: failures.length === 0, fp_cases: fp.length, inverses: nz.length, glv_cases: glv.length, batch_add_cases: pa.length,
                             failures: failures.slice(0, 5).map(String), modulus_bits: P.toString(2).length }));
process.exit(failures.length ? 1 : 0);
