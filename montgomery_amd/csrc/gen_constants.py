#!/usr/bin/env python3
"""Generates constants_gen.h: per-field limb tables for the device (30-bit limbs, R = 2^(30*NL))
and packed 32-bit word tables for the host.  Run from the repo root:
    python montgomery_amd/csrc/gen_constants.py
Parameter sources: src/concrete/bls12-377.params.ts:11-34, src/concrete/ed-on-bls12-377.params.ts:5-22
(values restated in oracle/msm_oracle.py are NOT imported here: the product does not depend on oracle/).
"""
import os

LB = 30

FP377 = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
FR377 = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
FR_ED = 0x4AAD957A68B2955982D1347970DEC005293A3AFC43C8AFEB95AEE9AC33FD9FF
LAMBDA = 0x12AB655E9A2CA55660B44D1E5C37B00114885F32400000000000000000000000
BETA = 0x1AE3A4617C510EABC8756BA8F8C524EB8882A75CC9BC8E359064EE822FB5BFFD1E945779FFFFFFFFFFFFFFFFFFFFFFF
G1X = 0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF
G1Y = 0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6
FP381 = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
FR381 = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
LAMBDA381 = 0xD201000000010000 ** 2 - 1
BETA381 = 0x1A0111EA397FE699EC02408663D4DE85AA0D857D89759AD4897D29650FB85F9B409427EB4F49FFFD8BFD00000000AAAC
G381X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G381Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
# Pallas (src/concrete/pasta.params.ts:10-53): y^2 = x^3 + 5, lambda = 5^((q-1)/3), beta = (5^((p-1)/3))^2
FP_PALLAS = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
FQ_PALLAS = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
LAMBDA_PALLAS = pow(5, (FQ_PALLAS - 1) // 3, FQ_PALLAS)
BETA_PALLAS = pow(pow(5, (FP_PALLAS - 1) // 3, FP_PALLAS), 2, FP_PALLAS)
GX_PALLAS = 1
GY_PALLAS = 0x1B74B5A30A12937C53DFA9F06378EE548F655BD4333D477119CF7A23CAED2ABB
ED_D = 3021
EDGX = 0x9F1B5A5BAF6ACF06FED91C9AE9EBFA06068DD2835790980894E2328F3EBCA05
EDGY = 0x9A20DF36571AC3CD906B256080BA8454453C177AAF3131BB50A67BF1A806781


def limbs(x, n, bits=LB):
    out = [(x >> (bits * i)) & ((1 << bits) - 1) for i in range(n)]
    assert x >> (bits * n) == 0
    return out


def arr(name, vals, ty="uint32_t"):
    body = ", ".join(f"0x{v:x}u" for v in vals)
    return f"  static constexpr {ty} {name}[{len(vals)}] = {{{body}}};\n"


def field_block(struct, p, nl, nw, extra=None, nla=None):
    R = 1 << (LB * nl)
    assert R > 4 * p
    mu = (-pow(p, -1, 1 << LB)) % (1 << LB)
    s = f"struct {struct} {{\n"
    s += f"  static constexpr int NL = {nl};   // 30-bit limbs in registers\n"
    s += f"  static constexpr int NW = {nw};   // packed 32-bit words in memory\n"
    nla = nla or nl
    assert (1 << (LB * nla)) > 64 * p   # every value the kernels form (sums of a few elements, < 64 p) fits its active limbs
    s += f"  static constexpr int NLA = {nla};  // limbs a value of this field can occupy: operands of fe_mul / fe_sqr are zero above\n"
    s += f"  static constexpr int BITS = {p.bit_length()};\n"
    s += f"  static constexpr uint32_t MU = 0x{mu:x}u;  // -p^-1 mod 2^30\n"
    s += f"  static constexpr uint32_t PINV30 = 0x{pow(p, -1, 1 << LB):x}u;  // p^-1 mod 2^30\n"
    s += arr("P", limbs(p, nl))
    s += arr("P2", limbs(2 * p, nl))
    s += arr("P4", limbs(4 * p, nl))
    s += arr("PW", limbs(p, nw, 32))
    s += arr("ONE", limbs(R % p, nl))            # Montgomery 1
    s += arr("ONEW", limbs(R % p, nw, 32))
    s += arr("R2W", limbs(R * R % p, nw, 32))    # to-Montgomery multiplier
    s += arr("PM2W", limbs(p - 2, nw, 32))       # Fermat exponent
    s += arr("R2", limbs(R * R % p, nl))
    s += arr("R3", limbs(R * R * R % p, nl))       # turns x^-1 (plain) of a Montgomery-form x into Montgomery form
    for k, v in (extra or {}).items():
        s += arr(k, limbs(v * R % p, nw, 32))
        s += arr(k[:-1] + "L", limbs(v * R % p, nl))
    s += "};\n\n"
    return s


def egcd_stop_early(lam, q):
    r0, r1, t0, t1 = q, lam, 0, 1
    while r1 * r1 > q:
        k = r0 // r1
        r0, r1 = r1, r0 - k * r1
        t0, t1 = t1, t0 - k * t1
    k = r0 // r1
    r2, t2 = r0 - k * r1, t0 - k * t1
    v00, v10 = r1, -t1
    v01, v11 = (r0, -t0) if max(r0, abs(t0)) <= max(r2, abs(t2)) else (r2, -t2)
    return v00, v01, v10, v11


def tdiv(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def glv_block(name="GlvBls377", lam=None, q=None, max_bits=126):
    # constants of glvGeneral (src/wasm/glv.ts:35-63) for w = 29: n = 9, n0 = 5, m = 145, k = 116
    lam = LAMBDA if lam is None else lam
    q = FR377 if q is None else q
    w, n = 29, 9
    n0 = (n + 1) // 2
    m, k = n0 * w, (n - n0) * w
    v00, v01, v10, v11 = egcd_stop_early(lam, q)
    det = v00 * v11 - v10 * v01
    m0 = tdiv((1 << (m + k)) * -v11, det)
    m1 = tdiv((1 << (m + k)) * v10, det)
    s = f"struct {name} {{\n"
    s += f"  static constexpr int M_SHIFT = {m};\n  static constexpr int K_SHIFT = {k};\n"
    s += f"  static constexpr int MAX_BITS = {max_bits};  // src/wasm/glv.ts:216-226 evaluated (checked in tests)\n"
    for name, v in (("V00", v00), ("V01", v01), ("V10", v10), ("V11", v11), ("M0", m0), ("M1", m1)):
        s += f"  static constexpr int {name}_NEG = {1 if v < 0 else 0};\n"
        s += arr(name, limbs(abs(v), 5, 32))
    s += arr("Q", limbs(q, 8, 32))
    s += "};\n\n"
    return s


def main():
    out = "// GENERATED by gen_constants.py -- do not edit.\n#pragma once\n#include <stdint.h>\nnamespace msm {\n\n"
    out += field_block("Fp377", FP377, 13, 12, {"BETAW": BETA, "BW": 1, "GXW": G1X, "GYW": G1Y})
    out += field_block("Fp253", FR377, 9, 8, {"K2DW": 2 * ED_D, "DW": ED_D, "GXW": EDGX, "GYW": EDGY})
    # BLS12-381 G1 (src/concrete/bls12-381.params.ts:6-55): y^2 = x^3 + 4, lambda = z^2 - 1, z = 0xd201000000010000
    out += field_block("Fp381", FP381, 13, 12, {"BETAW": BETA381, "BW": 4, "GXW": G381X, "GYW": G381Y})
    # Pallas runs through the same 13-limb / 12-word code path with zero upper limbs (R = 2^390 is a valid
    # Montgomery radix for any odd p < R): a curve "by constants only", at the 381-bit path's cost
    out += field_block("FpPallas", FP_PALLAS, 9, 8, {"BETAW": BETA_PALLAS, "BW": 5, "GXW": GX_PALLAS, "GYW": GY_PALLAS})
    out += glv_block()
    out += glv_block("GlvPallas", LAMBDA_PALLAS, FQ_PALLAS, 127)
    out += glv_block("GlvBls381", LAMBDA381, FR381, 127)
    out += arr("FR377_Q", limbs(FR377, 8, 32)).replace("  static", "static")
    out += arr("FRED_Q", limbs(FR_ED, 8, 32)).replace("  static", "static")
    out += "\n}  // namespace msm\n"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "constants_gen.h")
    with open(path, "w") as f:
        f.write(out)
    print("wrote", path)


if __name__ == "__main__":
    main()
