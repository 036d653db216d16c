"""The product's field / GLV templates (montgomery_amd/csrc/field.h, glv.h) compiled for the CPU and
checked against Python integers and the oracle -- host logic, no GPU needed."""
import ctypes as C
import os
import subprocess

import pytest

from oracle import msm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "csrc", "libfield_host.so")

FIELDS = {
    0: (O.BLS12_377.p, 12, 13),   # modulus, packed words, 30-bit limbs
    1: (O.ED_ON_BLS12_377.p, 8, 9),
    2: (0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, 12, 13),  # BLS12-381: p != 1 mod 2^30
    3: (O.PALLAS.p, 8, 9),    # Pallas: 255 bits on 9 limbs / 8 words
}


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-C", ROOT, "-s", "hosttest"])   # own shim library, never loaded by anything else
    lib = C.CDLL(LIB)
    lib.host_fp_op.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.host_glv.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    return lib


def fp_op(lib, field, which, a, b=0):
    _, nw, _ = FIELDS[field]
    A = (C.c_uint32 * nw)(*[(a >> (32 * i)) & 0xFFFFFFFF for i in range(nw)])
    B = (C.c_uint32 * nw)(*[(b >> (32 * i)) & 0xFFFFFFFF for i in range(nw)])
    out = (C.c_uint32 * nw)()
    lib.host_fp_op(field, which, A, B, out)
    return sum(int(w) << (32 * i) for i, w in enumerate(out))


@pytest.mark.parametrize("field", [0, 1, 2, 3])
def test_mul_sqr_add_sub(lib, field):
    p, _, nl = FIELDS[field]
    R = 1 << (30 * nl)
    rinv = pow(R, -1, p)
    vals = [0, 1, 2, p - 1, p - 2, (p + 1) // 2, (1 << 30) - 1, 1 << 30] + O.prng_ints(f"host/fp{field}", 300, p)
    for i, a in enumerate(vals):
        b = vals[-1 - i]
        assert fp_op(lib, field, 0, a, b) == a * b * rinv % p
        assert fp_op(lib, field, 1, a) == a * a * rinv % p
        assert fp_op(lib, field, 2, a, b) == (a + b) % p
        assert fp_op(lib, field, 3, a, b) == (a - b) % p


@pytest.mark.parametrize("field", [0, 1, 2, 3])
def test_inverse_divsteps_fermat_kaliski(lib, field):
    """fe_inv (division steps) == fe_inv_fermat == fe_inv_kaliski == a^-1 R^2 for Montgomery-form input a R."""
    p, _, nl = FIELDS[field]
    R = 1 << (30 * nl)
    vals = [1, 2, p - 1, p - 2, (p + 1) // 2, 3, 1 << 200, (1 << 252) - 1] + O.prng_ints(f"host/inv{field}", 400, p)
    for a in vals:
        a %= p
        if a == 0:
            continue
        exp = pow(a, -1, p) * R * R % p
        assert fp_op(lib, field, 4, a) == exp, hex(a)
    for a in vals[:40]:
        assert fp_op(lib, field, 5, a % p) == pow(a % p, -1, p) * R * R % p
    # the reference's own algorithm, Kaliski's almost-inverse (src/wasm/inverse.ts:136-218), as the third variant
    for a in vals[:120]:
        assert fp_op(lib, field, 6, a % p) == pow(a % p, -1, p) * R * R % p, hex(a)
    # the reference's experimental word-sliced almost-inverse (src/inverse/faster-inverse-wasm.ts:133-343), fourth variant
    for a in vals[:200] + [p - 1, 1, 2, (1 << 117) - 1, (1 << (p.bit_length() - 1)) + 1]:
        assert fp_op(lib, field, 7, a % p) == pow(a % p, -1, p) * R * R % p, hex(a)
    assert fp_op(lib, field, 4, 0) == 0 and fp_op(lib, field, 6, 0) == 0 and fp_op(lib, field, 7, 0) == 0


@pytest.mark.parametrize("curve", [0, 2, 3])
def test_glv_decompose_host_build(lib, curve):
    Cc = {0: O.BLS12_377, 2: O.BLS12_381, 3: O.PALLAS}[curve]
    g = O.glv_params(Cc.q, Cc.lam)
    for s in O.prng_ints("host/glv", 3000, Cc.q) + [0, 1, Cc.q - 1, Cc.lam, Cc.lam + 1]:
        S = (C.c_uint32 * 8)(*[(s >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
        out = (C.c_uint32 * 10)()
        lib.host_glv(curve, S, out)
        a0 = sum(int(out[i]) << (32 * i) for i in range(4))
        a1 = sum(int(out[4 + i]) << (32 * i) for i in range(4))
        assert (a0, a1, bool(out[8]), bool(out[9])) == O.glv_decompose(s, g)


def _worst_case_limb_operands(p, nl, seed):
    """Operands the kernels actually feed the multiplier (field.h: a * b < 2^12 p^2, limbs normalised): canonical values,
    sums / differences of a few elements up to 2^6 p, values just below multiples of p, and limb patterns that maximise
    the column sums (all limbs 2^30 - 1 up to the top limb of the bound)."""
    vals = [0, 1, p - 1, p, p + 1, 2 * p - 1, 4 * p - 3, 63 * p, 64 * p - 1, (1 << (30 * (nl - 1))) - 1]
    vals.append((1 << ((64 * p).bit_length() - 1)) - 1)   # every limb below the top one is all ones, still under the bound
    vals += [v * k + d for v in O.prng_ints(f"host/raw{seed}", 60, p) for k, d in ((1, 0), (7, 3), (63, 0))]
    return [v for v in vals if v < 64 * p]


@pytest.mark.parametrize("field", [0, 1, 2, 3])
def test_mul_sqr_on_unreduced_and_all_ones_operands(lib, field):
    """fe_mul / fe_sqr on raw limbs: the result is congruent to a b / R, below p + a b / R (so below 1.5 p whenever
    a b < 2^12 p^2 ... here up to 2^12 p^2 exactly at the corner), and its limbs are normalised.  Also the all-ones
    limb pattern 2^(30 NL) - 1, beyond the contract: no accumulator may wrap, the value must still be congruent."""
    p, _, nl = FIELDS[field]
    R = 1 << (30 * nl)
    rinv = pow(R, -1, p)
    lib.host_fp_raw.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]

    def raw(which, a, b):
        A = (C.c_uint32 * nl)(*[(a >> (30 * i)) & 0x3FFFFFFF if i < nl - 1 else a >> (30 * i) for i in range(nl)])
        B = (C.c_uint32 * nl)(*[(b >> (30 * i)) & 0x3FFFFFFF if i < nl - 1 else b >> (30 * i) for i in range(nl)])
        out = (C.c_uint32 * nl)()
        lib.host_fp_raw(field, which, A, B, out)
        assert all(int(w) < (1 << 30) for w in out[: nl - 1]), "limbs not normalised"
        return sum(int(w) << (30 * i) for i, w in enumerate(out))

    vals = _worst_case_limb_operands(p, nl, field)
    for i, a in enumerate(vals):
        b = vals[-1 - i]
        r = raw(0, a, b)
        assert r % p == a * b * rinv % p and r < p + a * b // R + 1
        r = raw(1, a, a)
        assert r % p == a * a * rinv % p and r < p + a * a // R + 1
    ones = R - 1
    assert raw(0, ones, ones) % p == ones * ones * rinv % p
    assert raw(1, ones, ones) % p == ones * ones * rinv % p
