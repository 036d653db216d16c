// Host-side unit-test shim: the product's field / GLV templates (montgomery_amd/csrc/field.h, glv.h)
// compiled for the CPU so their arithmetic can be checked without a GPU (tests/test_host_field.py).
#include "field.h"
#include "glv.h"
using namespace msm;

template <class C>
static void load(Fe<C>& r, const uint32_t* w) {
  uint32_t t[C::NW];
  for (int i = 0; i < C::NW; i++) t[i] = w[i];
  fe_unpack<C>(r, t);
}
template <class C>
static void store(uint32_t* w, Fe<C> a) {
  fe_reduce_4p<C>(a);
  uint32_t t[C::NW];
  fe_pack<C>(t, a);
  for (int i = 0; i < C::NW; i++) w[i] = t[i];
}

template <class C>
static void op(int which, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  Fe<C> x, y, r;
  load<C>(x, a);
  load<C>(y, b);
  switch (which) {
    case 0: fe_mul<C>(r, x, y); break;
    case 1: fe_sqr<C>(r, x); break;
    case 2: fe_add<C>(r, x, y); break;
    case 3: fe_sub_p<C>(r, x, y); break;
    case 4: fe_inv<C>(r, x); break;
    case 5: fe_inv_fermat<C>(r, x); break;
    case 6: fe_inv_kaliski<C>(r, x); break;
    case 7: fe_inv_wordsliced<C>(r, x); break;
    default: r = x;
  }
  store<C>(out, r);
}

// raw-limb multiplier / squaring: NL limbs in, NL limbs out, nothing reduced (worst-case operand tests)
template <class C>
static void op_raw(int which, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  Fe<C> x, y, r;
  for (int i = 0; i < C::NL; i++) { x.l[i] = a[i]; y.l[i] = b[i]; }
  if (which == 1) fe_sqr<C>(r, x); else fe_mul<C>(r, x, y);
  for (int i = 0; i < C::NL; i++) out[i] = r.l[i];
}
extern "C" {
// field 0 = Fp377 (12 words), 1 = Fp253 (8 words), 2 = Fp381 (12 words), 3 = FpPallas (8 words); operands are canonical Montgomery-form words
void host_fp_op(int field, int which, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  if (field == 0) op<Fp377>(which, a, b, out);
  else if (field == 1) op<Fp253>(which, a, b, out);
  else if (field == 2) op<Fp381>(which, a, b, out);
  else op<FpPallas>(which, a, b, out);
}
void host_fp_raw(int field, int which, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  if (field == 0) op_raw<Fp377>(which, a, b, out);
  else if (field == 1) op_raw<Fp253>(which, a, b, out);
  else if (field == 2) op_raw<Fp381>(which, a, b, out);
  else op_raw<FpPallas>(which, a, b, out);
}
// curve 0 = BLS12-377 lattice, 2 = BLS12-381 lattice, 3 = Pallas lattice
void host_glv(int curve, const uint32_t* s8, uint32_t* out10) {
  uint32_t s[8];
  for (int i = 0; i < 8; i++) s[i] = s8[i];
  GlvHalf h0, h1;
  if (curve == 2) glv_decompose<GlvBls381>(h0, h1, s);
  else if (curve == 3) glv_decompose<GlvPallas>(h0, h1, s);
  else glv_decompose<GlvBls377>(h0, h1, s);
  for (int i = 0; i < 4; i++) { out10[i] = h0.mag[i]; out10[4 + i] = h1.mag[i]; }
  out10[8] = h0.neg; out10[9] = h1.neg;
}
}
