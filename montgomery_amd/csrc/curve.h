// Short-Weierstrass (a = 0) group law on device, one point per lane.
//
//   affine pair addition with shared inversion : see k_batch_add in msm_kernels.h
//       (reference: batchAddNew / batchAddUnsafeNew, src/curve-affine.ts:376-522,
//        addAffine src/wasm/curve.ts:32-58, Affine.double src/curve-affine.ts:90-109)
//   homogeneous projective add / mixed add / double (this file)
//       (reference: src/curve-projective.ts:51-160 add-1998-cmo-2, :202-253 dbl-1998-cmo-2)
//
// A projective point holds X, Y, Z in register form; Z == 0 (canonical zero) is the identity.
// All coordinates handed between functions are < 2p with normalized limbs.
#pragma once
#include "field.h"

namespace msm {

template <class C>
struct Proj {
  Fe<C> X, Y, Z;
};

template <class C>
MSM_DEV void proj_set_zero(Proj<C>& P) {
#pragma unroll
  for (int i = 0; i < C::NL; i++) { P.X.l[i] = 0; P.Z.l[i] = 0; }
  fe_set_one<C>(P.Y);
}

// zero test for a lazily reduced value < 4p
template <class C>
MSM_DEV bool fe_is_zero_mod_p(Fe<C> a) {
  fe_reduce_4p<C>(a);
  return fe_is_zero_canonical<C>(a);
}

template <class C>
MSM_DEV bool proj_is_zero(const Proj<C>& P) { return fe_is_zero_mod_p<C>(P.Z); }

// small multiples with partial reduction: inputs < 2p, outputs < 2p
template <class C>
MSM_DEV void fe_dbl_r(Fe<C>& r, const Fe<C>& a) {
  fe_add<C>(r, a, a);          // < 4p
  fe_cond_sub<C, 2>(r);    // < 2p
}
template <class C>
MSM_DEV void fe_add_r(Fe<C>& r, const Fe<C>& a, const Fe<C>& b) {
  fe_add<C>(r, a, b);
  fe_cond_sub<C, 2>(r);
}
template <class C>
MSM_DEV void fe_sub_r(Fe<C>& r, const Fe<C>& a, const Fe<C>& b) {  // a, b < 2p -> < 2p
  fe_sub_2p<C>(r, a, b);       // < 4p
  fe_cond_sub<C, 2>(r);
}

// dbl-1998-cmo-2 (a = 0): reference src/curve-projective.ts:202-253
template <class C>
MSM_DEV void proj_double(Proj<C>& R, const Proj<C>& P) {
  if (proj_is_zero<C>(P)) { proj_set_zero<C>(R); return; }
  Fe<C> w, s, ss, sss, Rr, B, h, t, u;
  fe_sqr<C>(t, P.X);
  fe_add<C>(w, t, t);
  fe_add<C>(w, w, t);                 // w = 3 X^2   (< 4.5p, fine as a mul operand)
  fe_mul<C>(s, P.Y, P.Z);             // s = Y Z
  fe_sqr<C>(ss, s);
  fe_mul<C>(sss, s, ss);
  fe_mul<C>(Rr, P.Y, s);              // R = Y s
  fe_mul<C>(B, P.X, Rr);              // B = X R
  fe_sqr<C>(h, w);                    // w^2
  Fe<C> B2, B4, B8;
  fe_dbl_r<C>(B2, B);
  fe_dbl_r<C>(B4, B2);
  fe_dbl_r<C>(B8, B4);
  fe_sub_r<C>(h, h, B8);              // h = w^2 - 8B
  fe_mul<C>(t, h, s);
  fe_dbl_r<C>(R.X, t);                // X3 = 2 h s
  fe_sub_r<C>(u, B4, h);              // 4B - h
  fe_mul<C>(u, w, u);                 // w (4B - h)
  fe_sqr<C>(t, Rr);                   // R^2
  fe_dbl_r<C>(t, t);
  fe_dbl_r<C>(t, t);
  fe_dbl_r<C>(t, t);                  // 8 R^2
  fe_sub_r<C>(R.Y, u, t);             // Y3
  fe_dbl_r<C>(t, sss);
  fe_dbl_r<C>(t, t);
  fe_dbl_r<C>(R.Z, t);                // Z3 = 8 s^3
}

// add-1998-cmo-2 with the reference's edge cases (src/curve-projective.ts:51-160):
// zero operands, equal points (-> double), opposite points (-> zero).
// MIXED: Q has Z = 1 (Q.Z ignored), saving three multiplications.
template <class C, bool MIXED>
MSM_DEV void proj_add_impl(Proj<C>& R, const Proj<C>& P, const Proj<C>& Q, bool q_is_zero) {
  if (q_is_zero) { R = P; return; }
  if (proj_is_zero<C>(P)) {
    R.X = Q.X; R.Y = Q.Y;
    if (MIXED) fe_set_one<C>(R.Z); else R.Z = Q.Z;
    return;
  }
  Fe<C> Y1Z2, X1Z2, Z1Z2, u, v, t;
  if (MIXED) { Y1Z2 = P.Y; X1Z2 = P.X; Z1Z2 = P.Z; }
  else {
    fe_mul<C>(Y1Z2, P.Y, Q.Z);
    fe_mul<C>(X1Z2, P.X, Q.Z);
    fe_mul<C>(Z1Z2, P.Z, Q.Z);
  }
  fe_mul<C>(t, Q.Y, P.Z);
  fe_sub_r<C>(u, t, Y1Z2);            // u = Y2 Z1 - Y1 Z2
  fe_mul<C>(t, Q.X, P.Z);
  fe_sub_r<C>(v, t, X1Z2);            // v = X2 Z1 - X1 Z2
  if (fe_is_zero_mod_p<C>(v)) {
    if (fe_is_zero_mod_p<C>(u)) { proj_double<C>(R, P); return; }
    proj_set_zero<C>(R);
    return;
  }
  Fe<C> uu, vv, vvv, Rr, A;
  fe_sqr<C>(uu, u);
  fe_sqr<C>(vv, v);
  fe_mul<C>(vvv, v, vv);
  fe_mul<C>(Rr, vv, X1Z2);
  fe_mul<C>(A, uu, Z1Z2);
  fe_sub_r<C>(A, A, vvv);
  fe_dbl_r<C>(t, Rr);
  fe_sub_r<C>(A, A, t);               // A = uu Z1Z2 - vvv - 2R
  fe_mul<C>(R.X, v, A);
  fe_sub_r<C>(t, Rr, A);
  fe_mul<C>(t, u, t);                 // u (R - A)
  Fe<C> t2;
  fe_mul<C>(t2, vvv, Y1Z2);
  fe_sub_r<C>(R.Y, t, t2);
  fe_mul<C>(R.Z, vvv, Z1Z2);
}

template <class C>
MSM_DEV void proj_add(Proj<C>& R, const Proj<C>& P, const Proj<C>& Q) {
  proj_add_impl<C, false>(R, P, Q, proj_is_zero<C>(Q));
}
// Q affine (x, y) in Q.X, Q.Y; q_inf tells whether Q is the identity
template <class C>
MSM_DEV void proj_add_mixed(Proj<C>& R, const Proj<C>& P, const Proj<C>& Q, bool q_inf) {
  proj_add_impl<C, true>(R, P, Q, q_inf);
}

}  // namespace msm
