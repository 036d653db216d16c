// Host-side 384-bit prime-field and projective-curve arithmetic for the O(K*c) tail of the MSM:
// the Horner combination of the K window sums and the final projective -> affine conversion.
// The reference runs exactly this step on its main thread as well
// (src/msm-batched-affine.ts:306-333 "this whole stage takes < 0.2ms and is done on the main thread",
//  toAffine: src/curve-projective.ts:335-349).  It is independent of N: K*c doublings + K additions.
//
// 6 x 64-bit limbs of storage, values kept canonical in [0, p).  The Montgomery radix is 2^(64 nl) with nl the limbs the
// modulus really has: 6 (radix 2^384) for the 377- and 381-bit primes, 4 (radix 2^256) for the 255- and 253-bit ones, whose
// products then take 16 + 16 instead of 36 + 36 word multiplications (the host tail of the Edwards MSM: 0.085 -> 0.04 ms).
#pragma once
#include <stdint.h>
#include <string.h>

namespace msm_host {

typedef unsigned __int128 u128;

struct Fe6 {
  uint64_t v[6];
};

struct Field6 {
  Fe6 p;
  uint64_t pinv;  // -p^-1 mod 2^64
  Fe6 one;        // R mod p, R = 2^(64 nl)
  Fe6 r2;         // R^2 mod p
  int nl = 6;     // active limbs: p < 2^(64 nl - 1)
  int radix_bits() const { return 64 * nl; }

  static bool ge(const Fe6& a, const Fe6& b) {
    for (int i = 5; i >= 0; i--) {
      if (a.v[i] > b.v[i]) return true;
      if (a.v[i] < b.v[i]) return false;
    }
    return true;
  }
  static bool is_zero(const Fe6& a) {
    uint64_t o = 0;
    for (int i = 0; i < 6; i++) o |= a.v[i];
    return o == 0;
  }
  static bool eq(const Fe6& a, const Fe6& b) { return memcmp(a.v, b.v, sizeof(a.v)) == 0; }

  void sub_raw(Fe6& r, const Fe6& a, const Fe6& b) const {
    u128 br = 0;
    for (int i = 0; i < 6; i++) {
      u128 d = (u128)a.v[i] - b.v[i] - (uint64_t)br;
      r.v[i] = (uint64_t)d;
      br = (d >> 64) & 1;
    }
  }
  void add(Fe6& r, const Fe6& a, const Fe6& b) const {
    u128 c = 0;
    Fe6 t;
    for (int i = 0; i < 6; i++) {
      c += (u128)a.v[i] + b.v[i];
      t.v[i] = (uint64_t)c;
      c >>= 64;
    }
    if (c || ge(t, p)) sub_raw(t, t, p);
    r = t;
  }
  void sub(Fe6& r, const Fe6& a, const Fe6& b) const {
    Fe6 t;
    if (ge(a, b)) {
      sub_raw(t, a, b);
    } else {
      Fe6 u;
      sub_raw(u, b, a);
      sub_raw(t, p, u);
    }
    r = t;
  }
  void dbl(Fe6& r, const Fe6& a) const { add(r, a, a); }

  // Montgomery product a*b*2^-384 mod p, inputs canonical.  CIOS with the product row and the reduction row of every round
  // fused into one pass (two independent carry chains); p < 2^383 -- true of every modulus here (377, 381, 255, 253 bits) --
  // means the top word of a round cannot overflow, so no extra carry word is kept.
  void mul(Fe6& r, const Fe6& a, const Fe6& b) const {
    if (nl == 4) mul_n<4>(r, a, b); else mul_n<6>(r, a, b);
  }
  template <int N>
  void mul_n(Fe6& r, const Fe6& a, const Fe6& b) const {
    uint64_t t[N];
    for (int i = 0; i < N; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      u128 A = (u128)a.v[i] * b.v[0] + t[0];
      const uint64_t m = (uint64_t)A * pinv;
      u128 C = (u128)m * p.v[0] + (uint64_t)A;
      uint64_t ca = (uint64_t)(A >> 64), cc = (uint64_t)(C >> 64);
#pragma unroll
      for (int j = 1; j < N; j++) {
        A = (u128)a.v[i] * b.v[j] + t[j] + ca;
        ca = (uint64_t)(A >> 64);
        C = (u128)m * p.v[j] + (uint64_t)A + cc;
        cc = (uint64_t)(C >> 64);
        t[j - 1] = (uint64_t)C;
      }
      t[N - 1] = ca + cc;
    }
    Fe6 o = {{0, 0, 0, 0, 0, 0}};
    for (int i = 0; i < N; i++) o.v[i] = t[i];
    if (ge(o, p)) sub_raw(o, o, p);
    r = o;
  }
  void sqr(Fe6& r, const Fe6& a) const { mul(r, a, a); }

  // a^(p-2), Montgomery in / out
  // a R -> a^-1 R.  The plain inverse comes from the binary extended Euclid (shifts and subtractions on six limbs, a third
  // of the time of the p - 2 power: this inversion is on the critical path of every MSM, once), two products restore the form.
  void inv(Fe6& r, const Fe6& a) const {
    if (is_zero(a)) { r = a; return; }
    Fe6 t;
    inv_plain(t, a);      // a^-1 R^-1
    mul(t, t, r2);        // a^-1
    mul(r, t, r2);        // a^-1 R
  }
  // x^-1 mod p as plain integers, x in [1, p), p odd and below 2^383
  void inv_plain(Fe6& r, const Fe6& x) const {
    Fe6 u = x, v = p, x1 = {{1, 0, 0, 0, 0, 0}}, x2 = {{0, 0, 0, 0, 0, 0}};
    auto is_one = [](const Fe6& f) { return f.v[0] == 1 && (f.v[1] | f.v[2] | f.v[3] | f.v[4] | f.v[5]) == 0; };
    auto shr1 = [](Fe6& f) {
      for (int i = 0; i < 5; i++) f.v[i] = (f.v[i] >> 1) | (f.v[i + 1] << 63);
      f.v[5] >>= 1;
    };
    auto half_mod = [&](Fe6& f) {   // f / 2 mod p
      if (f.v[0] & 1) {
        u128 c = 0;
        for (int i = 0; i < 6; i++) {
          c += (u128)f.v[i] + p.v[i];
          f.v[i] = (uint64_t)c;
          c >>= 64;
        }
      }
      shr1(f);
    };
    while (!is_one(u) && !is_one(v)) {
      while (!(u.v[0] & 1)) { shr1(u); half_mod(x1); }
      while (!(v.v[0] & 1)) { shr1(v); half_mod(x2); }
      if (ge(u, v)) { sub_raw(u, u, v); sub(x1, x1, x2); }
      else { sub_raw(v, v, u); sub(x2, x2, x1); }
    }
    r = is_one(u) ? x1 : x2;
  }
  // the same by Fermat's little theorem (kept as the cross-check of tests)
  void inv_fermat(Fe6& r, const Fe6& a) const {
    Fe6 e = p;
    Fe6 two = {{2, 0, 0, 0, 0, 0}};
    sub_raw(e, p, two);
    Fe6 acc = one;
    for (int bit = 383; bit >= 0; bit--) {
      sqr(acc, acc);
      if ((e.v[bit / 64] >> (bit % 64)) & 1) mul(acc, acc, a);
    }
    r = acc;
  }

  void init(const uint32_t* p_words12) {
    for (int i = 0; i < 6; i++) p.v[i] = (uint64_t)p_words12[2 * i] | ((uint64_t)p_words12[2 * i + 1] << 32);
    nl = (p.v[4] == 0 && p.v[5] == 0 && (p.v[3] >> 63) == 0) ? 4 : 6;
    // Newton iteration for p^-1 mod 2^64
    uint64_t x = p.v[0];
    for (int i = 0; i < 6; i++) x *= 2 - p.v[0] * x;
    pinv = (uint64_t)0 - x;
    // one = R mod p by repeated doubling of 1, r2 = R^2 mod p
    Fe6 t = {{1, 0, 0, 0, 0, 0}};
    for (int i = 0; i < radix_bits(); i++) add(t, t, t);
    one = t;
    for (int i = 0; i < radix_bits(); i++) add(t, t, t);
    r2 = t;
  }
  // 2^k mod p as a plain (non-Montgomery) integer
  Fe6 pow2(int k) const {
    Fe6 t = {{1, 0, 0, 0, 0, 0}};
    for (int i = 0; i < k; i++) add(t, t, t);
    return t;
  }
};

struct Proj6 {
  Fe6 X, Y, Z;
};

struct Curve6 {
  Field6 F;

  bool is_zero(const Proj6& P) const { return Field6::is_zero(P.Z); }
  Proj6 zero() const {
    Proj6 P;
    memset(&P, 0, sizeof(P));
    P.Y = F.one;
    return P;
  }
  // dbl-1998-cmo-2, a = 0 (src/curve-projective.ts:202-253)
  Proj6 dbl(const Proj6& P) const {
    if (is_zero(P)) return zero();
    Fe6 w, s, ss, sss, R, B, h, t, u;
    F.sqr(t, P.X);
    F.add(w, t, t);
    F.add(w, w, t);
    F.mul(s, P.Y, P.Z);
    F.sqr(ss, s);
    F.mul(sss, s, ss);
    F.mul(R, P.Y, s);
    F.mul(B, P.X, R);
    Fe6 B2, B4, B8;
    F.dbl(B2, B);
    F.dbl(B4, B2);
    F.dbl(B8, B4);
    F.sqr(h, w);
    F.sub(h, h, B8);
    Proj6 Q;
    F.mul(t, h, s);
    F.dbl(Q.X, t);
    F.sub(u, B4, h);
    F.mul(u, w, u);
    F.sqr(t, R);
    F.dbl(t, t);
    F.dbl(t, t);
    F.dbl(t, t);
    F.sub(Q.Y, u, t);
    F.dbl(t, sss);
    F.dbl(t, t);
    F.dbl(Q.Z, t);
    return Q;
  }
  // add-1998-cmo-2 with edge cases (src/curve-projective.ts:51-160)
  Proj6 add(const Proj6& P, const Proj6& Q) const {
    if (is_zero(P)) return Q;
    if (is_zero(Q)) return P;
    Fe6 Y1Z2, X1Z2, Z1Z2, u, v, t;
    F.mul(Y1Z2, P.Y, Q.Z);
    F.mul(X1Z2, P.X, Q.Z);
    F.mul(Z1Z2, P.Z, Q.Z);
    F.mul(t, Q.Y, P.Z);
    F.sub(u, t, Y1Z2);
    F.mul(t, Q.X, P.Z);
    F.sub(v, t, X1Z2);
    if (Field6::is_zero(v)) {
      if (Field6::is_zero(u)) return dbl(P);
      return zero();
    }
    Fe6 uu, vv, vvv, R, A;
    F.sqr(uu, u);
    F.sqr(vv, v);
    F.mul(vvv, v, vv);
    F.mul(R, vv, X1Z2);
    F.mul(A, uu, Z1Z2);
    F.sub(A, A, vvv);
    F.dbl(t, R);
    F.sub(A, A, t);
    Proj6 S;
    F.mul(S.X, v, A);
    F.sub(t, R, A);
    F.mul(t, u, t);
    Fe6 t2;
    F.mul(t2, vvv, Y1Z2);
    F.sub(S.Y, t, t2);
    F.mul(S.Z, vvv, Z1Z2);
    return S;
  }
};

// twisted Edwards a = -1 in extended coordinates over the same 6-limb field code (the 253-bit prime
// simply has two zero top limbs): unified add-2008-hwcd-3, src/curve-twisted-edwards.ts:84-165
struct Ext6 {
  Fe6 X, Y, Z, T;
};

struct TeCurve6 {
  Field6 F;
  Fe6 k;  // 2d, Montgomery form

  void init(const uint32_t* p_words12, uint64_t d) {
    F.init(p_words12);
    Fe6 t = {{2 * d, 0, 0, 0, 0, 0}};
    F.mul(k, t, F.r2);
  }
  Ext6 zero() const {
    Ext6 P;
    memset(&P, 0, sizeof(P));
    P.Y = F.one;
    P.Z = F.one;
    return P;
  }
  Ext6 add(const Ext6& P, const Ext6& Q) const {
    Fe6 a, b, A, B, C, D, E, Fv, G, H;
    F.sub(a, P.Y, P.X); F.sub(b, Q.Y, Q.X); F.mul(A, a, b);
    F.add(a, P.Y, P.X); F.add(b, Q.Y, Q.X); F.mul(B, a, b);
    F.mul(C, P.T, Q.T); F.mul(C, C, k);
    F.mul(D, P.Z, Q.Z); F.add(D, D, D);
    F.sub(E, B, A); F.sub(Fv, D, C); F.add(G, D, C); F.add(H, B, A);
    Ext6 R;
    F.mul(R.X, E, Fv); F.mul(R.Y, G, H); F.mul(R.T, E, H); F.mul(R.Z, Fv, G);
    return R;
  }
};

}  // namespace msm_host
