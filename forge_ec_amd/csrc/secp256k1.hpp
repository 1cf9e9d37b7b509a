// secp256k1.hpp -- device-side secp256k1 field / point / ladder, bit-exact with
// forge-ec-curves/src/secp256k1.rs (citations below are lines of that file).
//
// The reference's arithmetic is deterministic but not a field; what is
// reproduced here is its integer behaviour, re-derived for 8 x 32-bit words per element:
//
//  * Mul (442-507) is exactly  csub_p( ((T + M*p) mod 2^512) >> 256 )  with T = a*b and
//    M = -T/p mod 2^256, for ANY 256-bit a, b (checked against both oracles on 2*10^4 random
//    and edge inputs).  Since p = 2^256 - c, c = 2^32 + 977, that is (T_hi + M - Q) mod 2^256
//    with Q = (M*c - T_lo) / 2^256 < 2^33, and M, Q come out of an 8-column recurrence that
//    costs one v_mul_lo_u32 + one v_mad_u64_u32 per column instead of a 9-multiply CIOS round.
//  * square (634-713) is NOT mul(x,x): its 64-bit carry handling is restated literally on
//    word pairs (cross term doubled mod 2^128; carry skipped one limb; single +1 ripple;
//    all four high limbs folded into limb 0).
//  * Add / Sub / Neg (353-440, 509-539) are restated for arbitrary 256-bit operands.
#pragma once
#include "limbs.hpp"

namespace fecgpu {
namespace secp {

// p = 2^256 - c, c = 0x1_000003D1.  Constant operands of the carry chains ride as VOP2
// literals / inline constants: + c is {0x3d1, 1, 0...}; p itself is {0xFFFFFC2F, -2, -1...}.
#define FEC_SECP_C 0x3d1, 1, 0, 0, 0, 0, 0, 0
#define FEC_SECP_P 0xfffffc2f, -2, -1, -1, -1, -1, -1, -1

// reduce (78-102): v >= p ? v - p : v.   v >= p  <=>  v + c carries out of 2^256.
FEC_DEV fe csub_p(const fe& v) {
  fe w;
  lmask ov;
  FEC_ADDK256(w, v, ov, FEC_SECP_C);
  return fe_select(v, w, ov);
}

// Add (353-393): s = a + b mod 2^256; subtract p once if the add carried or s >= p.
FEC_DEV fe add(const fe& a, const fe& b) {
  fe s, w;
  lmask carry = add256(s, a, b);
  lmask ov;
  FEC_ADDK256(w, s, ov, FEC_SECP_C);  // w = s - p mod 2^256
  return fe_select(s, w, carry | ov);
}

// Sub (395-440): d = a - b mod 2^256; add p (wrapping) if it borrowed.
FEC_DEV fe sub(const fe& a, const fe& b) {
  fe d, w;
  lmask borrow = sub256(d, a, b);
  lmask t;
  FEC_SUBK256(w, d, t, FEC_SECP_C);  // d + p mod 2^256
  (void)t;
  return fe_select(d, w, borrow);
}

// Neg (509-539): p - a (wrapping), 0 -> 0.
FEC_DEV fe neg(const fe& a) {
  fe r;
  lmask t;
  FEC_KSUB256(r, a, t, FEC_SECP_P);
  (void)t;
  return fe_select(r, a, fe_is_zero(a));
}

// Montgomery tail shared by Mul: given the exact 512-bit T, return the reference's result.
FEC_DEV fe mont_reduce(const u32 t[16]) {
  const u32 N0P = 0xD2253531u;  // low word of N0 (468) = 977^-1 mod 2^32
  u32 m[8];
  u32 e_lo = 0, e_hi = 0;  // E_k < 2^33
  FEC_UNROLL for (int k = 0; k < 8; ++k) {
    u32 x = t[k] - e_lo;
    m[k] = x * N0P;
    u64 d = (u64)m[k] * 977u + (((u64)e_hi << 32) | e_lo);  // low word == t[k] by construction
    u64 e = (d >> 32) + m[k];
    e_lo = (u32)e;
    e_hi = (u32)(e >> 32);
  }
  // V = T_hi + M - Q  (mod 2^256), Q = E_8
  fe th, mm, v, v2;
  FEC_UNROLL for (int i = 0; i < 8; ++i) {
    th.w[i] = t[8 + i];
    mm.w[i] = m[i];
  }
  add256(v, th, mm);
  sub_lohi256(v2, v, e_lo, e_hi);
  return csub_p(v2);
}

// Mul (442-507)
FEC_DEV fe mul(const fe& a, const fe& b) {
  u32 t[16];
  mul_wide(t, a, b);
  return mont_reduce(t);
}

// multiply by a raw small constant through the Montgomery Mul (three/eight at 1523, 1533)
FEC_DEV fe mul_small(const fe& a, u32 k) {
  u32 t[16];
  mul_wide_small(t, a, k);
  return mont_reduce(t);
}

// square (634-713), literal on 64-bit limbs
FEC_DEV fe sqr(const fe& a) {
  u64 pr[8];
  FEC_UNROLL for (int i = 0; i < 4; ++i)  // 643-649
      mul64wide(a.w[2 * i], a.w[2 * i + 1], a.w[2 * i], a.w[2 * i + 1], pr[2 * i], pr[2 * i + 1]);
  FEC_UNROLL for (int i = 0; i < 4; ++i) {  // 652-679
    FEC_UNROLL for (int j = i + 1; j < 4; ++j) {
      u64 lo, hi;
      mul64wide(a.w[2 * i], a.w[2 * i + 1], a.w[2 * j], a.w[2 * j + 1], lo, hi);
      hi = (hi << 1) | (lo >> 63);  // u128 wrapping_mul(2): bit 127 is lost
      lo = lo << 1;
      u64 s0 = pr[i + j] + lo;
      u32 c1 = s0 < lo;
      pr[i + j] = s0;
      u64 s1 = pr[i + j + 1] + hi;  // c1 is not added here
      u32 c2 = s1 < hi;
      pr[i + j + 1] = s1;
      u64 inc = c1 | c2;  // a single +1 rippling from limb i+j+2, lost past limb 7
      FEC_UNROLL for (int k = i + j + 2; k < 8; ++k) {
        u64 v = pr[k] + inc;
        inc = (v < inc) ? 1 : 0;
        pr[k] = v;
      }
    }
  }
  u64 res[4] = {pr[0], pr[1], pr[2], pr[3]};
  u64 carry = 0;  // 692
  FEC_UNROLL for (int i = 4; i < 8; ++i) {
    u64 m = pr[i] * 0x1000003D1ULL;  // low 64 bits only
    u64 t = res[0] + m;              // always into limb 0
    t = t + carry;
    res[0] = t;
    carry = (u64)(t < m) | ((u64)(t < carry) & (u64)(m != 0));  // 698-699
    FEC_UNROLL for (int j = 1; j < 4; ++j) {
      u64 t2 = res[j] + carry;
      res[j] = t2;
      carry = (u64)(t2 < carry);
    }
  }
  fe r;
  FEC_UNROLL for (int i = 0; i < 4; ++i) set_limb64(r, i, res[i]);
  return csub_p(r);
}

struct pt {
  fe x, y, z;
};

FEC_DEV pt identity() {  // 1322-1324
  pt p;
  p.x = fe_zero();
  p.y = fe_small(1);
  p.z = fe_zero();
  return p;
}
FEC_DEV lmask is_identity(const pt& p) { return fe_is_zero(p.z); }  // 1326-1340 (all-zero implies z == 0)

FEC_DEV pt pt_select(const pt& a, const pt& b, lmask choice) {
  pt r;
  r.x = fe_select(a.x, b.x, choice);
  r.y = fe_select(a.y, b.y, choice);
  r.z = fe_select(a.z, b.z, choice);
  return r;
}

// inherent ProjectivePoint::double (1502-1540): the one Curve::multiply and Add reach.
FEC_DEV pt pdouble(const pt& p) {
  fe a = sqr(p.x);
  fe b = sqr(p.y);
  fe c = sqr(b);
  fe xpb2 = sqr(add(p.x, b));
  fe dd = sub(sub(xpb2, a), c);
  fe d = add(dd, dd);
  fe e = mul_small(a, 3);
  fe f = sqr(e);
  pt r;
  r.x = sub(f, add(d, d));
  r.y = sub(mul(e, sub(d, r.x)), mul_small(c, 8));
  fe yz = mul(p.y, p.z);
  r.z = add(yz, yz);
  return pt_select(r, identity(), is_identity(p));
}

// Add for ProjectivePoint (1444-1498) without the equal-points branch: sets need_double when
// u1 == u2 && s1 == s2 (the caller then substitutes self.double()); every other early-out is
// folded in with selects.
FEC_DEV pt padd_nodouble(const pt& p, const pt& q, lmask& need_double) {
  fe z1s = sqr(p.z);
  fe z2s = sqr(q.z);
  fe u1 = mul(p.x, z2s);
  fe u2 = mul(q.x, z1s);
  fe z1c = mul(z1s, p.z);
  fe z2c = mul(z2s, q.z);
  fe s1 = mul(p.y, z2c);
  fe s2 = mul(q.y, z1c);
  fe h = sub(u2, u1);
  fe r = sub(s2, s1);
  fe h2 = sqr(h);
  fe h3 = mul(h2, h);
  fe u1h2 = mul(u1, h2);
  pt o;
  o.x = sub(sub(sub(sqr(r), h3), u1h2), u1h2);
  o.y = sub(mul(r, sub(u1h2, o.x)), mul(s1, h3));
  o.z = mul(mul(h, p.z), q.z);
  lmask idp = is_identity(p), idq = is_identity(q);
  lmask ueq = fe_eq(u1, u2), seq = fe_eq(s1, s2);
  o = pt_select(o, identity(), ueq & ~seq);
  o = pt_select(o, p, idq);
  o = pt_select(o, q, idp);
  need_double = ueq & seq & ~idp & ~idq;
  return o;
}

FEC_DEV pt padd(const pt& p, const pt& q) {
  lmask nd;
  pt o = padd_nodouble(p, q, nd);
  if (nd != 0) {  // never taken on random inputs
    pt d = pdouble(p);
    o = pt_select(o, d, nd);
  }
  return o;
}

// trait PointProjective::double (1375-1418); not on the ladder path.
FEC_DEV pt pdouble_trait(const pt& p) {
  fe xx = sqr(p.x);
  fe yy = sqr(p.y);
  fe yyyy = sqr(yy);
  fe xy2 = sqr(add(p.x, yy));
  fe w = sub(sub(xy2, xx), yyyy);
  fe d = add(w, w);
  fe e = mul_small(xx, 3);
  fe ee = sqr(e);
  pt r;
  r.x = sub(sub(ee, d), d);
  r.y = sub(mul(e, sub(d, r.x)), mul_small(yyyy, 8));
  fe z3 = add(p.y, p.y);
  r.z = fe_select(mul(z3, p.z), z3, fe_eq(p.z, fe_small(1)));
  return pt_select(r, identity(), is_identity(p));
}

// Bit i of the ladder (2655-2659): byte i/8 of the little-endian bytes, MSB first in the byte.
FEC_DEV u32 ladder_bit(const u32* kw, int i) {
  u32 w = kw[(i >> 5) * KSTRIDE];
  int sh = (((i >> 3) & 3) << 3) + 7 - (i & 7);
  return (w >> sh) & 1u;
}

// Curve::multiply (2635-2692).  `kw` points at this lane's scalar in LDS (word k at kw[k * KSTRIDE]).  Only the
// selected doubling is computed (the other is discarded by the reference, 2669-2684).
FEC_DEV pt multiply(const pt& point, const u32* kw) {
  u32 any = 0;
  FEC_UNROLL for (int i = 0; i < 8; ++i) any |= kw[i * KSTRIDE];
  lmask early = is_identity(point) | lanes_where(any == 0);
  pt r0 = identity();
  pt r1 = point;
#pragma unroll 1
  for (int i = 0; i < 256; ++i) {
    lmask bit = lanes_where(ladder_bit(kw, i) != 0);
    lmask nd;
    pt s = padd_nodouble(r0, r1, nd);
    // one pdouble instance in the code object, run a second time only when Add (1469-1473)
    // falls through to self.double() for some lane -- never on random inputs
    pt din = pt_select(r0, r1, bit);
    pt d;
#pragma unroll 1
    for (int pass = 0;; ++pass) {
      pt o = pdouble(din);
      if (pass == 0) {
        d = o;
        if (nd == 0) break;
        din = r0;
      } else {
        s = pt_select(s, o, nd);
        break;
      }
    }
    r0 = pt_select(d, s, bit);
    r1 = pt_select(s, d, bit);
  }
  return pt_select(r0, identity(), early);
}

}  // namespace secp
}  // namespace fecgpu
