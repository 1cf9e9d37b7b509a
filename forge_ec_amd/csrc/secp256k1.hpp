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
// lanes where v >= p is possible at all: p = 2^256 - 2^32 - 977 has words 2..7 all ones
FEC_DEV lmask maybe_ge_p(const fe& v) {
  u32 ones = v.w[2] & v.w[3] & v.w[4] & v.w[5] & v.w[6] & v.w[7];
  return lanes_where(ones == 0xFFFFFFFFu);
}
// the same reduce where v >= p is improbable (2^-224 for a Mul/square result): the chain and the
// select sit behind a wave-uniform branch
FEC_DEV fe csub_p_unlikely(const fe& v) {
  if (__builtin_expect(maybe_ge_p(v) != 0, 0)) return csub_p(v);
  return v;
}

// Add (353-393): s = a + b mod 2^256; subtract p once -- i.e. add c = 2^32 + 977 mod 2^256 -- if the add carried or
// s >= p.  s >= p needs a top word of all ones (2^-32 per lane): such wavefronts take the literal form.  Otherwise
// the condition is the carry alone, and c touches words 0..1 (short chain, limbs.hpp).
FEC_DEV fe add(const fe& a, const fe& b) {
  fe s;
  const lmask carry = add256(s, a, b);
  if (__builtin_expect(lanes_where(s.w[7] == 0xFFFFFFFFu) != 0, 0)) {
    fe w;
    lmask ov;
    FEC_ADDK256(w, s, ov, FEC_SECP_C);  // w = s - p mod 2^256
    return fe_select(s, w, carry | ov);
  }
  lmask cy;
  FEC_ADD_SHORT2(s, carry, cy, 0x3d1);
  if (__builtin_expect(cy != 0, 0)) carry_from<2>(s, cy);
  return s;
}

// a + a through Add (353-393) -- FieldElement::double (105-109) is `s + s` -- with one operand list (dbl256)
FEC_DEV fe dbl(const fe& a) {
  fe s;
  const lmask carry = dbl256(s, a);
  if (__builtin_expect(lanes_where(s.w[7] == 0xFFFFFFFFu) != 0, 0)) {
    fe w;
    lmask ov;
    FEC_ADDK256(w, s, ov, FEC_SECP_C);  // w = s - p mod 2^256
    return fe_select(s, w, carry | ov);
  }
  lmask cy;
  FEC_ADD_SHORT2(s, carry, cy, 0x3d1);
  if (__builtin_expect(cy != 0, 0)) carry_from<2>(s, cy);
  return s;
}

// Sub (395-440): d = a - b mod 2^256; add p (wrapping), i.e. subtract c, if it borrowed.
FEC_DEV fe sub(const fe& a, const fe& b) {
  fe d;
  const lmask borrow = sub256(d, a, b);
  lmask bw;
  FEC_SUB_SHORT2(d, borrow, bw, 0x3d1);
  if (__builtin_expect(bw != 0, 0)) borrow_from<2>(d, bw);
  return d;
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
  return csub_p_unlikely(v2);
}

// multiply by a raw small constant through the Montgomery Mul (three/eight at 1523, 1533)
FEC_DEV fe mul_small_cxx(const fe& a, u32 k) {
  u32 t[16];
  mul_wide_small(t, a, k);
  return mont_reduce(t);
}

// Mul (442-507), compiler-scheduled form: the host emulation's Mul and the cross-check of the
// hand-allocated one below
FEC_DEV fe mul_cxx(const fe& a, const fe& b) {
  u32 t[16];
  mul_wide(t, a, b);
  return mont_reduce(t);
}

#ifdef FEC_HOST_EMUL
FEC_DEV fe mul(const fe& a, const fe& b) { return mul_cxx(a, b); }
#else
// v -= 2^64 on the lanes of bw: the borrow out of word 1 continued through words 2..7
FEC_DEV fe borrow_from_word2(const fe& v, lmask bw) {
  fe x = v;
  asm("s_mov_b64 vcc, %6\n\t"
      "v_subbrev_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
      "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_subbrev_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
      "v_subbrev_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
      "v_subbrev_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
      "v_subbrev_co_u32_e32 %5, vcc, 0, %5, vcc"
      : "+v"(x.w[2]), "+v"(x.w[3]), "+v"(x.w[4]), "+v"(x.w[5]), "+v"(x.w[6]), "+v"(x.w[7])
      : "s"(bw)
      : "vcc");
  return x;
}
// reduce (78-102) of a Mul / square result: v >= p needs words 2..7 all ones, so the whole test sits
// behind one compare of the top word (2^-32 per lane)
FEC_DEV fe csub_p_top(const fe& v) {
  if (__builtin_expect(lanes_where(v.w[7] == 0xFFFFFFFFu) != 0, 0)) return csub_p_unlikely(v);
  return v;
}
// Mul (442-507) as ONE hand-allocated asm statement (tools/gen_field_asm.py): the 512-bit product by
// product scanning with the Montgomery word recurrence of mont_reduce() interleaved, then
// V = T_hi + M - Q.  Q < 2^33 is subtracted from words 0..1 only; the borrow out of word 1
// (2^-31 per lane) is exported as a lane mask and continued behind a wave-uniform branch.
FEC_DEV fe mul(const fe& a, const fe& b) {
  fe r;
  lmask sc, bw;
  asm(FEC_SECP_MUL_ASM
      : "=v"(r.w[0]), "=v"(r.w[1]), "=v"(r.w[2]), "=v"(r.w[3]), "=v"(r.w[4]), "=v"(r.w[5]), "=v"(r.w[6]),
        "=v"(r.w[7]), "=&s"(sc), "=&s"(bw)
      : FEC_V8(a), FEC_V8(b), "s"(0xD2253531u), "s"(977u)
      : FEC_SECP_MUL_CLOBBERS);
  if (__builtin_expect(bw != 0, 0)) r = borrow_from_word2(r, bw);
  return csub_p_top(r);
}
// Mul by raw 3 / raw 8 (1523, 1533): the nine-word product a*k and the same recurrence, one asm statement
template <u32 K>
FEC_DEV fe mul_small_k(const fe& a) {
  static_assert(K == 3 || K == 8, "the ladder multiplies by 3 and 8 only");
  fe r;
  lmask sc, bw, exc;
  if (K == 3) {
    asm(FEC_SECP_MUL3_ASM
        : "=v"(r.w[0]), "=v"(r.w[1]), "=v"(r.w[2]), "=v"(r.w[3]), "=v"(r.w[4]), "=v"(r.w[5]), "=v"(r.w[6]),
          "=v"(r.w[7]), "=&s"(sc), "=&s"(bw), "=&s"(exc)
        : FEC_V8(a), "s"(0xD2253531u), "s"(977u)
        : FEC_SECP_MUL3_CLOBBERS);
  } else {
    asm(FEC_SECP_MUL8_ASM
        : "=v"(r.w[0]), "=v"(r.w[1]), "=v"(r.w[2]), "=v"(r.w[3]), "=v"(r.w[4]), "=v"(r.w[5]), "=v"(r.w[6]),
          "=v"(r.w[7]), "=&s"(sc), "=&s"(bw), "=&s"(exc)
        : FEC_V8(a), "s"(0xD2253531u), "s"(977u)
        : FEC_SECP_MUL8_CLOBBERS);
  }
  // the rare legs are selected INTO r: a leg that returned its own registers would cost the common leg eight v_mov at
  // the join
  if (__builtin_expect(exc != 0, 0)) r = fe_select(r, mul_small_cxx(a, K), ~(lmask)0);  // m0 + t8 carried out of word 0
  else if (__builtin_expect(bw != 0, 0)) r = borrow_from_word2(r, bw);
  return csub_p_top(r);
}
#endif
#ifdef FEC_HOST_EMUL
FEC_DEV fe mul_small(const fe& a, u32 k) { return mul_small_cxx(a, k); }
#else
FEC_DEV fe mul_small(const fe& a, u32 k) { return k == 3 ? mul_small_k<3>(a) : (k == 8 ? mul_small_k<8>(a) : mul_small_cxx(a, k)); }
#endif


// ---- square() (634-713) ---------------------------------------------------------------------
// NOT Montgomery and NOT mul(x,x).  Restated on 16 words w[] (the reference's product[0..8]):
//   * limb squares into w (643-649);
//   * each cross term a_i*a_j, doubled mod 2^128 (bit 127 lost), is added as TWO independent
//     64-bit adds at limbs i+j and i+j+1 -- the first add's carry is not fed to the second -- and
//     if either carried a single +1 ripples from limb i+j+2, dropped past limb 7 (652-679);
//   * every high limb h is folded into limb 0 (always limb 0) as the low 64 bits of
//     h * 0x1000003D1, with the reference's carry rule (693-707), then one conditional subtract.
// The data-dependent ripples almost never travel (a +1 leaves a limb only if it was 2^64-1), so
// each ripple is performed into its first limb and continued under a wave-uniform unlikely branch;
// likewise the fold's incoming carry (692) is 0 unless a ripple crossed limbs 1..3, in which case
// that fold step runs the reference's general rule.  Every branch is exact; none is taken on
// random data (tests/golden/secp256k1_sqr_ripple_operands.json forces them).
#ifdef FEC_HOST_EMUL
static unsigned long fec_host_rare_sqr = 0;  // coverage counter for tests (host emulation only)
FEC_DEV lmask cross_add(u32& w0, u32& w1, u32& w2, u32& w3, u32& w4, u32& w5, u32 x0, u32 x1, u32 x2, u32 x3) {
  u64 lo = ((u64)w1 << 32) | w0, hi = ((u64)w3 << 32) | w2, nx = ((u64)w5 << 32) | w4;
  u64 xl = ((u64)x1 << 32) | x0, xh = ((u64)x3 << 32) | x2;
  u64 s0 = lo + xl, s1 = hi + xh;
  u64 inc = (s0 < xl) | (s1 < xh);
  u64 n2 = nx + inc;
  w0 = (u32)s0; w1 = (u32)(s0 >> 32); w2 = (u32)s1; w3 = (u32)(s1 >> 32); w4 = (u32)n2; w5 = (u32)(n2 >> 32);
  return (n2 < inc) ? ~0ull : 0ull;
}
FEC_DEV lmask fold_add(u32& r0, u32& r1, u32& r2, u32& r3, u32 m0, u32 m1) {
  u64 l0 = ((u64)r1 << 32) | r0, l1 = ((u64)r3 << 32) | r2, m = ((u64)m1 << 32) | m0;
  u64 t = l0 + m;
  u64 c = t < m;
  u64 t1 = l1 + c;
  r0 = (u32)t; r1 = (u32)(t >> 32); r2 = (u32)t1; r3 = (u32)(t1 >> 32);
  return (t1 < c) ? ~0ull : 0ull;
}
FEC_DEV lmask ripple2(u32& a, u32& b, lmask cin) {
  u64 v = ((u64)b << 32) | a, c = cin ? 1 : 0;
  u64 n = v + c;
  a = (u32)n; b = (u32)(n >> 32);
  return (n < c) ? ~0ull : 0ull;
}
#else
// words w0..w3 += (x0..x3) as two independent 64-bit adds; +1 (if either carried) into limb
// (w4,w5); returns the lanes where that limb carried out.
FEC_DEV lmask cross_add(u32& w0, u32& w1, u32& w2, u32& w3, u32& w4, u32& w5, u32 x0, u32 x1, u32 x2, u32 x3) {
  lmask more;
  asm("v_add_co_u32_e32 %0, vcc, %0, %7\n\t"
      "v_addc_co_u32_e32 %1, vcc, %1, %8, vcc\n\t"
      "s_mov_b64 %6, vcc\n\t"
      "v_add_co_u32_e32 %2, vcc, %2, %9\n\t"
      "v_addc_co_u32_e32 %3, vcc, %3, %10, vcc\n\t"
      "s_or_b64 vcc, vcc, %6\n\t"
      "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
      "v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
      "s_mov_b64 %6, vcc"
      : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "=&s"(more)
      : "v"(x0), "v"(x1), "v"(x2), "v"(x3)
      : "vcc");
  return more;
}
// limb0 (r0,r1) += (m1:m0); its carry-out goes into limb1 (r2,r3); returns limb1's carry-out.
FEC_DEV lmask fold_add(u32& r0, u32& r1, u32& r2, u32& r3, u32 m0, u32 m1) {
  lmask more;
  asm("v_add_co_u32_e32 %0, vcc, %0, %5\n\t"
      "v_addc_co_u32_e32 %1, vcc, %1, %6, vcc\n\t"
      "v_addc_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
      "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
      "s_mov_b64 %4, vcc"
      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=s"(more)
      : "v"(m0), "v"(m1)
      : "vcc");
  return more;
}
// limb (a,b) += cin (one bit per lane); returns its carry-out.
FEC_DEV lmask ripple2(u32& a, u32& b, lmask cin) {
  lmask cout;
  asm("s_mov_b64 vcc, %3\n\t"
      "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "s_mov_b64 %2, vcc"
      : "+v"(a), "+v"(b), "=s"(cout)
      : "s"(cin)
      : "vcc");
  return cout;
}
#endif

// One fold step in the reference's full generality (693-707): r[0..7] are the words of result[0..4],
// cin the per-lane incoming carry.  Returns the carry out of limb 3.
FEC_DEV lmask fold_general(u32 r[8], u32 m0, u32 m1, lmask cin) {
  u64 m = ((u64)m1 << 32) | m0;
  u64 carry = word_select(0u, 1u, cin);
  u64 res0 = ((u64)r[1] << 32) | r[0];
  u64 t = res0 + m;
  t = t + carry;
  carry = (u64)(t < m) | ((u64)(t < carry) & (u64)(m != 0));  // 698-699
  r[0] = (u32)t;
  r[1] = (u32)(t >> 32);
  lmask c = lanes_where(carry != 0);
  FEC_UNROLL for (int j = 1; j < 4; ++j) c = ripple2(r[2 * j], r[2 * j + 1], c);  // 702-706
  return c;
}

FEC_DEV fe sqr_cxx(const fe& a) {
  u32 w[16];
  FEC_UNROLL for (int i = 0; i < 4; ++i)  // 643-649: limb squares
      mul64_words(&w[4 * i], a.w[2 * i], a.w[2 * i + 1], a.w[2 * i], a.w[2 * i + 1]);
  FEC_UNROLL for (int i = 0; i < 4; ++i) {  // 652-679: doubled cross terms
    FEC_UNROLL for (int j = i + 1; j < 4; ++j) {
      u32 p[4];
      mul64_words(p, a.w[2 * i], a.w[2 * i + 1], a.w[2 * j], a.w[2 * j + 1]);
      u32 x0 = p[0] << 1;  // u128 wrapping_mul(2): bit 127 is lost
      u32 x1 = (p[1] << 1) | (p[0] >> 31);
      u32 x2 = (p[2] << 1) | (p[1] >> 31);
      u32 x3 = (p[3] << 1) | (p[2] >> 31);
      const int B = 2 * (i + j);
      lmask more = cross_add(w[B], w[B + 1], w[B + 2], w[B + 3], w[B + 4], w[B + 5], x0, x1, x2, x3);
      if (B + 6 < 16) {  // the ripple may continue into limbs i+j+3 .. 7
        if (__builtin_expect(more != 0, 0)) {
#ifdef FEC_HOST_EMUL
          ++fec_host_rare_sqr;
#endif
          FEC_UNROLL for (int k = B + 6; k < 16; k += 2) more = ripple2(w[k], w[k + 1], more);
        }
      }
    }
  }
  // 681-707: every high limb folded into limb 0 with the low 64 bits of limb * 0x1000003D1
  u32 r[8];
  FEC_UNROLL for (int i = 0; i < 8; ++i) r[i] = w[i];
  lmask carry = 0;  // 692: carry out of limb 3 of the previous fold step
  FEC_UNROLL for (int i = 4; i < 8; ++i) {
    u32 h0 = w[2 * i], h1 = w[2 * i + 1];
    u64 q = (u64)h0 * 977u;
    u32 m0 = (u32)q;
    u32 m1 = (u32)(q >> 32) + h1 * 977u + h0;
    if (__builtin_expect(carry == 0, 1)) {
      // incoming carry 0: the rule at 698-699 is the plain carry-out of limb0 + m
      lmask more = fold_add(r[0], r[1], r[2], r[3], m0, m1);
      if (__builtin_expect(more != 0, 0)) {
#ifdef FEC_HOST_EMUL
        ++fec_host_rare_sqr;
#endif
        more = ripple2(r[4], r[5], more);
        carry = ripple2(r[6], r[7], more);
      }
    } else {
#ifdef FEC_HOST_EMUL
      ++fec_host_rare_sqr;
#endif
      carry = fold_general(r, m0, m1, carry);
    }
  }
  fe o;
  FEC_UNROLL for (int i = 0; i < 8; ++i) o.w[i] = r[i];
  return csub_p_unlikely(o);
}

#ifdef FEC_HOST_EMUL
FEC_DEV fe sqr(const fe& a) { return sqr_cxx(a); }
#else
// square() as ONE hand-allocated asm statement (tools/gen_field_asm.py) for the path on which no +1
// ripples past the limb it is added to (each continuation needs a 64-bit limb of all ones).  The
// lanes where one would are returned as a mask, and such a wavefront recomputes with sqr_cxx().
FEC_DEV fe sqr(const fe& a) {
  fe r;
  lmask tmp, exc;
  asm(FEC_SECP_SQR_ASM
      : "=v"(r.w[0]), "=v"(r.w[1]), "=v"(r.w[2]), "=v"(r.w[3]), "=&v"(r.w[4]), "=&v"(r.w[5]), "=&v"(r.w[6]),
        "=&v"(r.w[7]), "=&s"(tmp), "=&s"(exc)
      : FEC_V8(a), "s"(977u)
      : FEC_SECP_SQR_CLOBBERS);
  if (__builtin_expect(exc != 0, 0)) r = fe_select(r, sqr_cxx(a), ~(lmask)0);
  // (the rare leg is selected INTO r: a leg that returned its own registers would cost the common leg eight v_mov at the join)
  return csub_p_top(r);
}
#endif

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
  fe d = dbl(dd);
  fe e = mul_small(a, 3);
  fe f = sqr(e);
  pt r;
  r.x = sub(f, dbl(d));
  r.y = sub(mul(e, sub(d, r.x)), mul_small(c, 8));
  fe yz = mul(p.y, p.z);
  r.z = dbl(yz);
  lmask idp = is_identity(p);
  if (__builtin_expect(idp != 0, 0)) r = pt_select(r, identity(), idp);
  return r;
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
  lmask ueq = fe_eq(u1, u2);
  need_double = 0;
  if (__builtin_expect((idp | idq | ueq) != 0, 0)) {  // early-outs: only the ladder's first steps
    lmask seq = fe_eq(s1, s2);
    o = pt_select(o, identity(), ueq & ~seq);
    o = pt_select(o, p, idq);
    o = pt_select(o, q, idp);
    need_double = ueq & seq & ~idp & ~idq;
  }
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

// ---- Add for ProjectivePoint (1444-1498) spread over FOUR lanes of one wavefront ----------------
// The ordered folds (Curve::multi_scalar_multiply, core lib.rs:944-948; schnorr::batch_verify 262-281)
// are a strictly sequential chain of additions: the reference's Add is neither associative nor
// commutative, so the order is part of the result and one addition cannot start before the previous
// one ends.  What CAN run in parallel is the inside of one addition: its 16 field multiplications /
// squarings form a dependency graph of depth 6 and width <= 4.  padd_coop() executes that graph level
// by level -- the same operations on the same operands as padd_nodouble(), so the sum is bit-identical
// -- with lanes 0..3 of the wavefront each taking one operation of the level.  Values travel between
// lanes through 27 eight-word slots of LDS (a wavefront's LDS accesses execute in program order, so no
// barrier is needed); the cheap field subtractions in between are done by every lane redundantly.
//   level 1 (square)   z1s = z1^2            z2s = z2^2
//   level 2 (Mul)      u1 = x1 z2s           u2 = x2 z1s          z1c = z1s z1      z2c = z2s z2
//   level 3 (Mul)      s1 = y1 z2c           s2 = y2 z1c                                  h = u2 - u1, r = s2 - s1
//   level 4 (square)   h2 = h^2              r2 = r^2
//   level 5 (Mul)      h3 = h2 h             u1h2 = u1 h2         hz = h z1               x3 = r2 - h3 - 2 u1h2
//   level 6 (Mul)      t = r (u1h2 - x3)     s1h3 = s1 h3         z3 = hz z2              y3 = t - s1h3
// 6 field-operation latencies per addition instead of 16.
namespace coop {
enum { PX = 0, PY, PZ, QX, QY, QZ, Z1S, Z2S, U1, U2, Z1C, Z2C, S1, S2, H, R, H2, R2, H3, U1H2, HZ, T, S1H3, Z3, X3, DD, ONE, SLOTS };
constexpr int WORDS = SLOTS * 8;
#ifdef FEC_HOST_EMUL
FEC_DEV fe ld(const u32* sh, int slot) {
  fe a;
  FEC_UNROLL for (int i = 0; i < 8; ++i) a.w[i] = sh[slot * 8 + i];
  return a;
}
FEC_DEV void st(u32* sh, int slot, const fe& a) {
  FEC_UNROLL for (int i = 0; i < 8; ++i) sh[slot * 8 + i] = a.w[i];
}
#else
// a slot is 32 bytes, 16-byte aligned (the caller's array is): two ds_read_b128 / ds_write_b128
FEC_DEV fe ld(const u32* sh, int slot) {
  const uint4* s4 = reinterpret_cast<const uint4*>(sh + slot * 8);
  const uint4 lo = s4[0], hi = s4[1];
  fe a;
  a.w[0] = lo.x; a.w[1] = lo.y; a.w[2] = lo.z; a.w[3] = lo.w;
  a.w[4] = hi.x; a.w[5] = hi.y; a.w[6] = hi.z; a.w[7] = hi.w;
  return a;
}
FEC_DEV void st(u32* sh, int slot, const fe& a) {
  uint4* s4 = reinterpret_cast<uint4*>(sh + slot * 8);
  s4[0] = make_uint4(a.w[0], a.w[1], a.w[2], a.w[3]);
  s4[1] = make_uint4(a.w[4], a.w[5], a.w[6], a.w[7]);
}
#endif
#ifdef FEC_HOST_EMUL
FEC_DEV void sync() {}
FEC_DEV int lane_id() { return 0; }
#else
// A wavefront's LDS instructions execute in program order, so a read issued after another lane's
// write (same wavefront, later instruction) returns the written data: no s_waitcnt is needed between
// them, only compiler ordering (the fences emit no instruction at wavefront scope).
FEC_DEV void sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
FEC_DEV int lane_id() { return (int)(threadIdx.x & 63); }
#endif
FEC_DEV int pick(int lane, int a0, int a1, int a2, int a3) { return lane == 0 ? a0 : (lane == 1 ? a1 : (lane == 2 ? a2 : (lane == 3 ? a3 : ONE))); }
// one level: lane l (< n) computes op(slot a_l, slot b_l) into slot o_l; the other lanes work on the constant 1
template <bool SQR>
FEC_DEV void level(u32* sh, int n, int a0, int a1, int a2, int a3, int b0, int b1, int b2, int b3, int o0, int o1, int o2, int o3) {
  const int lane = lane_id();
  const fe a = ld(sh, pick(lane, a0, a1, a2, a3));
  fe res;
  if (SQR) {
    res = sqr(a);
  } else {
    const fe b = ld(sh, pick(lane, b0, b1, b2, b3));
    res = mul(a, b);
  }
  // no level overwrites a slot it reads (outputs go to fresh slots), and a wavefront's LDS accesses
  // execute in program order, so one synchronisation per level -- after the stores -- is enough
  if (lane < n) st(sh, pick(lane, o0, o1, o2, o3), res);
  sync();
}
}  // namespace coop

// sh: coop::WORDS words of LDS owned by this wavefront; slots PX..QZ hold p and q on entry (written by
// the caller, followed by coop::sync()); slot ONE holds the constant 1.  Every lane returns the sum.
FEC_DEV pt padd_coop(u32* sh) {
  using namespace coop;
#ifdef FEC_HOST_EMUL
  const pt p0 = {ld(sh, PX), ld(sh, PY), ld(sh, PZ)}, q0 = {ld(sh, QX), ld(sh, QY), ld(sh, QZ)};
  return padd(p0, q0);
#else
  const int lane = lane_id();
  level<true>(sh, 2, PZ, QZ, ONE, ONE, ONE, ONE, ONE, ONE, Z1S, Z2S, ONE, ONE);
  level<false>(sh, 4, PX, QX, Z1S, Z2S, Z2S, Z1S, PZ, QZ, U1, U2, Z1C, Z2C);
  level<false>(sh, 2, PY, QY, ONE, ONE, Z2C, Z1C, ONE, ONE, S1, S2, ONE, ONE);
  {  // level 4 with its subtraction folded in: lane 0 h = u2 - u1, lane 1 r = s2 - s1, then the squares
    const fe d = sub(ld(sh, pick(lane, U2, S2, ONE, ONE)), ld(sh, pick(lane, U1, S1, ONE, ONE)));
    const fe d2 = sqr(d);
    if (lane < 2) {
      st(sh, pick(lane, H, R, ONE, ONE), d);
      st(sh, pick(lane, H2, R2, ONE, ONE), d2);
    }
    sync();
  }
  level<false>(sh, 3, H2, U1, H, ONE, H, H2, PZ, ONE, H3, U1H2, HZ, ONE);
  pt o;
  {  // level 6 with x3 and (u1h2 - x3) computed by every lane on the way in
    const fe h3 = ld(sh, H3), u1h2 = ld(sh, U1H2), r2 = ld(sh, R2);
    o.x = sub(sub(sub(r2, h3), u1h2), u1h2);
    const fe dd = sub(u1h2, o.x);
    const fe a = ld(sh, pick(lane, R, S1, HZ, ONE));
    const fe b = fe_select(ld(sh, pick(lane, ONE, H3, QZ, ONE)), dd, lanes_where(lane == 0));
    const fe res = mul(a, b);
    if (lane < 3) st(sh, pick(lane, T, S1H3, Z3, ONE), res);
    sync();
  }
  o.y = sub(ld(sh, T), ld(sh, S1H3));
  o.z = ld(sh, Z3);
  const fe u1 = ld(sh, U1), u2 = ld(sh, U2), s1 = ld(sh, S1), s2 = ld(sh, S2);
  const pt p = {ld(sh, PX), ld(sh, PY), ld(sh, PZ)}, q = {ld(sh, QX), ld(sh, QY), ld(sh, QZ)};
  const lmask idp = is_identity(p), idq = is_identity(q);
  const lmask ueq = fe_eq(u1, u2);
  if (__builtin_expect((idp | idq | ueq) != 0, 0)) {  // the early-outs of Add (1446-1473)
    const lmask seq = fe_eq(s1, s2);
    o = pt_select(o, identity(), uniform_mask(ueq & ~seq));
    o = pt_select(o, p, idq);
    o = pt_select(o, q, idp);
    const lmask nd = uniform_mask(ueq & seq & ~idp & ~idq);
    if (nd != 0) o = pt_select(o, pdouble(p), nd);
  }
  return o;
#endif
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

// invert (599-632): square-and-multiply over p-2, limbs visited LS->MS, bits MS->LS inside a limb;
// zero has no inverse (CtOption none; the value is zero).
FEC_DEV fe inv(const fe& a) {
  const u64 e[4] = {0xFFFFFFFEFFFFFC2DULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL};
  fe result = fe_small(1);
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
#pragma unroll 1
    for (int j = 63; j >= 0; --j) {
      result = sqr(result);
      if ((e[i] >> j) & 1) result = mul(result, a);  // exponent bits are uniform
    }
  }
  return fe_select(result, fe_zero(), fe_is_zero(a));
}

// to_affine (1342-1363): x = X * (Z^-1)^2, y = Y * (Z^-1)^2 * Z^-1; identity -> (0, 0, infinity)
FEC_DEV lmask to_affine(const pt& p, fe& x, fe& y) {
  lmask inf = is_identity(p);
  fe zi = inv(p.z);
  fe zi2 = sqr(zi);
  fe zi3 = mul(zi2, zi);
  x = fe_select(mul(p.x, zi2), fe_zero(), inf);
  y = fe_select(mul(p.y, zi3), fe_zero(), inf);
  return inf;
}

// ---- point decoding: FieldElement::from_bytes / sqrt / PointAffine::{new, from_bytes} ----
// to_montgomery (219-235): Mul by the reference's R_SQUARED constant
FEC_DEV fe to_montgomery(const fe& a) {
  fe r2 = fe_zero();
  r2.w[0] = 0x000E9F61u; r2.w[2] = 0x07A20000u; r2.w[4] = 0x00000100u;
  return mul(a, r2);
}
// FieldElement::from_bytes (182-212) on the VALUE (big-endian bytes already assembled into limbs):
// valid iff value < p; Some(to_montgomery(value)), or zero when invalid
FEC_DEV fe from_value(const fe& v, lmask& valid) {
  fe t;
  lmask ov;
  FEC_ADDK256(t, v, ov, FEC_SECP_C);  // v >= p  <=>  v + c carries out of 2^256
  valid = ~ov;
  return fe_select(fe_zero(), to_montgomery(v), uniform_mask(valid));
}
// trait pow (715-735): LSB first over 64-bit limbs; `result *= base` on set bits, `base = base.square()`.
// The exponents are compile-time constants of the reference, so every branch is wave-uniform.
FEC_DEV fe pow_lsb(const fe& a, const u64 (&e)[4]) {
  fe result = fe_small(1), base = a;
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
#pragma unroll 1
    for (int j = 0; j < 64; ++j) {
      if ((e[w] >> j) & 1) result = mul(result, base);
      base = sqr(base);
    }
  }
  return result;
}
// inherent FieldElement::sqrt (112-131): the exponent is (p+1)/4 written as FOUR 16-BIT WORDS into
// 64-bit limbs, so sqrt^2 == self fails for essentially every input; reproduced as is
FEC_DEV fe sqrt_inherent(const fe& a, lmask& is_sqrt) {
  const u64 e[4] = {0xFF0CULL, 0xFFFFULL, 0xFFFEULL, 0x3FFFULL};
  fe s = pow_lsb(a, e);
  is_sqrt = fe_eq(sqr(s), a);
  return s;
}
// PointAffine::new (856-869) / is_on_curve (978-1004): y.square() == x.square() * x + to_montgomery(7)
FEC_DEV lmask affine_on_curve(const fe& x, const fe& y) {
  fe rhs = add(mul(sqr(x), x), to_montgomery(fe_small(7)));
  return fe_eq(sqr(y), rhs);
}
// PointAffine::from_bytes (896-976) after the prefix tests: xv = the 32 x bytes as a value.
// Returns the lanes that yield Some(point).
FEC_DEV lmask decompress(const fe& xv, lmask want_odd, fe& x, fe& y) {
  lmask valid;
  x = from_value(xv, valid);
  fe y2 = add(mul(sqr(x), x), fe_small(7));   // the RAW seven (935)
  lmask is_sqrt;
  fe ye = sqrt_inherent(y2, is_sqrt);
  fe yo = neg(ye);
  fe red = mul(ye, fe_small(1));               // to_bytes: mont_reduce; byte 31 = least significant byte
  lmask parity = lanes_where((red.w[0] & 1u) != 0);
  y = fe_select(ye, yo, uniform_mask(want_odd ^ parity));
  return valid & is_sqrt & affine_on_curve(x, y);
}
// forge-ec-encoding UncompressedPoint::to_affine (point.rs:214-281) for C = Secp256k1
FEC_DEV lmask decode_uncompressed(const fe& xv, const fe& yv, fe& x, fe& y) {
  lmask vx, vy;
  x = from_value(xv, vx);
  y = from_value(yv, vy);
  fe x3 = mul(mul(x, x), x);                   // `x * x`, not square() (251-252)
  fe ax = mul(fe_zero(), x);                   // get_a() = zero (2712-2715)
  fe rhs = add(add(x3, ax), fe_small(7));      // get_b() = raw 7 (2717-2720)
  return vx & vy & fe_eq(mul(y, y), rhs) & affine_on_curve(x, y);
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

// ---- scalar field (mod n) as the reference implements it, for ECDSA verify -------------------
// The reference's N (27-28) is [0xBFD25E8CD0364141, 0xBAAEDCE6AF48A03B, 0xFFFFFFFFFFFFFFFF,
// 0xFFFFFFFFFFFFFFFE] in little-endian limbs: its two top limbs are SWAPPED relative to the true
// group order (which has 0x...FFFE in limb 2).  Every scalar-field routine of the reference
// (reduce, Mul, invert's exponent n-2, from_bytes, ct_lt against get_order()) uses that constant,
// so parity means using it too.
FEC_DEV fe N_() {
  fe n;
  n.w[0] = 0xD0364141u; n.w[1] = 0xBFD25E8Cu; n.w[2] = 0xAF48A03Bu; n.w[3] = 0xBAAEDCE6u;
  n.w[4] = 0xFFFFFFFFu; n.w[5] = 0xFFFFFFFFu; n.w[6] = 0xFFFFFFFEu; n.w[7] = 0xFFFFFFFFu;
  return n;
}
// a >= n (the comparison spelled out at 1955-1958), and Scalar::reduce (1953-1969)
FEC_DEV lmask sc_ge_n(const fe& a) {
  fe t;
  return ~sub256(t, a, N_());
}
FEC_DEV fe sc_reduce(const fe& a) {
  fe t;
  lmask borrow = sub256(t, a, N_());
  return fe_select(a, t, ~borrow);
}
// Mul for Scalar (2410-2456): the exact product of which ONLY the low 256 bits are kept, then
// reduce() and `while >= n { reduce() }` -- after one subtraction the value is < 2^256 - n < n, so
// the loop never runs a second time.
FEC_DEV fe sc_mul(const fe& a, const fe& b) {
  u32 t[8];
  mul_low256(t, a, b);
  fe r;
  FEC_UNROLL for (int i = 0; i < 8; ++i) r.w[i] = t[i];
  return sc_reduce(r);
}
// Add for Scalar (2358-2378): the 256-bit sum with its carry out DROPPED, then one reduce()
FEC_DEV fe sc_add(const fe& a, const fe& b) {
  fe t;
  (void)add256(t, a, b);
  return sc_reduce(t);
}
// invert for Scalar (2162-2195): a^(n-2), limbs LS->MS, bits MS->LS; `square()` is s * s
FEC_DEV fe sc_inv(const fe& a) {
  const u64 e[4] = {0xBFD25E8CD036413FULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFEULL};
  fe result = fe_small(1);
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
#pragma unroll 1
    for (int j = 63; j >= 0; --j) {
      result = sc_mul(result, result);
      if ((e[i] >> j) & 1) result = sc_mul(result, a);  // exponent bits are uniform
    }
  }
  return result;
}

}  // namespace secp
}  // namespace fecgpu
