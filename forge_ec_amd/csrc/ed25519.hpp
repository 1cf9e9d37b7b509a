// ed25519.hpp -- device-side Ed25519 field / extended-point / binary method, bit-exact with
// forge-ec-curves/src/ed25519.rs (citations are lines of that file).
//
// Quirks reproduced: reduce_wide folds the top carry with x19 instead of x38 (278); the
// addition uses d (not 2d) and Z1*Z2 (not 2*Z1*Z2) (1897-1900); the negation early-out compares
// raw projective coordinates (1878), so doubling a point with X = 0 returns the identity.
#pragma once
#include "limbs.hpp"
#include "coop.hpp"

namespace fecgpu {
namespace ed {

FEC_DEV fe D_() {  // 86-91
  fe d;
  d.w[0] = 0x135EDEFFu; d.w[1] = 0x75EB4DCAu; d.w[2] = 0x8283B156u; d.w[3] = 0x00E0149Au;
  d.w[4] = 0xEEF3D130u; d.w[5] = 0x198E80F2u; d.w[6] = 0xC61A8E3Cu; d.w[7] = 0x2406875Cu;
  return d;
}

// p = 2^255 - 19 (64-69).  x - p = (x + 19) with bit 255 toggled, x + p = (x - 19) with bit 255
// toggled (mod 2^256), so every p-operand chain is a chain on the small constant 19.

// reduce (214-247): clear bit 255 and add 19 if it was set (no carry out is possible); then
// subtract p if the value is >= p, i.e. if value + 19 reaches 2^255.
FEC_DEV fe reduce(const fe& a) {
  fe v = a;
  const u32 top = a.w[7] >> 31;
  v.w[7] &= 0x7FFFFFFFu;
  const lmask cy = add_short1(v, top * 19u);          // leaves word 0 with probability 19 / 2^32
  if (__builtin_expect(cy != 0, 0)) carry_from<1>(v, cy);
  // v < 2^255 + 19.  v >= p = 2^255 - 19 needs either bit 255 set again or words 1..6 all ones with
  // w7 == 0x7FFFFFFF, either way a top word >= 0x7FFFFFFF: ~2^-31 per lane, so the exact comparison sits behind a
  // wave-uniform branch.
  if (__builtin_expect(lanes_where(v.w[7] >= 0x7FFFFFFFu) != 0, 0)) {
    fe u;
    lmask t;
    FEC_ADDK256(u, v, t, 19, 0, 0, 0, 0, 0, 0, 0);  // u = v + 19 < 2^256
    (void)t;
    lmask ge = lanes_where((u.w[7] >> 31) != 0);  // v >= p
    u.w[7] &= 0x7FFFFFFFu;                       // u - 2^255 = v - p
    v = fe_select(v, u, ge);
  }
  return v;
}

// Add (458-488): a + b; if it carried out of 2^256, add 19 (wrapping); then reduce().  Canonical operands
// (< 2^255) never carry: the + 19 sits behind a wave-uniform branch.
FEC_DEV fe add(const fe& a, const fe& b) {
  fe s;
  const lmask carry = add256(s, a, b);
  if (__builtin_expect(carry != 0, 0)) {
    fe s2;
    add_word256(s2, s, word_select(0u, 19u, carry));
    s = s2;
  }
  return reduce(s);
}

// Sub (490-520): a - b; add p (wrapping) if it borrowed.  No reduce.  d + p = d - 19 + 2^255 (mod 2^256).
FEC_DEV fe sub(const fe& a, const fe& b) {
  fe d;
  const lmask borrow = sub256(d, a, b);
  lmask bw;  // the borrow of the - 19 leaves word 0 with probability 19 / 2^32
#ifdef FEC_HOST_EMUL
  bw = sub_short1(d, word_select(0u, 19u, borrow));
  d.w[7] ^= word_select(0u, 0x80000000u, borrow);
#else
  u32 t;
  asm("v_cndmask_b32_e64 %3, 0, 19, %4\n\t"
      "v_sub_co_u32_e32 %0, vcc, %0, %3\n\t"
      "v_cndmask_b32_e64 %3, 0, 1, %4\n\t"
      "v_lshl_add_u32 %1, %3, 31, %1\n\t"     // toggling bit 255 = adding 2^31 to the top word
      "s_mov_b64 %2, vcc"                      // last: %2 may share its registers with %4
      : "+v"(d.w[0]), "+v"(d.w[7]), "=s"(bw), "=&v"(t) : "s"(borrow) : "vcc");
#endif
  if (__builtin_expect(bw != 0, 0)) borrow_from<1>(d, bw);  // commutes with the toggle (both are additions mod 2^32 on word 7)
  return d;
}

// Neg (547-570): p - a (wrapping), 0 -> 0.
FEC_DEV fe neg(const fe& a) {
  fe r;
  lmask t;
  FEC_KSUB256(r, a, t, 0xffffffed, -1, -1, -1, -1, -1, -1, -1);  // 2^256 - 19 - a
  (void)t;
  r.w[7] ^= 0x80000000u;                                         // - 2^255
  return fe_select(r, fe_zero(), fe_is_zero(a));
}

// reduce_wide (260-289): low + 38*high with the carry chain of the reference; the carry out of
// limb 3 is multiplied by 19 (not 38) and added back, a second carry out is dropped; reduce().
FEC_DEV fe reduce_wide(const u32 t[16]) {
  fe low;
  u64 carry = 0;
  FEC_UNROLL for (int i = 0; i < 8; ++i) {
    u64 p = (u64)t[8 + i] * 38u + t[i] + carry;
    low.w[i] = (u32)p;
    carry = p >> 32;
  }
  // carry == floor((low + 38*high) / 2^256) <= 38
  fe low2;
  add_word256(low2, low, (u32)carry * 19u);
  return reduce(low2);
}

// Mul (522-545), compiler-scheduled form: the host emulation's Mul, the cross-check of the
// hand-allocated one below and its fallback for the wavefronts where a small addition carries
FEC_DEV fe mul_cxx(const fe& a, const fe& b) {
  u32 t[16];
  mul_wide(t, a, b);
  return reduce_wide(t);
}
// self * self (623-625) with the same exact 512-bit product formed by the cheaper squaring
FEC_DEV fe sqr_cxx(const fe& a) {
  u32 t[16];
  sqr_wide(t, a);
  return reduce_wide(t);
}
#ifdef FEC_HOST_EMUL
FEC_DEV fe mul(const fe& a, const fe& b) { return mul_cxx(a, b); }
FEC_DEV fe sqr_exact(const fe& a) { return sqr_cxx(a); }
#else
// Mul / self*self as ONE hand-allocated asm statement each (tools/gen_field_asm.py), for the path on
// which neither of reduce_wide's small additions (carry*19, then 19 for bit 255) carries out of
// word 0 and the value stays below p; the other lanes (~2^-21 per product) are returned as a mask
// and such a wavefront recomputes with the compiler-scheduled routine.
FEC_DEV fe mul(const fe& a, const fe& b) {
  fe r;
  lmask sink, exc;
  asm(FEC_ED_MUL_ASM
      : "=v"(r.w[0]), "=v"(r.w[1]), "=v"(r.w[2]), "=v"(r.w[3]), "=v"(r.w[4]), "=v"(r.w[5]), "=v"(r.w[6]),
        "=v"(r.w[7]), "=&s"(sink), "=&s"(exc)
      : FEC_V8(a), FEC_V8(b)
      : FEC_ED_MUL_CLOBBERS);
  if (__builtin_expect(exc != 0, 0)) r = fe_select(r, mul_cxx(a, b), ~(lmask)0);
  // (the rare leg is selected INTO r: a leg that returned its own registers would cost the common leg eight v_mov at the join)
  return r;
}
FEC_DEV fe sqr_exact(const fe& a) {
  fe r;
  lmask sink, exc;
  asm(FEC_ED_SQR_ASM
      : "=v"(r.w[0]), "=v"(r.w[1]), "=v"(r.w[2]), "=v"(r.w[3]), "=v"(r.w[4]), "=v"(r.w[5]), "=v"(r.w[6]),
        "=v"(r.w[7]), "=&s"(sink), "=&s"(exc)
      : FEC_V8(a)
      : FEC_ED_SQR_CLOBBERS);
  if (__builtin_expect(exc != 0, 0)) r = fe_select(r, sqr_cxx(a), ~(lmask)0);
  return r;
}
#endif

struct pt {
  fe x, y, z, t;
};

FEC_DEV pt identity() {  // 1776-1783
  pt p;
  p.x = fe_zero();
  p.y = fe_small(1);
  p.z = fe_small(1);
  p.t = fe_zero();
  return p;
}
FEC_DEV lmask is_identity(const pt& p) {  // 1785-1791
  return fe_is_zero(p.x) & fe_eq(p.y, p.z) & fe_is_zero(p.t);
}
FEC_DEV pt pt_select(const pt& a, const pt& b, lmask choice) {
  pt r;
  r.x = fe_select(a.x, b.x, choice);
  r.y = fe_select(a.y, b.y, choice);
  r.z = fe_select(a.z, b.z, choice);
  r.t = fe_select(a.t, b.t, choice);
  return r;
}

// Add for ExtendedPoint (1864-1928); double() is add(self, self) (1828-1832).
FEC_DEV pt padd(const pt& p, const pt& q) {
  fe a = mul(sub(p.y, p.x), sub(q.y, q.x));
  fe b = mul(add(p.y, p.x), add(q.y, q.x));
  fe c = mul(mul(p.t, q.t), D_());
  fe d = mul(p.z, q.z);
  fe e = sub(b, a);
  fe f = sub(d, c);
  fe g = add(d, c);
  fe h = add(b, a);
  pt o;
  o.x = mul(e, f);
  o.y = mul(g, h);
  o.t = mul(e, h);
  o.z = mul(f, g);
  // the three early-outs (1869-1880) are improbable after the first addition of a lane (whose
  // accumulator is still the identity): selects only when some lane of the wavefront needs them
  lmask opposite = fe_eq(p.x, neg(q.x)) & fe_eq(p.y, q.y);  // 1878, raw coordinates
  lmask idp = is_identity(p), idq = is_identity(q);
  if (__builtin_expect((opposite | idp | idq) != 0, 0)) {
    o = pt_select(o, identity(), opposite);
    o = pt_select(o, p, idq);
    o = pt_select(o, q, idp);
  }
  return o;
}

// ---- Add for ExtendedPoint (1864-1928) spread over FOUR lanes of one wavefront (see coop.hpp) ----------------
//   level 1   a = (y1-x1)(y2-x2)   b = (y1+x1)(y2+x2)   tt = t1 t2   d = z1 z2
//   level 2   c = tt D                                                      e = b-a, f = d-c, g = d+c, h = b+a
//   level 3   x3 = e f             y3 = g h             t3 = e h     z3 = f g
// 3 field-operation latencies per addition instead of 9; the same products on the same operands as padd().
namespace coop {
enum { PX = 0, PY, PZ, PT, QX, QY, QZ, QT, YMX1, YMX2, YPX1, YPX2, A, B, TT, D, DCONST, C, E, F, G, H, X3, Y3, T3, Z3, ONE, SLOTS };
constexpr int WORDS = SLOTS * 8;
FEC_DEV void level(u32* sh, int n, int a0, int a1, int a2, int a3, int b0, int b1, int b2, int b3, int o0, int o1, int o2, int o3) {
  using namespace coopx;
  const int lane = lane_id();
  const fe a = ld(sh, pick(lane, a0, a1, a2, a3, ONE, ONE)), b = ld(sh, pick(lane, b0, b1, b2, b3, ONE, ONE));
  const fe res = mul(a, b);
  if (lane < n) st(sh, pick(lane, o0, o1, o2, o3, ONE, ONE), res);
  sync();
}
}  // namespace coop

// sh: coop::WORDS words of LDS owned by this wavefront (16-byte aligned); slots PX..QT hold p and q on entry
// (written by the caller, followed by coopx::sync()); slots ONE and DCONST hold 1 and the curve constant d.
FEC_DEV pt padd_coop(u32* sh) {
  using namespace coop;
  using coopx::ld;
  using coopx::st;
  const pt p = {ld(sh, PX), ld(sh, PY), ld(sh, PZ), ld(sh, PT)}, q = {ld(sh, QX), ld(sh, QY), ld(sh, QZ), ld(sh, QT)};
#ifdef FEC_HOST_EMUL
  return padd(p, q);
#else
  const int lane = coopx::lane_id();
  if (lane == 0) {
    st(sh, YMX1, sub(p.y, p.x));
    st(sh, YMX2, sub(q.y, q.x));
    st(sh, YPX1, add(p.y, p.x));
    st(sh, YPX2, add(q.y, q.x));
  }
  coopx::sync();
  level(sh, 4, YMX1, YPX1, PT, PZ, YMX2, YPX2, QT, QZ, A, B, TT, D);
  level(sh, 1, TT, ONE, ONE, ONE, DCONST, ONE, ONE, ONE, C, ONE, ONE, ONE);
  const fe a = ld(sh, A), b = ld(sh, B), c = ld(sh, C), d = ld(sh, D);
  if (lane == 0) {
    st(sh, E, sub(b, a));
    st(sh, F, sub(d, c));
    st(sh, G, add(d, c));
    st(sh, H, add(b, a));
  }
  coopx::sync();
  level(sh, 4, E, G, E, F, F, H, H, G, X3, Y3, T3, Z3);
  pt o = {ld(sh, X3), ld(sh, Y3), ld(sh, Z3), ld(sh, T3)};
  const lmask opposite = fe_eq(p.x, neg(q.x)) & fe_eq(p.y, q.y);  // 1878, raw coordinates
  const lmask idp = is_identity(p), idq = is_identity(q);
  if (__builtin_expect((opposite | idp | idq) != 0, 0)) {
    o = pt_select(o, identity(), uniform_mask(opposite));
    o = pt_select(o, p, idq);
    o = pt_select(o, q, idp);
  }
  return o;
#endif
}

// double() (1828-1832) = self + self: Add with both operands equal.  Its four self-products
// (y-x)^2, (y+x)^2, t*t, z*z are the same exact 512-bit products as Mul's, formed with the squaring.
FEC_DEV pt pdbl(const pt& p) {
  fe a = sqr_exact(sub(p.y, p.x));
  fe b = sqr_exact(add(p.y, p.x));
  fe c = mul(sqr_exact(p.t), D_());
  fe d = sqr_exact(p.z);
  fe e = sub(b, a);
  fe f = sub(d, c);
  fe g = add(d, c);
  fe h = add(b, a);
  pt o;
  o.x = mul(e, f);
  o.y = mul(g, h);
  o.t = mul(e, h);
  o.z = mul(f, g);
  // Add's early-outs with q = p (1869-1880): "opposite" needs x == -x, i.e. x == 0
  lmask opposite = fe_eq(p.x, neg(p.x));
  lmask idp = is_identity(p);
  if (__builtin_expect((opposite | idp) != 0, 0)) {
    o = pt_select(o, identity(), opposite);
    o = pt_select(o, p, idp);
  }
  return o;
}

// pow (410-431): every bit computes result * base and selects it on the bit; base = base.square().
// invert (603-621): a^(p-2) (for zero the CtOption is none; the value pow returns is 0).
FEC_DEV fe inv(const fe& a) {
  const u64 e[4] = {0xFFFFFFFFFFFFFFEBULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x7FFFFFFFFFFFFFFFULL};
  fe result = fe_small(1);
  fe base = a;
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
      // the discarded product (bit clear) is dead work in the reference; the bit is uniform
      if ((e[w] >> i) & 1) result = mul(result, base);
      base = sqr_exact(base);
    }
  }
  return result;
}

// ---- point decoding ----
// inherent pow (410-431) with a constant (wave-uniform) exponent: the products of clear bits are
// computed and discarded by the reference; only the kept ones are formed here
FEC_DEV fe pow_lsb(const fe& a, const u64 (&e)[4]) {
  fe result = fe_small(1), base = a;
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
      if ((e[w] >> i) & 1) result = mul(result, base);
      base = sqr_exact(base);
    }
  }
  return result;
}
// FieldElement::from_bytes (315-357) on the little-endian value: the comparison loop returns None as
// soon as ANY 64-bit limb exceeds the same limb of p (346-348), whatever the higher limbs decided;
// otherwise valid iff value < p.  p's limbs 1..2 are all ones, so only limb 0 (> 2^64 - 19) and limb 3
// (> 2^63 - 1) can exceed.
FEC_DEV lmask value_valid(const fe& v) {
  lmask gt0 = lanes_where(v.w[1] == 0xFFFFFFFFu && v.w[0] > 0xFFFFFFEDu);
  lmask gt3 = lanes_where((v.w[7] >> 31) != 0);
  // value < p  <=>  value + 19 < 2^255 (for values below 2^255)
  fe t;
  lmask c;
  FEC_ADDK256(t, v, c, 19, 0, 0, 0, 0, 0, 0, 0);
  lmask lt = lanes_where((t.w[7] >> 31) == 0) & ~c;
  return lt & ~gt0 & ~gt3;
}
FEC_DEV fe SQRT_M1_() {  // 132-137
  fe s;
  s.w[0] = 0x4A0EA0B0u; s.w[1] = 0xC4EE1B27u; s.w[2] = 0xAD2FE478u; s.w[3] = 0x2F431806u;
  s.w[4] = 0x3DFBD7A7u; s.w[5] = 0x2B4D0099u; s.w[6] = 0x4FC1DF0Bu; s.w[7] = 0x2B832480u;
  return s;
}
// inherent sqrt (359-402)
FEC_DEV fe sqrt_inherent(const fe& a, lmask& valid) {
  const u64 e1[4] = {0x7FFFFFFFFFFFFFF6ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x3FFFFFFFFFFFFFFFULL};
  const u64 e2[4] = {0x1FFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x0FFFFFFFFFFFFFFFULL};
  fe leg = pow_lsb(a, e1);
  lmask is_qr = fe_eq(leg, fe_small(1)) | fe_is_zero(leg);
  fe cand = pow_lsb(a, e2);
  lmask ok1 = fe_eq(sqr_exact(cand), a);
  fe alt = mul(cand, SQRT_M1_());
  lmask ok2 = fe_eq(sqr_exact(alt), a);
  valid = is_qr & (ok1 | ok2);
  return fe_select(cand, alt, uniform_mask(ok2));
}
// PointAffine::from_bytes (1526-1582) after the prefix tests: the MONTGOMERY-curve equation
// x^3 + a x^2 + x with a = 0x7FFFFFDA is what the reference evaluates
FEC_DEV lmask decompress(const fe& xv, lmask want_odd, fe& x, fe& y) {
  x = xv;
  lmask valid = value_valid(xv);
  fe x2 = sqr_exact(x);
  fe x3 = mul(x2, x);
  fe y2 = add(add(x3, mul(fe_small(0x7FFFFFDAu), x2)), x);
  lmask is_sqrt;
  fe r = sqrt_inherent(y2, is_sqrt);
  fe red = reduce(r);                                // to_bytes: reduce(), little-endian -> byte 31 is the top byte
  lmask parity = lanes_where(((red.w[7] >> 24) & 1u) != 0);
  y = fe_select(r, neg(r), uniform_mask(parity ^ want_odd));
  return valid & is_sqrt;
}
// UncompressedPoint::to_affine (point.rs:214-281) for C = Ed25519, then PointAffine::new (1476-1498)
FEC_DEV lmask decode_uncompressed(const fe& xv, const fe& yv, fe& x, fe& y) {
  x = xv;
  y = yv;
  lmask v = value_valid(xv) & value_valid(yv);
  fe a = fe_zero();  // get_a() exactly as written (2107-2110)
  a.w[0] = 0xFFFFFFEDu; a.w[1] = 0x7FFFFFFFu; a.w[2] = 0xFFFFFFFFu; a.w[3] = 0x0007FFFFu;
  fe x3 = mul(mul(x, x), x);
  fe rhs = add(add(x3, mul(a, x)), fe_zero());       // get_b() = zero
  lmask eq1 = fe_eq(mul(y, y), rhs);
  fe x2 = sqr_exact(x), y2 = sqr_exact(y);
  fe lhs = add(neg(x2), y2);
  fe rh2 = add(fe_small(1), mul(D_(), mul(x2, y2)));
  return v & eq1 & fe_eq(lhs, rh2);
}

// to_affine (1793-1811): x = X * Z^-1, y = Y * Z^-1; identity -> (0, 0, infinity).
FEC_DEV lmask to_affine(const pt& p, fe& x, fe& y) {
  lmask inf = is_identity(p);
  fe zi = inv(p.z);
  x = fe_select(mul(p.x, zi), fe_zero(), inf);
  y = fe_select(mul(p.y, zi), fe_zero(), inf);
  return inf;
}

// Curve::multiply (2062-2097): LSB-first over scalar.to_raw(); every step computes
// result + addend, selects it on the bit, and doubles the addend.
FEC_DEV pt multiply(const pt& point, const u32* kw) {
  u32 any = 0;
  FEC_UNROLL for (int i = 0; i < 8; ++i) any |= kw[i * KSTRIDE];
  lmask early = is_identity(point) | lanes_where(any == 0);
  pt result = identity();
  pt addend = point;
#pragma unroll 1
  for (int i = 0; i < 256; ++i) {
    lmask bit = lanes_where(((kw[(i >> 5) * KSTRIDE] >> (i & 31)) & 1u) != 0);
    // one padd instance: pass 0 is result + addend (kept where the bit is set), pass 1 doubles
    // the addend.  The reference discards pass 0 when the bit is clear (2085-2086).
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      pt lhs = pass == 0 ? result : addend;
      pt o = padd(lhs, addend);
      if (pass == 0) result = pt_select(result, o, bit);
      else addend = o;
    }
  }
  return pt_select(result, identity(), early);
}

// Fixed-base Curve::multiply (BASELINE config 3).  The reference's addend sequence
// B, B+B, (B+B)+(B+B), ... (2089: `addend = addend.double()`) does not depend on the scalar, so
// the 256 addends are computed once (k_ed_build_table, the same padd chain) and staged in LDS;
// each lane then performs only the additions its set bits select, in the reference's order
// (ascending bit index), from its own position in the table: `result + addend` is computed for
// every bit by the reference but kept only where the bit is set (2085-2086).  Entry j, word w of
// the table sits at tab[j * ED_TSTRIDE + w]; the odd stride spreads lanes reading different
// entries over the LDS banks.
constexpr int ED_TSTRIDE = 33;

FEC_DEV pt table_entry(const u32* tab, u32 j) {
  const u32* e = tab + j * ED_TSTRIDE;
  pt p;
  FEC_UNROLL for (int i = 0; i < 8; ++i) {
    p.x.w[i] = e[i];
    p.y.w[i] = e[8 + i];
    p.z.w[i] = e[16 + i];
    p.t.w[i] = e[24 + i];
  }
  return p;
}

FEC_DEV pt multiply_fixed(const pt& base, const u32* tab, const u32* kw) {
  u32 any = 0;
  FEC_UNROLL for (int i = 0; i < 8; ++i) any |= kw[i * KSTRIDE];
  lmask early = is_identity(base) | lanes_where(any == 0);
  pt result = identity();
  int wi = 0;
  u32 cur = kw[0];
#pragma unroll 1
  for (;;) {
    while (cur == 0 && wi < 7) {  // advance to this lane's next non-zero scalar word
      ++wi;
      cur = kw[wi * KSTRIDE];
    }
    bool have = cur != 0;
    lmask active = lanes_where(have);
    if (active == 0) break;  // every lane of the wavefront has consumed its set bits
    u32 b = have ? (u32)__builtin_ctz(cur) : 0u;
    cur &= cur - 1;
    pt addend = table_entry(tab, have ? (u32)wi * 32u + b : 0u);
    pt sum = padd(result, addend);
    result = pt_select(result, sum, active);
  }
  return pt_select(result, identity(), early);
}


// ---- Scalar arithmetic the generic schnorr::batch_verify::<Ed25519, D> needs (schnorr.rs:264: s_i * a_i) ----
struct sc4 {
  u64 l[4];
};
FEC_DEV bool sc_ge_order(const sc4& r) {   // the comparison loop of 1215-1226 / 1303-1312 / 1354-1363: r >= ORDER
  const u64 ORDER[4] = {0x5812631A5CF5D3EDULL, 0x14DEF9DEA2F79CD6ULL, 0ULL, 0x1000000000000000ULL};
  bool ge = true, decided = false;
  FEC_UNROLL for (int i = 3; i >= 0; --i) {
    if (!decided && r.l[i] < ORDER[i]) {
      ge = false;
      decided = true;
    } else if (!decided && r.l[i] > ORDER[i]) {
      decided = true;
    }
  }
  return ge;
}
FEC_DEV sc4 sc_sub_order(const sc4& r) {   // 1229-1236 / 1365-1372: the limbs minus ORDER with a borrow chain
  const u64 ORDER[4] = {0x5812631A5CF5D3EDULL, 0x14DEF9DEA2F79CD6ULL, 0ULL, 0x1000000000000000ULL};
  sc4 o;
  u64 borrow = 0;
  FEC_UNROLL for (int i = 0; i < 4; ++i) {
    const u64 d1 = r.l[i] - ORDER[i];
    const u64 b1 = r.l[i] < ORDER[i];
    const u64 d2 = d1 - borrow;
    const u64 b2 = d1 < borrow;
    o.l[i] = d2;
    borrow = b1 | b2;
  }
  return o;
}
// impl Add for Scalar (1193-1239): the 256-bit sum (the carry out of the top limb is dropped), then ONE conditional
// subtraction of the order
FEC_DEV sc4 sc_add(const sc4& a, const sc4& b) {
  sc4 r;
  u64 carry = 0;
  FEC_UNROLL for (int i = 0; i < 4; ++i) {
    const u64 s1 = a.l[i] + b.l[i];
    const u64 o1 = s1 < a.l[i];
    const u64 s2 = s1 + carry;
    const u64 o2 = s2 < s1;
    r.l[i] = s2;
    carry = o1 | o2;
  }
  return sc_ge_order(r) ? sc_sub_order(r) : r;
}
// impl Mul for Scalar (1256-1376) AS THE RELEASE PROFILE RUNS IT (the reference's Cargo.toml:53-58 has no
// overflow-checks: integer overflow wraps).  The eight column sums `product[i + j] += a_i * b_j` (1268-1272) add up to
// four 128-bit products in a u128, and `product[i] += carry` (1278) adds once more: both can pass 2^128.  A debug build
// panics there; the release build -- the one whose throughput BASELINE times -- keeps the sum modulo 2^128 and goes on,
// which is what this computes.  `overflowed` is set when any of those additions wrapped, i.e. when a debug build would
// have panicked on these operands.  Then, literally: the low four limbs; if any high limb is non-zero, 256 times
// `result += high` with the Add above (1343-1349); one conditional subtraction of the order (1352-1373).
FEC_DEV sc4 sc_mul_release(const sc4& a, const sc4& b, bool& overflowed) {
  u64 lo[8], hi[8];
  FEC_UNROLL for (int k = 0; k < 8; ++k) lo[k] = hi[k] = 0;
  bool ovf = false;
  FEC_UNROLL for (int i = 0; i < 4; ++i) {
    FEC_UNROLL for (int j = 0; j < 4; ++j) {
      const u64 pl = a.l[i] * b.l[j], ph = mulhi64(a.l[i], b.l[j]);
      const u64 nl = lo[i + j] + pl;
      const u64 c = nl < pl;
      const u64 nh = hi[i + j] + ph;
      const bool o1 = nh < ph;
      const u64 nh2 = nh + c;
      const bool o2 = nh2 < nh;
      ovf = ovf || o1 || o2;
      lo[i + j] = nl;
      hi[i + j] = nh2;
    }
  }
  u64 limb[8];
  u64 carry = 0;   // (`carry = product[i] >> 64` is below 2^64)
  FEC_UNROLL for (int k = 0; k < 8; ++k) {
    const u64 nl = lo[k] + carry;
    const u64 c = nl < carry;
    const u64 nh = hi[k] + c;
    ovf = ovf || (nh < c);
    limb[k] = nl;
    carry = nh;
  }
  overflowed = ovf;
  sc4 result, high;
  FEC_UNROLL for (int i = 0; i < 4; ++i) {
    result.l[i] = limb[i];
    high.l[i] = limb[4 + i];
  }
  const bool high_nonzero = (high.l[0] | high.l[1] | high.l[2] | high.l[3]) != 0;
  if (!(high_nonzero || sc_ge_order(result))) return result;           // 1299-1313: already below the order
  if (high_nonzero) {
#pragma unroll 1
    for (int k = 0; k < 256; ++k) result = sc_add(result, high);       // 1347-1349
  }
  return sc_ge_order(result) ? sc_sub_order(result) : result;          // 1352-1373
}

}  // namespace ed
}  // namespace fecgpu
