// p256.hpp -- device-side NIST P-256 field / point / double-and-add, bit-exact with
// forge-ec-curves/src/p256.rs (citations are lines of that file).
//
// P-256 elements are NOT always canonical in the reference: Sub (470-496) returns
// (a-b) mod 2^256 when a < b, which is >= p with probability ~2^-32 per borrowing subtraction
// (about once per 2^20-element batch).  Every routine here is therefore defined on arbitrary
// 256-bit operands and follows the reference's loops; loops whose trip count is provably
// bounded are unrolled to that bound, with the proof next to them.
#pragma once
#include "limbs.hpp"
#include "coop.hpp"

namespace fecgpu {
namespace p256 {

// p (18-19) and 2^256 - p (437-442) as VOP2 inline constants, word 0 first
#define FEC_P256_P -1, -1, -1, 0, 0, 0, 1, -1
#define FEC_P256_RED 1, 0, 0, -1, -1, -1, -2, 0

// reduce (88-99): `while v >= p { v -= p }`.  p > 2^255, so v - p < p: at most one trip.
FEC_DEV fe csub_p(const fe& v) {
  fe w;
  lmask borrow;
  FEC_SUBK256(w, v, borrow, FEC_P256_P);
  return fe_select(v, w, ~borrow);
}

// Add (416-468), literal for arbitrary 256-bit operands, from the limb loop's sum s and carry on (the loops that follow
// depend on a and b only through them).  carry <= 1 after the limb loop.
// `while carry > 0` (436-452): the first trip adds 2^256-p; it carries again only if
// (a+b-2^256) >= p, and then the second trip cannot (s' < 2^256-p, so s' + (2^256-p) < 2^256): at
// most two trips; then reduce() (one trip).  Works IN PLACE on s: every step is a select into s.
FEC_DEV void add_tail_general(fe& s, lmask carry) {
  fe s1;
  lmask ac;
  FEC_ADDK256(s1, s, ac, FEC_P256_RED);
  s = fe_select(s, s1, carry);
  lmask again = carry & ac;  // carry - 1 + add_carry
  if (again != 0) {
    fe s2;
    lmask t;
    FEC_ADDK256(s2, s, t, FEC_P256_RED);
    (void)t;
    s = fe_select(s, s2, again);
  }
  fe w;
  lmask borrow;
  FEC_SUBK256(w, s, borrow, FEC_P256_P);
  s = fe_select(s, w, ~borrow);
}
FEC_DEV fe add_general(const fe& a, const fe& b) {
  fe s;
  lmask carry = add256(s, a, b);
  add_tail_general(s, carry);
  return s;
}

// 2^256 - p = {1, 0, 0, -1, -1, -1, -2, 0} added IN PLACE on the lanes of m
#ifdef FEC_HOST_EMUL
FEC_DEV void add_red_masked(fe& s, lmask m) {
  if (!m) return;
  fe w;
  lmask t;
  FEC_ADDK256(w, s, t, FEC_P256_RED);
  (void)t;
  s = w;
}
#else
FEC_DEV void add_red_masked(fe& s, lmask m) {
  u32 t1, tm, t2;
  asm("v_cndmask_b32_e64 %8, 0, 1, %11\n\t"
      "v_cndmask_b32_e64 %9, 0, -1, %11\n\t"
      "v_cndmask_b32_e64 %10, 0, -2, %11\n\t"
      "v_add_co_u32_e32 %0, vcc, %0, %8\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_addc_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
      "v_addc_co_u32_e32 %3, vcc, %3, %9, vcc\n\t"
      "v_addc_co_u32_e32 %4, vcc, %4, %9, vcc\n\t"
      "v_addc_co_u32_e32 %5, vcc, %5, %9, vcc\n\t"
      "v_addc_co_u32_e32 %6, vcc, %6, %10, vcc\n\t"
      "v_addc_co_u32_e32 %7, vcc, 0, %7, vcc"
      : FEC_RW8(s), "=&v"(t1), "=&v"(tm), "=&v"(t2) : "s"(m) : "vcc");
}
#endif

// Add for the common case.  For canonical operands (a, b < p) the loops of the literal form collapse to
// (a + b) mod p: with a carry, s + (2^256-p) = a+b-p < p and nothing else fires; without one the
// result is s - p if s >= p, which needs a top word of all ones in s (2^-32 per lane).  A non-canonical operand
// (possible after the reference's Sub, ~2^-32 per subtraction) needs a top word of 0xFFFFFFFF too.  Wavefronts with
// any of the three take the literal tail; everywhere else the condition is the carry alone and 2^256 - p is added
// under the carry mask.  Both legs work IN PLACE on the sum: a rare leg that produced its result elsewhere would
// cost the common leg eight v_mov at the join.
FEC_DEV fe add(const fe& a, const fe& b) {
  const lmask noncanon = lanes_where(a.w[7] == 0xFFFFFFFFu || b.w[7] == 0xFFFFFFFFu);
  fe s;
  const lmask carry = add256(s, a, b);
  if (__builtin_expect((noncanon | lanes_where(s.w[7] == 0xFFFFFFFFu)) != 0, 0)) add_tail_general(s, carry);
  else add_red_masked(s, carry);
  return s;
}
// a + a through Add (the doublings of the point formulas), with one operand list (dbl256)
FEC_DEV fe dbl(const fe& a) {
  const lmask noncanon = lanes_where(a.w[7] == 0xFFFFFFFFu);
  fe s;
  const lmask carry = dbl256(s, a);
  if (__builtin_expect((noncanon | lanes_where(s.w[7] == 0xFFFFFFFFu)) != 0, 0)) add_tail_general(s, carry);
  else add_red_masked(s, carry);
  return s;
}

// Sub (470-496): if a < b { a = a + P  (Add reduces that back to canon(a)) }; then a wrapping
// 256-bit subtraction.  For a < p the `+= P` is a no-op; for p <= a < b it subtracts p: the result is
// (a - b) - p = (a - b) + (2^256 - p) mod 2^256 on the lanes where a >= p and the subtraction borrowed.
// a >= p needs a.w[7] == 0xFFFFFFFF (2^-32 per lane): the exact test runs BEFORE the subtraction and leaves only a
// lane mask, and the correction is applied IN PLACE, so that the common path neither keeps a copy of a nor pays
// moves at a join.
FEC_DEV fe sub(const fe& a, const fe& b) {
  lmask ge = 0;
  if (__builtin_expect(lanes_where(a.w[7] == 0xFFFFFFFFu) != 0, 0)) {
    fe t;
    lmask bo;
    FEC_SUBK256(t, a, bo, FEC_P256_P);
    ge = ~bo;  // a >= p, exactly
  }
  fe d;
  const lmask borrow = sub256(d, a, b);
  if (__builtin_expect((ge & borrow) != 0, 0)) add_red_masked(d, uniform_mask(ge & borrow));
  return d;
}

// Neg (707-729): 0 -> 0, else p - a (wrapping).
FEC_DEV fe neg(const fe& a) {
  fe r;
  lmask t;
  FEC_KSUB256(r, a, t, FEC_P256_P);
  (void)t;
  return fe_select(r, a, fe_is_zero(a));
}

// reduce_wide_p256 (544-704) on the exact 512-bit product, words c0..c15.
FEC_DEV fe reduce_wide(const u32 c[16]) {
  typedef long long i64;
  i64 acc[8];
  // s1 + 2*s2 + 2*s3 + s4 + s5 - s6 - s7 - s8 - s9, word by word (582-654)
  acc[0] = (i64)c[0] + c[8] + c[9] - c[11] - c[12] - c[13] - c[14];
  acc[1] = (i64)c[1] + c[9] + c[10] - c[12] - c[13] - c[14] - c[15];
  acc[2] = (i64)c[2] + c[10] + c[11] - c[13] - c[14] - c[15];
  acc[3] = (i64)c[3] + 2 * (i64)c[11] + 2 * (i64)c[12] + c[13] - c[15] - c[8] - c[9];
  acc[4] = (i64)c[4] + 2 * (i64)c[12] + 2 * (i64)c[13] + c[14] - c[9] - c[10];
  acc[5] = (i64)c[5] + 2 * (i64)c[13] + 2 * (i64)c[14] + c[15] - c[10] - c[11];
  acc[6] = (i64)c[6] + 2 * (i64)c[14] + 2 * (i64)c[15] + c[14] + c[13] - c[8] - c[9];
  acc[7] = (i64)c[7] + 2 * (i64)c[15] + c[15] + c[8] - c[10] - c[11] - c[12] - c[13];
  // carry propagation with arithmetic shift (657-665)
  FEC_UNROLL for (int i = 0; i < 7; ++i) {
    i64 carry = acc[i] >> 32;
    acc[i] &= 0xFFFFFFFFLL;
    acc[i + 1] += carry;
  }
  i64 carry = acc[7] >> 32;
  acc[7] &= 0xFFFFFFFFLL;
  // `while carry > 0 { r -= p }` / `while carry < 0 { r += p }` (677-698), all wrapping:
  // r = L - carry*p mod 2^256 = L + carry*(2^224 - 2^192 - 2^96 + 1) mod 2^256.
  acc[0] += carry;
  acc[3] -= carry;
  acc[6] -= carry;
  acc[7] += carry;
  fe r;
  FEC_UNROLL for (int i = 0; i < 7; ++i) {
    i64 cy = acc[i] >> 32;
    r.w[i] = (u32)acc[i];
    acc[i + 1] += cy;
  }
  r.w[7] = (u32)acc[7];
  return csub_p(r);  // reduce() (701)
}

FEC_DEV fe mul_small_cxx(const fe& a, u32 k) {  // FieldElement::from(k) * a  (1893-1904)
  u32 t[16];
  mul_wide_small(t, a, k);
  return reduce_wide(t);
}

// Mul (498-534): exact schoolbook product, then reduce_wide_p256.  Compiler-scheduled form: the host
// emulation's Mul and the cross-check of the hand-allocated one below.
FEC_DEV fe mul_cxx(const fe& a, const fe& b) {
  u32 t[16];
  mul_wide(t, a, b);
  return reduce_wide(t);
}
// square() (772-776) is self * self: the same exact 512-bit product, formed with the 97-instruction
// squaring (28 doubled cross products + 8 squares) instead of the 128-instruction general product
FEC_DEV fe sqr_cxx(const fe& a) {
  u32 t[16];
  sqr_wide(t, a);
  return reduce_wide(t);
}
#ifdef FEC_HOST_EMUL
FEC_DEV fe mul(const fe& a, const fe& b) { return mul_cxx(a, b); }
FEC_DEV fe sqr(const fe& a) { return sqr_cxx(a); }
#else
// the closing reduce() (701) of a product: r >= p needs a top word of all ones (2^-32 per lane)
FEC_DEV fe csub_p_top(const fe& v) {
  if (__builtin_expect(lanes_where(v.w[7] == 0xFFFFFFFFu) != 0, 0)) return csub_p(v);
  return v;
}
// Mul / square() as ONE hand-allocated asm statement each (tools/gen_field_asm.py): product scanning
// in a fixed register block, then reduce_wide_p256 as 256-bit carry chains with a signed ninth word
// and r = L - carry*p (677-698) built from the carry word.
FEC_DEV fe mul(const fe& a, const fe& b) {
  fe r;
  asm(FEC_P256_MUL_ASM
      : "=v"(r.w[0]), "=v"(r.w[1]), "=v"(r.w[2]), "=v"(r.w[3]), "=v"(r.w[4]), "=v"(r.w[5]), "=v"(r.w[6]),
        "=v"(r.w[7])
      : FEC_V8(a), FEC_V8(b)
      : FEC_P256_MUL_CLOBBERS);
  return csub_p_top(r);
}
// FieldElement::from(3) * a and from(8) * a (1893, 1904): nine-word product, S = T_lo + c8*(2^256 - p)
template <u32 K>
FEC_DEV fe mul_small_k(const fe& a) {
  static_assert(K == 3 || K == 8, "the doubling multiplies by 3 and 8 only");
  fe r;
  lmask sink, exc;
  if (K == 3) {
    asm(FEC_P256_MUL3_ASM
        : "=v"(r.w[0]), "=v"(r.w[1]), "=v"(r.w[2]), "=v"(r.w[3]), "=v"(r.w[4]), "=v"(r.w[5]), "=v"(r.w[6]),
          "=v"(r.w[7]), "=&s"(sink), "=&s"(exc)
        : FEC_V8(a)
        : FEC_P256_MUL3_CLOBBERS);
  } else {
    asm(FEC_P256_MUL8_ASM
        : "=v"(r.w[0]), "=v"(r.w[1]), "=v"(r.w[2]), "=v"(r.w[3]), "=v"(r.w[4]), "=v"(r.w[5]), "=v"(r.w[6]),
          "=v"(r.w[7]), "=&s"(sink), "=&s"(exc)
        : FEC_V8(a)
        : FEC_P256_MUL8_CLOBBERS);
  }
  if (__builtin_expect(exc != 0, 0)) r = fe_select(r, mul_small_cxx(a, K), ~(lmask)0);  // the sum carried out of 2^256
  return csub_p_top(r);
}
FEC_DEV fe sqr(const fe& a) {
  fe r;
  lmask sink;
  asm(FEC_P256_SQR_ASM
      : "=v"(r.w[0]), "=v"(r.w[1]), "=v"(r.w[2]), "=v"(r.w[3]), "=v"(r.w[4]), "=v"(r.w[5]), "=v"(r.w[6]),
        "=v"(r.w[7]), "=&s"(sink)
      : FEC_V8(a)
      : FEC_P256_SQR_CLOBBERS);
  return csub_p_top(r);
}
#endif
#ifdef FEC_HOST_EMUL
FEC_DEV fe mul_small(const fe& a, u32 k) { return mul_small_cxx(a, k); }
#else
FEC_DEV fe mul_small(const fe& a, u32 k) { return k == 3 ? mul_small_k<3>(a) : (k == 8 ? mul_small_k<8>(a) : mul_small_cxx(a, k)); }
#endif


struct pt {
  fe x, y, z;
};

FEC_DEV pt identity() {  // 1827-1829
  pt p;
  p.x = fe_zero();
  p.y = fe_small(1);
  p.z = fe_zero();
  return p;
}
FEC_DEV lmask is_identity(const pt& p) { return fe_is_zero(p.z); }  // 1831-1833
FEC_DEV pt pt_select(const pt& a, const pt& b, lmask choice) {
  pt r;
  r.x = fe_select(a.x, b.x, choice);
  r.y = fe_select(a.y, b.y, choice);
  r.z = fe_select(a.z, b.z, choice);
  return r;
}

// double (1869-1912): dbl-2009-l for a = 0 (no a*Z^4 term although a = -3); Z == 1 shortcut.
FEC_DEV pt pdouble(const pt& p) {
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

// Add (1938-2007) with ConstantTimeEq (2034-2068) folded in: the ct_eq products are the same
// z1z1, z2z2, u1, u2, s1, s2 the addition needs, so they are computed once.  need_double is set
// where the reference returns self.double().
FEC_DEV pt padd_nodouble(const pt& p, const pt& q, lmask& need_double) {
  fe z1z1 = sqr(p.z);
  fe z2z2 = sqr(q.z);
  fe u1 = mul(p.x, z2z2);
  fe u2 = mul(q.x, z1z1);
  fe s1 = mul(mul(p.y, q.z), z2z2);
  fe s2 = mul(mul(q.y, p.z), z1z1);
  fe h = sub(u2, u1);
  fe i = sqr(add(h, h));
  fe j = mul(h, i);
  fe s21 = sub(s2, s1);
  fe r = add(s21, s21);
  fe v = mul(u1, i);
  pt o;
  o.x = sub(sub(sub(sqr(r), j), v), v);
  o.y = sub(mul(r, sub(v, o.x)), mul(add(s1, s1), j));
  o.z = mul(sub(sub(sqr(add(p.z, q.z)), z1z1), z2z2), h);
  lmask idp = is_identity(p), idq = is_identity(q);
  lmask ueq = fe_eq(u1, u2);
  lmask same = ueq & fe_eq(s1, s2);  // ct_eq (1951)
  lmask opposite = 0;
  if (ueq != 0) opposite = ueq & fe_eq(s1, neg(s2));  // 1977; u1 == u2 never holds on random inputs
  o = pt_select(o, identity(), opposite);
  o = pt_select(o, p, idq);
  o = pt_select(o, q, idp);
  need_double = same & ~idp & ~idq;
  return o;
}

FEC_DEV pt padd(const pt& p, const pt& q) {
  lmask nd;
  pt o = padd_nodouble(p, q, nd);
  if (nd != 0) {
    pt d = pdouble(p);
    o = pt_select(o, d, nd);
  }
  return o;
}

// ---- Add (1938-2007) spread over up to FIVE lanes of one wavefront (see coop.hpp) ----------------------------
// The same sixteen products on the same operands as padd_nodouble() (square() is self * self in the reference,
// so every one is a Mul), level by level:
//   level 1   z1z1 = z1 z1     z2z2 = z2 z2     t1 = y1 z2      t2 = y2 z1      zz = (z1+z2)(z1+z2)
//   level 2   u1 = x1 z2z2     u2 = x2 z1z1     s1 = t1 z2z2    s2 = t2 z1z1
//             h = u2 - u1, r = 2 (s2 - s1), hh = 2 h, zs = zz - z1z1 - z2z2
//   level 3   i = hh hh        rr = r r         z3 = zs h
//   level 4   j = h i          v = u1 i                         x3 = rr - j - 2 v
//   level 5   t = r (v - x3)   w = (2 s1) j                     y3 = t - w
// 5 field-operation latencies per addition instead of 16.
namespace coop {
enum { PX = 0, PY, PZ, QX, QY, QZ, ZSUM, Z1Z1, Z2Z2, T1, T2, ZZ, U1, U2, S1, S2, H, R, HH, ZS, I, RR, Z3, J, V, DD, S1D, T, W, ONE, SLOTS };
constexpr int WORDS = SLOTS * 8;
// lane l (< n) multiplies slot a_l by slot b_l into slot o_l; the other lanes work on the constant 1
FEC_DEV void level(u32* sh, int n, int a0, int a1, int a2, int a3, int a4, int b0, int b1, int b2, int b3, int b4, int o0,
                   int o1, int o2, int o3, int o4) {
  using namespace coopx;
  const int lane = lane_id();
  const fe a = ld(sh, pick(lane, a0, a1, a2, a3, a4, ONE)), b = ld(sh, pick(lane, b0, b1, b2, b3, b4, ONE));
  const fe res = mul(a, b);
  if (lane < n) st(sh, pick(lane, o0, o1, o2, o3, o4, ONE), res);
  sync();
}
}  // namespace coop

// sh: coop::WORDS words of LDS owned by this wavefront (16-byte aligned); slots PX..QZ hold p and q on entry
// (written by the caller, followed by coopx::sync()); slot ONE holds the constant 1.  Every lane returns the sum.
FEC_DEV pt padd_coop(u32* sh) {
  using namespace coop;
  using coopx::ld;
  using coopx::st;
  const pt p = {ld(sh, PX), ld(sh, PY), ld(sh, PZ)}, q = {ld(sh, QX), ld(sh, QY), ld(sh, QZ)};
#ifdef FEC_HOST_EMUL
  return padd(p, q);
#else
  const int lane = coopx::lane_id();
  if (lane == 0) st(sh, ZSUM, add(p.z, q.z));
  coopx::sync();
  level(sh, 5, PZ, QZ, PY, QY, ZSUM, PZ, QZ, QZ, PZ, ZSUM, Z1Z1, Z2Z2, T1, T2, ZZ);
  level(sh, 4, PX, QX, T1, T2, ONE, Z2Z2, Z1Z1, Z2Z2, Z1Z1, ONE, U1, U2, S1, S2, ONE);
  const fe u1 = ld(sh, U1), u2 = ld(sh, U2), s1 = ld(sh, S1), s2 = ld(sh, S2);
  const fe h = sub(u2, u1);
  const fe s21 = sub(s2, s1);
  const fe r = add(s21, s21);
  if (lane == 0) {
    st(sh, H, h);
    st(sh, R, r);
    st(sh, HH, add(h, h));
    st(sh, ZS, sub(sub(ld(sh, ZZ), ld(sh, Z1Z1)), ld(sh, Z2Z2)));
    st(sh, S1D, add(s1, s1));
  }
  coopx::sync();
  level(sh, 3, HH, R, ZS, ONE, ONE, HH, R, H, ONE, ONE, I, RR, Z3, ONE, ONE);
  level(sh, 2, H, U1, ONE, ONE, ONE, I, I, ONE, ONE, ONE, J, V, ONE, ONE, ONE);
  const fe j = ld(sh, J), v = ld(sh, V);
  pt o;
  o.x = sub(sub(sub(ld(sh, RR), j), v), v);
  if (lane == 0) st(sh, DD, sub(v, o.x));
  coopx::sync();
  level(sh, 2, R, S1D, ONE, ONE, ONE, DD, J, ONE, ONE, ONE, T, W, ONE, ONE, ONE);
  o.y = sub(ld(sh, T), ld(sh, W));
  o.z = ld(sh, Z3);
  const lmask idp = is_identity(p), idq = is_identity(q);
  const lmask ueq = fe_eq(u1, u2);
  if (__builtin_expect((idp | idq | ueq) != 0, 0)) {  // the early-outs of Add, as in padd_nodouble / padd
    const lmask same = uniform_mask(ueq & fe_eq(s1, s2));
    const lmask opposite = uniform_mask(ueq & fe_eq(s1, neg(s2)));
    o = pt_select(o, identity(), opposite);
    o = pt_select(o, p, idq);
    o = pt_select(o, q, idp);
    const lmask nd = uniform_mask(same & ~idp & ~idq);
    if (nd != 0) o = pt_select(o, pdouble(p), nd);
  }
  return o;
#endif
}

// pow (376-393), LSB first: `if e & 1 { result *= base }; base = base.square()`
// invert (343-370): zero -> none (value zero); exponent p - 2 from the limb-wise borrow loop.
FEC_DEV fe inv(const fe& a) {
  const u64 e[4] = {0xFFFFFFFFFFFFFFFDULL, 0x00000000FFFFFFFFULL, 0x0000000000000000ULL, 0xFFFFFFFF00000001ULL};
  fe result = fe_small(1);
  fe base = a;
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
      if ((e[w] >> i) & 1) result = mul(result, base);  // exponent bits are uniform
      base = sqr(base);
    }
  }
  return fe_select(result, fe_zero(), fe_is_zero(a));
}

// ---- point decoding ----
FEC_DEV fe B_() {  // 28-33
  fe b;
  b.w[0] = 0x27D2604Bu; b.w[1] = 0x3BCE3C3Eu; b.w[2] = 0xCC53B0F6u; b.w[3] = 0x651D06B0u;
  b.w[4] = 0x769886BCu; b.w[5] = 0xB3EBBD55u; b.w[6] = 0xAA3A93E7u; b.w[7] = 0x5AC635D8u;
  return b;
}
// pow (376-393), LSB first, with a constant (wave-uniform) exponent
FEC_DEV fe pow_lsb(const fe& a, const u64 (&e)[4]) {
  fe result = fe_small(1), base = a;
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
      if ((e[w] >> i) & 1) result = mul(result, base);
      base = sqr(base);
    }
  }
  return result;
}
// FieldElement::from_bytes (303-317): the value itself, valid iff < p
FEC_DEV lmask value_lt_p(const fe& v) {
  fe t;
  lmask borrow;
  FEC_SUBK256(t, v, borrow, FEC_P256_P);
  return borrow;
}
// x^3 - 3x + b exactly as PointAffine::new (1536-1543) and from_bytes (1612-1618) spell it
FEC_DEV fe curve_rhs(const fe& x) {
  fe x3 = mul(sqr(x), x);
  fe three_x = mul(fe_small(3), x);
  return add(sub(x3, three_x), B_());
}
// inherent sqrt (320-339) with the reference's own exponent
FEC_DEV fe sqrt_inherent(const fe& a, lmask& is_sqrt) {
  const u64 e[4] = {0xC0000000ULL, 0x40000000ULL, 0x4000000000000000ULL, 0x40000000C0000000ULL};
  fe s = pow_lsb(a, e);
  is_sqrt = fe_eq(sqr(s), a);
  return s;
}
// PointAffine::from_bytes (1580-1639) after the prefix tests; no curve check at the end (1638)
FEC_DEV lmask decompress(const fe& xv, lmask want_odd, fe& x, fe& y) {
  x = xv;
  lmask valid = value_lt_p(xv);
  lmask is_sqrt;
  fe r = sqrt_inherent(curve_rhs(x), is_sqrt);
  lmask parity = lanes_where((r.w[0] & 1u) != 0);   // to_bytes (288-300) is the raw limbs
  y = fe_select(r, neg(r), uniform_mask(parity ^ want_odd));
  return valid & is_sqrt;
}
// UncompressedPoint::to_affine (point.rs:214-281) for C = P256, then PointAffine::new (1535-1552)
FEC_DEV lmask decode_uncompressed(const fe& xv, const fe& yv, fe& x, fe& y) {
  x = xv;
  y = yv;
  lmask v = value_lt_p(xv) & value_lt_p(yv);
  fe a;  // get_a() exactly as written (2177-2180): 64-bit limbs holding 32-bit patterns
  a.w[0] = 0xFFFFFFFCu; a.w[1] = 0; a.w[2] = 0xFFFFFFFFu; a.w[3] = 0; a.w[4] = 0xFFFFFFFEu; a.w[5] = 0;
  a.w[6] = 0xFFFFFFFFu; a.w[7] = 0;
  fe x3 = mul(mul(x, x), x);
  fe rhs = add(add(x3, mul(a, x)), B_());
  return v & fe_eq(mul(y, y), rhs) & fe_eq(sqr(y), curve_rhs(x));
}

// to_affine (1835-1857)
FEC_DEV lmask to_affine(const pt& p, fe& x, fe& y) {
  lmask inf = is_identity(p);
  fe zi = inv(p.z);
  fe zi2 = sqr(zi);
  fe zi3 = mul(zi2, zi);
  x = fe_select(mul(p.x, zi2), fe_zero(), inf);
  y = fe_select(mul(p.y, zi3), fe_zero(), inf);
  return inf;
}

// Curve::multiply (2120-2156): MSB-first over the big-endian inherent Scalar::to_bytes
// (1026-1038), i.e. bit 255-i of the 256-bit scalar at step i; `if bit == 1 { result + point }`
// is per-lane data dependent, so the addition is computed for the wavefront and selected.
FEC_DEV pt multiply(const pt& point, const u32* kw) {
  u32 any = 0;
  FEC_UNROLL for (int i = 0; i < 8; ++i) any |= kw[i * KSTRIDE];
  lmask early = is_identity(point) | lanes_where(any == 0);
  pt result = identity();
#pragma unroll 1
  for (int i = 0; i < 256; ++i) {
    int b = 255 - i;
    lmask bit = lanes_where(((kw[(b >> 5) * KSTRIDE] >> (b & 31)) & 1u) != 0);
    // one pdouble instance in the code object; the second pass runs only when Add (1951) returns
    // self.double() for some lane (result == point projectively) -- never on random inputs
    pt din = result;
    pt d, s;
    lmask nd = 0;
#pragma unroll 1
    for (int pass = 0;; ++pass) {
      pt o = pdouble(din);
      if (pass == 0) {
        d = o;
        s = padd_nodouble(d, point, nd);
        nd = nd & bit;
        if (nd == 0) break;
        din = d;
      } else {
        s = pt_select(s, o, nd);
        break;
      }
    }
    result = pt_select(d, s, bit);
  }
  return pt_select(result, identity(), early);
}

// ---- scalar field as the reference implements it (p256.rs:875-1038, 1409-1432), for ECDSA verify ----
// 64-bit limbs like the reference, one element per lane, ordinary (divergent) control flow: this is a
// few hundred multiplications per signature beside the two 256-step point multiplications.
// Mul = the exact 512-bit product, then reduce_wide (924-1020), which is NOT a reduction mod n: its
// second folding round adds only the low four limbs of high2 * (2^256 - n) (993-998).
struct sc { u64 l[4]; };
FEC_DEV sc sc_of(const fe& a) {
  sc r;
  FEC_UNROLL for (int i = 0; i < 4; ++i) r.l[i] = (u64)a.w[2 * i] | ((u64)a.w[2 * i + 1] << 32);
  return r;
}
FEC_DEV fe sc_fe(const sc& a) {
  fe r;
  FEC_UNROLL for (int i = 0; i < 4; ++i) { r.w[2 * i] = (u32)a.l[i]; r.w[2 * i + 1] = (u32)(a.l[i] >> 32); }
  return r;
}
FEC_DEV bool sc_is_zero(const sc& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
FEC_DEV bool sc_ge_n(const sc& a) {  // compare_with_n(a) >= 0 (889-905); N at 23-24
  const u64 N[4] = {0xF3B9CAC2FC632551ULL, 0xBCE6FAADA7179E84ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFF00000000ULL};
  u64 borrow = 0;
  FEC_UNROLL for (int i = 0; i < 4; ++i) {
    const u64 d = a.l[i] - N[i];
    const u64 b1 = a.l[i] < N[i];
    const u64 b2 = d < borrow;
    borrow = b1 | b2;
  }
  return borrow == 0;
}
// Mul (1409-1432) = the exact 512-bit product, then reduce_wide (924-1020), on 32-bit words and branch-free:
//   first  = low + high * C, exact (943-969; the reference's u128 columns cannot overflow: C's 64-bit limbs are
//            60, 63, 0 and 32 bits wide);
//   second : low2 += the LOW FOUR LIMBS of high2 * C only (993-998) -- computed unconditionally: high2 == 0
//            contributes zero, exactly like the skipped branch at 976;
//   carry  : if that addition carries out, limb i receives c * C[i] with c the RUNNING carry (1000-1007), i.e.
//            C is added limb by limb only for as long as each limb addition itself carries;
//   last   : while >= n subtract n (runs at most once: low2 < 2^256 < 2n).
FEC_DEV fe SC_C_() {  // TWO_256_MINUS_N (932-937)
  fe c;
  c.w[0] = 0x039CDAAFu; c.w[1] = 0x0C46353Du; c.w[2] = 0x58E8617Bu; c.w[3] = 0x43190552u;
  c.w[4] = 0; c.w[5] = 0; c.w[6] = 0xFFFFFFFFu; c.w[7] = 0;
  return c;
}
FEC_DEV fe SC_N_() {  // 23-24
  fe n;
  n.w[0] = 0xFC632551u; n.w[1] = 0xF3B9CAC2u; n.w[2] = 0xA7179E84u; n.w[3] = 0xBCE6FAADu;
  n.w[4] = 0xFFFFFFFFu; n.w[5] = 0xFFFFFFFFu; n.w[6] = 0x00000000u; n.w[7] = 0xFFFFFFFFu;
  return n;
}
FEC_DEV fe sc_reduce_wide(const u32 (&w)[16]) {
  fe low, high;
  FEC_UNROLL for (int i = 0; i < 8; ++i) { low.w[i] = w[i]; high.w[i] = w[8 + i]; }
  u32 p[16];
  mul_wide(p, high, SC_C_());
  fe plo, phi;
  FEC_UNROLL for (int i = 0; i < 8; ++i) { plo.w[i] = p[i]; phi.w[i] = p[8 + i]; }
  fe low2, high2;
  const lmask c0 = add256(low2, plo, low);
  (void)add256_cin(high2, phi, fe_zero(), c0);            // first < 2^481: no carry out of the top
  u32 p2[8];
  mul_low256(p2, high2, SC_C_());
  fe p2f;
  FEC_UNROLL for (int i = 0; i < 8; ++i) p2f.w[i] = p2[i];
  fe sum;
  const lmask cy = add256(sum, low2, p2f);
  // the carry round in the reference's 64-bit limbs, selected where the addition carried
  const u64 C[4] = {0x0C46353D039CDAAFULL, 0x4319055258E8617BULL, 0ULL, 0x00000000FFFFFFFFULL};
  fe fixed;
  u64 c = 1;
  FEC_UNROLL for (int i = 0; i < 4; ++i) {
    const u64 l = (u64)sum.w[2 * i] | ((u64)sum.w[2 * i + 1] << 32);
    const u64 lo = c * C[i], hi = mulhi64(c, C[i]);
    const u64 t = l + lo;
    c = hi + (t < lo);
    fixed.w[2 * i] = (u32)t;
    fixed.w[2 * i + 1] = (u32)(t >> 32);
  }
  const fe v = fe_select(sum, fixed, cy);
  fe d;
  const lmask borrow = sub256(d, v, SC_N_());
  return fe_select(d, v, borrow);
}
FEC_DEV fe sc_mul32(const fe& a, const fe& b) {
  u32 t[16];
  mul_wide(t, a, b);
  return sc_reduce_wide(t);
}
FEC_DEV sc sc_mul(const sc& a, const sc& b) { return sc_of(sc_mul32(sc_fe(a), sc_fe(b))); }  // 1409-1432
// invert (1057-1080) = pow(n - 2) (1083-1100): limbs and bits LS -> MS, `result *= base` on a set bit,
// base = base.square() = base * base every step
FEC_DEV fe sc_inv32(const fe& a) {
  const u64 e[4] = {0xF3B9CAC2FC63254FULL, 0xBCE6FAADA7179E84ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFF00000000ULL};
  fe result = fe_small(1), base = a;
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
#pragma unroll 1
    for (int k = 0; k < 64; ++k) {
      if ((e[i] >> k) & 1) result = sc_mul32(result, base);  // exponent bits are uniform
      base = sc_mul32(base, base);
    }
  }
  return result;
}
FEC_DEV sc sc_inv(const sc& a) { return sc_of(sc_inv32(sc_fe(a))); }
// Add (1352-1375): on a carry out of the top limb reduce() still sees only the low 256 bits
FEC_DEV sc sc_add(const sc& a, const sc& b) {
  const u64 N[4] = {0xF3B9CAC2FC632551ULL, 0xBCE6FAADA7179E84ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFF00000000ULL};
  sc r;
  u64 carry = 0;
  FEC_UNROLL for (int i = 0; i < 4; ++i) {
    const u64 s1 = a.l[i] + b.l[i];
    const u64 o1 = s1 < a.l[i];
    const u64 s2 = s1 + carry;
    const u64 o2 = s2 < s1;
    r.l[i] = s2;
    carry = o1 + o2;
  }
  if (carry > 0 || sc_ge_n(r)) {
    while (sc_ge_n(r)) {  // reduce (911-920)
      u64 borrow = 0;
      FEC_UNROLL for (int i = 0; i < 4; ++i) {
        const u64 d1 = r.l[i] - N[i];
        const u64 b1 = r.l[i] < N[i];
        const u64 d2 = d1 - borrow;
        const u64 b2 = d1 < borrow;
        r.l[i] = d2;
        borrow = b1 + b2;
      }
    }
  }
  return r;
}
// Scalar::ct_lt(self, get_order()) -- P-256 keeps the trait DEFAULT (forge-ec-core/src/lib.rs:497-531):
// over big-endian bytes, result |= eq_so_far & !borrow(other_byte - self_byte), i.e. "self_byte <=
// other_byte" while all earlier bytes were equal.  At byte 0 the chain is trivially equal, and a byte-0
// pair with self > other both leaves result clear and breaks the chain for good: the verdict is
// top_byte(self) <= top_byte(other).  Against n (top byte 0xFF) that is true for every value.
FEC_DEV bool sc_ct_lt_default(const sc& a, const sc& other) { return (a.l[3] >> 56) <= (other.l[3] >> 56); }

}  // namespace p256
}  // namespace fecgpu
