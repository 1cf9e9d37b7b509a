// canon_curves.hpp -- CANONICAL-MATH MODE: the REAL curves.
//
// NOT reference parity.  The reference's curve arithmetic is not a field / not a group (DESIGN.md
// section 2), so its outputs are not keys any other library accepts.  This header implements
//   secp256k1   y^2 = x^3 + 7        over p = 2^256 - 2^32 - 977
//   P-256       y^2 = x^3 - 3x + b   over p = 2^256 - 2^224 + 2^192 + 2^96 - 1
// for callers who want standard results (key generation, ECDH) at GPU speed: plain canonical
// field elements (8 x u32, always in [0, p), no Montgomery form), Jacobian coordinates, a 4-bit
// fixed-base comb for k*G, 4-bit fixed windows for k*P, a batched Montgomery-trick normalisation.
// It is validated against an independent big-integer model kept with the tests and public
// standard vectors (SEC2 / BIP-340 multiples of G, the RFC 6979 A.2.5 key pair), never against
// the reference.
#pragma once
#include "ed25519.hpp"
#include "p256.hpp"
#include "secp256k1.hpp"

namespace fecgpu {
namespace canon {

// a^e for a public constant exponent (8 words, little-endian): square-and-multiply, MSB first
template <class F>
FEC_DEV fe pow_const(const fe& a, const u32* e) {
  fe r = a;
  int i = 255;
  while (i > 0 && !((e[i >> 5] >> (i & 31)) & 1u)) --i;  // leading one: r = a
#pragma unroll 1
  for (--i; i >= 0; --i) {
    r = F::sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1u) r = F::mul(r, a);
  }
  return r;
}

// ---- F_p, p = 2^256 - 2^32 - 977 (secp256k1) ----------------------------------------------------
struct FpSecp {
// add / sub / neg of the reference ARE correct modulo p for canonical operands (secp256k1.rs
// 353-440, 509-539 restate the textbook conditional-subtract forms); they are reused.
FEC_SDEV fe add(const fe& a, const fe& b) { return secp::add(a, b); }
FEC_SDEV fe sub(const fe& a, const fe& b) { return secp::sub(a, b); }
FEC_SDEV fe neg(const fe& a) { return secp::neg(a); }
FEC_SDEV fe dbl(const fe& a) { return secp::add(a, a); }

// t (512 bits) mod p for any t.  2^256 = c (mod p), c = 2^32 + 977:  t = lo + hi * c.
FEC_SDEV fe reduce512_general(const u32 t[16]) {
  // u = hi * 977  (9 words)
  u32 u[9];
  {
    u32 carry = 0;
    FEC_UNROLL for (int k = 0; k < 8; ++k) {
      u64 p = (u64)t[8 + k] * 977u + carry;
      u[k] = (u32)p;
      carry = (u32)(p >> 32);
    }
    u[8] = carry;
  }
  fe lo, ul, hs, v, w;
  FEC_UNROLL for (int i = 0; i < 8; ++i) {
    lo.w[i] = t[i];
    ul.w[i] = u[i];
  }
  hs.w[0] = 0;  // (hi << 32), low 8 words
  FEC_UNROLL for (int i = 1; i < 8; ++i) hs.w[i] = t[7 + i];
  lmask c1 = add256(v, lo, ul);
  lmask c2 = add256(w, v, hs);
  // what spilled past 2^256: u[8] + hi[7] + carries  (< 2^34)
  u64 top = (u64)u[8] + t[15] + word_select(0u, 1u, c1) + word_select(0u, 1u, c2);
  // fold it once more: w += top * c = top * 977 + (top << 32)
  u64 f0 = (u64)(u32)top * 977u;                 // top_lo * 977
  u64 f1 = (u64)(u32)(top >> 32) * 977u;         // top_hi * 977 (top_hi <= 3)
  // addend words: a0 = lo(f0); a1 = hi(f0) + lo(f1) + top_lo; a2 = carries + top_hi  (all small)
  u64 a1 = (f0 >> 32) + (u32)f1 + (u32)top;
  u64 a2 = (a1 >> 32) + (f1 >> 32) + (top >> 32);
  fe addend = fe_zero();
  addend.w[0] = (u32)f0;
  addend.w[1] = (u32)a1;
  addend.w[2] = (u32)a2;
  fe r;
  lmask c3 = add256(r, w, addend);
  if (__builtin_expect(c3 != 0, 0)) {  // wrapped past 2^256 once more (probability ~2^-190): add c
    fe r2;
    lmask t2;
    FEC_ADDK256(r2, r, t2, FEC_SECP_C);
    (void)t2;
    r = fe_select(r, r2, c3);
  }
  return secp::csub_p_unlikely(r);
}

// The same reduction for the common case, 27 VALU instructions.  Column k of  lo + hi * c  is
//   p_k = l_k + h_k * 977 + h_k * 2^32  =  mad(h_k, 977, {l_k, h_k})          (one v_mad_u64_u32)
// which fits 64 bits whenever h_k <= 2^32 - 978; the 9-word sum of the columns is one carry
// chain, and its top word r8 <= h_7 + 978 folds the same way.  Lanes with any h_k >= 2^32 - 4096
// (about one multiplication in 2^17) take reduce512_general instead.
FEC_SDEV fe reduce512(const u32 t[16]) {
  u32 hmax = t[8];
  FEC_UNROLL for (int k = 9; k < 16; ++k) hmax = hmax > t[k] ? hmax : t[k];
  if (__builtin_expect(lanes_where(hmax >= 0xFFFFF000u) != 0, 0)) return reduce512_general(t);
  fe a, b;
  u32 top;
  {
    u64 p[8];
    FEC_UNROLL for (int k = 0; k < 8; ++k) p[k] = (u64)t[8 + k] * 977u + (((u64)t[8 + k] << 32) | t[k]);
    FEC_UNROLL for (int k = 0; k < 8; ++k) a.w[k] = (u32)p[k];
    b.w[0] = 0;
    FEC_UNROLL for (int k = 1; k < 8; ++k) b.w[k] = (u32)(p[k - 1] >> 32);
    top = (u32)(p[7] >> 32);
  }
  fe r;
  lmask c = add256(r, a, b);
  top += word_select(0u, 1u, c);  // <= h_7 + 978: no wrap
  u64 q = (u64)top * 977u + (((u64)top << 32) | r.w[0]);
  r.w[0] = (u32)q;
  fe r2;
  lmask c2 = add_lohi256(r2, r, 0u, (u32)(q >> 32));
  if (__builtin_expect(c2 != 0, 0)) {  // wrapped past 2^256 (probability ~2^-190): add c once more
    fe r3;
    lmask t2;
    FEC_ADDK256(r3, r2, t2, FEC_SECP_C);
    (void)t2;
    r2 = fe_select(r2, r3, c2);
  }
  return secp::csub_p_unlikely(r2);
}

FEC_SDEV fe mul(const fe& a, const fe& b) {
  u32 t[16];
  mul_wide(t, a, b);
  return reduce512(t);
}
FEC_SDEV fe sqr(const fe& a) {
  u32 t[16];
  sqr_wide(t, a);
  return reduce512(t);
}
FEC_SDEV fe mul3(const fe& a) { return add(add(a, a), a); }
FEC_SDEV fe mul8(const fe& a) { return dbl(dbl(dbl(a))); }

// a^(p-2): the standard addition chain for p = 2^256 - 2^32 - 977 (255 squarings, 15 multiplications)
FEC_SDEV fe sqr_n(fe a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) a = sqr(a);
  return a;
}
FEC_SDEV fe inv(const fe& a) {
  fe x2 = mul(sqr(a), a);
  fe x3 = mul(sqr(x2), a);
  fe x6 = mul(sqr_n(x3, 3), x3);
  fe x9 = mul(sqr_n(x6, 3), x3);
  fe x11 = mul(sqr_n(x9, 2), x2);
  fe x22 = mul(sqr_n(x11, 11), x11);
  fe x44 = mul(sqr_n(x22, 22), x22);
  fe x88 = mul(sqr_n(x44, 44), x44);
  fe x176 = mul(sqr_n(x88, 88), x88);
  fe x220 = mul(sqr_n(x176, 44), x44);
  fe x223 = mul(sqr_n(x220, 3), x3);
  // p - 2 = 2^256 - 2^32 - 979: binary = 223 ones, 0, 22 ones, 0000, 1, 0, 11, 0, 1  (1..1 0 1..1 0000101101)
  fe t = sqr_n(x223, 23);
  t = mul(t, x22);
  t = sqr_n(t, 5);
  t = mul(t, a);
  t = sqr_n(t, 3);
  t = mul(t, x2);
  t = sqr_n(t, 2);
  t = mul(t, a);
  return t;
}
// x, y < p and y^2 == x^3 + 7
FEC_SDEV lmask ge_p(const fe& v) {  // v >= p  <=>  v + c carries out of 2^256
  fe w;
  lmask ov;
  FEC_ADDK256(w, v, ov, FEC_SECP_C);
  return ov;
}
};

// ---- F_p, p = 2^256 - 2^224 + 2^192 + 2^96 - 1 (P-256) -----------------------------------------
// p = {-1, -1, -1, 0, 0, 0, 1, -1} and 2^256 - p = {1, 0, 0, -1, -1, -1, -2, 0} as 32-bit words: every
// word is a VOP2 inline constant, so adding / subtracting p costs no registers.
#define FEC_P256_P -1, -1, -1, 0, 0, 0, 1, -1
#define FEC_P256_NEGP 1, 0, 0, -1, -1, -1, -2, 0
#define FEC_P256_5P_LOW -5, -1, -1, 4, 0, 0, 5, -5 /* 5p = 4 * 2^256 + this */
struct FpP256 {
  // the reference's Add and Neg are the textbook forms and correct modulo p (p256.rs:416-468, 707-729)
  FEC_SDEV fe add(const fe& a, const fe& b) { return p256::add(a, b); }
  FEC_SDEV fe neg(const fe& a) { return p256::neg(a); }
  FEC_SDEV lmask ge_p(const fe& v) {  // v >= p  <=>  v + (2^256 - p) carries
    fe w;
    lmask ov;
    FEC_ADDK256(w, v, ov, FEC_P256_NEGP);
    return ov;
  }
  FEC_SDEV fe csub_p_unlikely(const fe& v) {  // v >= p needs w7 = 0xFFFFFFFF and w6 >= 1: ~2^-32
    if (__builtin_expect(lanes_where(v.w[7] == 0xFFFFFFFFu && v.w[6] != 0) != 0, 0)) {
      fe w;
      lmask ov;
      FEC_ADDK256(w, v, ov, FEC_P256_NEGP);
      return fe_select(v, w, ov);
    }
    return v;
  }
  FEC_SDEV fe sub(const fe& a, const fe& b) {  // a - b, plus p when it borrowed
    fe d, d2;
    lmask borrow = sub256(d, a, b);
    lmask t;
    FEC_ADDK256(d2, d, t, FEC_P256_P);
    (void)t;
    return fe_select(d, d2, borrow);
  }
  // t (512 bits) mod p: the NIST fast reduction (FIPS 186-4 D.2.3) on 32-bit words c0..c15,
  //   r = s1 + 2 s2 + 2 s3 + s4 + s5 - s6 - s7 - s8 - s9,
  // as 256-bit carry chains: the positive terms plus 5p (so the total cannot go negative), minus
  // the negative terms; what spilled past 2^256 (t in 0..12, counted by a ninth instruction on
  // each chain) is folded back with 2^256 = 2^224 - 2^192 - 2^96 + 1 (mod p).
  FEC_SDEV fe reduce512(const u32 c[16]) {
    fe s1, s2, s3, s4, s5, s6, s7, s8, s9;
    FEC_UNROLL for (int i = 0; i < 8; ++i) s1.w[i] = c[i];
    s2.w[0] = 0; s2.w[1] = 0; s2.w[2] = 0; s2.w[3] = c[11]; s2.w[4] = c[12]; s2.w[5] = c[13]; s2.w[6] = c[14]; s2.w[7] = c[15];
    s3.w[0] = 0; s3.w[1] = 0; s3.w[2] = 0; s3.w[3] = c[12]; s3.w[4] = c[13]; s3.w[5] = c[14]; s3.w[6] = c[15]; s3.w[7] = 0;
    s4.w[0] = c[8]; s4.w[1] = c[9]; s4.w[2] = c[10]; s4.w[3] = 0; s4.w[4] = 0; s4.w[5] = 0; s4.w[6] = c[14]; s4.w[7] = c[15];
    s5.w[0] = c[9]; s5.w[1] = c[10]; s5.w[2] = c[11]; s5.w[3] = c[13]; s5.w[4] = c[14]; s5.w[5] = c[15]; s5.w[6] = c[13]; s5.w[7] = c[8];
    s6.w[0] = c[11]; s6.w[1] = c[12]; s6.w[2] = c[13]; s6.w[3] = 0; s6.w[4] = 0; s6.w[5] = 0; s6.w[6] = c[8]; s6.w[7] = c[10];
    s7.w[0] = c[12]; s7.w[1] = c[13]; s7.w[2] = c[14]; s7.w[3] = c[15]; s7.w[4] = 0; s7.w[5] = 0; s7.w[6] = c[9]; s7.w[7] = c[11];
    s8.w[0] = c[13]; s8.w[1] = c[14]; s8.w[2] = c[15]; s8.w[3] = c[8]; s8.w[4] = c[9]; s8.w[5] = c[10]; s8.w[6] = 0; s8.w[7] = c[12];
    s9.w[0] = c[14]; s9.w[1] = c[15]; s9.w[2] = 0; s9.w[3] = c[9]; s9.w[4] = c[10]; s9.w[5] = c[11]; s9.w[6] = 0; s9.w[7] = c[13];
    u32 tp = 0;
    fe a, b;
    add256c(a, s2, s3, tp);
    tp += tp;              // 2 (s2 + s3) = 2 a + 2 carry * 2^256
    add256c(b, a, a, tp);
    add256c(b, b, s1, tp);
    add256c(b, b, s4, tp);
    add256c(b, b, s5, tp);
    {
      lmask ov;
      FEC_ADDK256(b, b, ov, FEC_P256_5P_LOW);
      tp += 4u + word_select(0u, 1u, ov);
    }
    u32 tn = 0;
    fe n;
    add256c(n, s6, s7, tn);
    add256c(n, n, s8, tn);
    add256c(n, n, s9, tn);
    u32 t = tp - tn;
    fe v;
    sub256c(v, b, n, t);   // t in 0..12
    // K(t) = t * (2^256 - p) = {t, 0, 0, -t, -nz, -nz, -t - nz, t - nz},  nz = (t != 0)
    const u32 nz = t != 0 ? 1u : 0u;
    fe k;
    k.w[0] = t; k.w[1] = 0; k.w[2] = 0; k.w[3] = 0u - t; k.w[4] = 0u - nz; k.w[5] = 0u - nz;
    k.w[6] = 0u - t - nz; k.w[7] = t - nz;
    fe r;
    lmask c2 = add256(r, v, k);
    if (__builtin_expect(c2 != 0, 0)) {  // wrapped once more (probability ~ t / 2^32): add 2^256 - p again
      fe r2;
      lmask t2;
      FEC_ADDK256(r2, r, t2, FEC_P256_NEGP);
      (void)t2;
      r = fe_select(r, r2, c2);
    }
    return csub_p_unlikely(r);
  }
  FEC_SDEV fe mul(const fe& a, const fe& b) {
    u32 t[16];
    mul_wide(t, a, b);
    return reduce512(t);
  }
  FEC_SDEV fe sqr(const fe& a) {
    u32 t[16];
    sqr_wide(t, a);
    return reduce512(t);
  }
  FEC_SDEV fe inv(const fe& a) {  // a^(p-2)
    const u32 e[8] = {0xFFFFFFFDu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 1u, 0xFFFFFFFFu};
    return pow_const<FpP256>(a, e);
  }
};

struct jac {
  fe x, y, z;
};  // Z == 0 <=> point at infinity
struct aff {
  fe x, y;
};

FEC_DEV jac jac_infinity() {
  jac p;
  p.x = fe_small(1);
  p.y = fe_small(1);
  p.z = fe_zero();
  return p;
}
FEC_DEV jac jac_select(const jac& a, const jac& b, lmask m) {
  jac r;
  r.x = fe_select(a.x, b.x, m);
  r.y = fe_select(a.y, b.y, m);
  r.z = fe_select(a.z, b.z, m);
  return r;
}

// ---- per-lane global-memory helpers -------------------------------------------------------------
#ifdef FEC_HOST_EMUL
FEC_DEV fe ld8(const u32* p) {
  fe a;
  for (int i = 0; i < 8; ++i) a.w[i] = p[i];
  return a;
}
FEC_DEV void st8(u32* p, const fe& a) {
  for (int i = 0; i < 8; ++i) p[i] = a.w[i];
}
#else
FEC_DEV fe ld8(const u32* p) {
  const uint4* s = reinterpret_cast<const uint4*>(p);
  uint4 a = s[0], b = s[1];
  fe r;
  r.w[0] = a.x; r.w[1] = a.y; r.w[2] = a.z; r.w[3] = a.w;
  r.w[4] = b.x; r.w[5] = b.y; r.w[6] = b.z; r.w[7] = b.w;
  return r;
}
FEC_DEV void st8(u32* p, const fe& a) {
  uint4* d = reinterpret_cast<uint4*>(p);
  d[0] = make_uint4(a.w[0], a.w[1], a.w[2], a.w[3]);
  d[1] = make_uint4(a.w[4], a.w[5], a.w[6], a.w[7]);
}
#endif

// Windowed variable-base k*P, 4-bit fixed windows, most significant first.  The lane's table of
// 1P..15P (Jacobian, 32 words per entry: X, Y, Z, pad) lives in its own slice of a global scratch
// buffer -- written and read only by this lane, one 128-byte line per lookup.
constexpr int WIN_ENTRY_WORDS = 32, WIN_ENTRIES = 15;
#ifdef FEC_HOST_EMUL
FEC_DEV void win_store(u32* slot, const jac& p) {
  for (int i = 0; i < 8; ++i) {
    slot[i] = p.x.w[i];
    slot[8 + i] = p.y.w[i];
    slot[16 + i] = p.z.w[i];
  }
}
FEC_DEV jac win_load(const u32* slot) {
  jac p;
  for (int i = 0; i < 8; ++i) {
    p.x.w[i] = slot[i];
    p.y.w[i] = slot[8 + i];
    p.z.w[i] = slot[16 + i];
  }
  return p;
}
#else
FEC_DEV void win_store(u32* slot, const jac& p) {
  uint4* d = reinterpret_cast<uint4*>(slot);
  d[0] = make_uint4(p.x.w[0], p.x.w[1], p.x.w[2], p.x.w[3]);
  d[1] = make_uint4(p.x.w[4], p.x.w[5], p.x.w[6], p.x.w[7]);
  d[2] = make_uint4(p.y.w[0], p.y.w[1], p.y.w[2], p.y.w[3]);
  d[3] = make_uint4(p.y.w[4], p.y.w[5], p.y.w[6], p.y.w[7]);
  d[4] = make_uint4(p.z.w[0], p.z.w[1], p.z.w[2], p.z.w[3]);
  d[5] = make_uint4(p.z.w[4], p.z.w[5], p.z.w[6], p.z.w[7]);
}
FEC_DEV jac win_load(const u32* slot) {
  const uint4* s = reinterpret_cast<const uint4*>(slot);
  uint4 a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5];
  jac p;
  p.x.w[0] = a.x; p.x.w[1] = a.y; p.x.w[2] = a.z; p.x.w[3] = a.w;
  p.x.w[4] = b.x; p.x.w[5] = b.y; p.x.w[6] = b.z; p.x.w[7] = b.w;
  p.y.w[0] = c.x; p.y.w[1] = c.y; p.y.w[2] = c.z; p.y.w[3] = c.w;
  p.y.w[4] = d.x; p.y.w[5] = d.y; p.y.w[6] = d.z; p.y.w[7] = d.w;
  p.z.w[0] = e.x; p.z.w[1] = e.y; p.z.w[2] = e.z; p.z.w[3] = e.w;
  p.z.w[4] = f.x; p.z.w[5] = f.y; p.z.w[6] = f.z; p.z.w[7] = f.w;
  return p;
}
#endif

// ---- batched normalisation (Montgomery's trick) ------------------------------------------------
// The scalar-multiplication kernels leave Jacobian results in memory -- X, Y in the caller's
// out_xy slots (16 words per element), Z in a side buffer (8 words per element); one lane then
// normalises NORM_GROUP of them with a single inversion:  c_j = z_0 ... z_j,  u = 1 / c_last,
// walking back  1/z_j = u * c_{j-1},  u *= z_j.   (270 + 3 (G-1)) / G + 5 multiplications per
// element instead of 275.  Elements with Z = 0 (infinity) or a rejected input take z = 1 in the
// chain and are written as zeros.
constexpr int NORM_GROUP = 8;
// status values shared with the kernels / the ABI (fec_canon_status)
constexpr unsigned char ST_FINITE = 0, ST_INFINITY = 1, ST_BAD_POINT = 2;

// Comb table for k*G with 4-bit digits: entry (i, j), j = 1..15, is the AFFINE point j * 16^i * G.
// 64 windows x 15 entries x 16 words, entry stride COMB_STRIDE words (odd, spreads LDS banks).
constexpr int COMB_WINDOWS = 64, COMB_ENTRIES = 15, COMB_STRIDE = 17;
constexpr int COMB_WORDS = COMB_WINDOWS * COMB_ENTRIES * COMB_STRIDE;
// 8-bit comb: 32 windows x 255 affine multiples of 256^w * G, 16 words (64 bytes) each, 510 KiB:
// lives in global memory and stays L2-resident; one 64-byte gather per lane per window.
constexpr int COMB8_WINDOWS = 32, COMB8_ENTRIES = 255;
constexpr size_t COMB8_WORDS = (size_t)COMB8_WINDOWS * COMB8_ENTRIES * 16;

// This lane's group: elements first, first + stride, ... (NORM_GROUP of them, those < n).
// status[i] on entry: ST_BAD_POINT for rejected inputs, anything else is recomputed here.
template <class F, bool JACOBIAN>
FEC_DEV void normalize_group_t(u32* xy, const u32* zbuf, unsigned char* status, size_t first, size_t stride,
                             size_t n) {
  fe c[NORM_GROUP];
  fe run = fe_small(1);
  FEC_UNROLL for (int j = 0; j < NORM_GROUP; ++j) {
    const size_t i = first + (size_t)j * stride;
    fe z = fe_small(1);
    if (i < n && status[i] != ST_BAD_POINT) z = ld8(zbuf + i * 8);
    z = fe_select(z, fe_small(1), fe_is_zero(z));
    run = j == 0 ? z : F::mul(run, z);
    c[j] = run;
  }
  fe u = F::inv(run);
#pragma unroll 1
  for (int j = NORM_GROUP - 1; j >= 0; --j) {
    const size_t i = first + (size_t)j * stride;
    const bool live = i < n;
    fe z = fe_small(1);
    bool bad = false;
    if (live) {
      bad = status[i] == ST_BAD_POINT;
      if (!bad) z = ld8(zbuf + i * 8);
    }
    const lmask zero = fe_is_zero(z);
    z = fe_select(z, fe_small(1), zero);
    fe prev = fe_small(1);
    FEC_UNROLL for (int t = 0; t < NORM_GROUP - 1; ++t) prev = fe_select(prev, c[t], lanes_where(t == j - 1));
    fe zi = F::mul(u, prev);  // 1 / z_j
    u = F::mul(u, z);
    if (live) {
      fe x = ld8(xy + i * 16), y = ld8(xy + i * 16 + 8);
      if (JACOBIAN) {  // x = X / Z^2, y = Y / Z^3
        fe zi2 = F::sqr(zi);
        x = F::mul(x, zi2);
        y = F::mul(F::mul(y, zi2), zi);
      } else {         // x = X / Z, y = Y / Z
        x = F::mul(x, zi);
        y = F::mul(y, zi);
      }
      const lmask wipe = zero | lanes_where(bad);
      x = fe_select(x, fe_zero(), wipe);
      y = fe_select(y, fe_zero(), wipe);
      st8(xy + i * 16, x);
      st8(xy + i * 16 + 8, y);
      status[i] = bad ? ST_BAD_POINT : (lane_of(zero) ? ST_INFINITY : ST_FINITE);
    }
  }
}

// ---- short Weierstrass curve y^2 = x^3 + a x + b with a in {0, -3}, Jacobian coordinates --------
// P supplies: F (the field), A_IS_ZERO, b(), generator().
template <class P>
struct wei {
  using F = typename P::F;
  static constexpr bool P_HAS_GLV = P::HAS_GLV;
  FEC_SDEV fe add(const fe& a, const fe& b) { return F::add(a, b); }
  FEC_SDEV fe sub(const fe& a, const fe& b) { return F::sub(a, b); }
  FEC_SDEV fe neg(const fe& a) { return F::neg(a); }
  FEC_SDEV fe dbl(const fe& a) { return F::add(a, a); }
  FEC_SDEV fe mul(const fe& a, const fe& b) { return F::mul(a, b); }
  FEC_SDEV fe sqr(const fe& a) { return F::sqr(a); }
  FEC_SDEV fe inv(const fe& a) { return F::inv(a); }
  FEC_SDEV fe mul3(const fe& a) { return add(add(a, a), a); }
  FEC_SDEV fe mul8(const fe& a) { return dbl(dbl(dbl(a))); }
  FEC_SDEV lmask ge_p(const fe& a) { return F::ge_p(a); }
  FEC_SDEV aff generator() { return P::generator(); }

// Doubling.  a = 0 (dbl-2009-l): 2M + 5S; a = -3 (dbl-2001-b): 3M + 5S.  Infinity (Z = 0) doubles
// to Z3 = 0; Y = 0 cannot occur on these curves (odd prime order).
FEC_SDEV jac jdouble(const jac& p) {
  jac r;
  if (P::A_IS_ZERO) {
    fe A = sqr(p.x), B = sqr(p.y), C = sqr(B);
    fe t = sqr(add(p.x, B));
    fe D = dbl(sub(sub(t, A), C));
    fe E = mul3(A);
    fe Fq = sqr(E);
    r.x = sub(Fq, dbl(D));
    r.y = sub(mul(E, sub(D, r.x)), mul8(C));
    r.z = dbl(mul(p.y, p.z));
  } else {
    fe delta = sqr(p.z), gamma = sqr(p.y);
    fe beta = mul(p.x, gamma);
    fe alpha = mul3(mul(sub(p.x, delta), add(p.x, delta)));
    r.x = sub(sqr(alpha), mul8(beta));
    r.z = sub(sub(sqr(add(p.y, p.z)), gamma), delta);
    r.y = sub(mul(alpha, sub(dbl(dbl(beta)), r.x)), mul8(sqr(gamma)));
  }
  return r;
}

// madd-2007-bl: Jacobian + affine, 7M + 4S, with the exceptional cases (P infinite; P == +-Q)
// behind wave-uniform branches.  `skip` lanes return p unchanged (a zero comb digit).
FEC_SDEV jac jadd_affine(const jac& p, const aff& q, lmask skip) {
  fe z1z1 = sqr(p.z);
  fe u2 = mul(q.x, z1z1);
  fe s2 = mul(mul(q.y, p.z), z1z1);
  fe h = sub(u2, p.x);
  fe hh = sqr(h);
  fe i = dbl(dbl(hh));
  fe j = mul(h, i);
  fe rr = dbl(sub(s2, p.y));
  fe v = mul(p.x, i);
  jac o;
  o.x = sub(sub(sqr(rr), j), dbl(v));
  o.y = sub(mul(rr, sub(v, o.x)), dbl(mul(p.y, j)));
  o.z = sub(sub(sqr(add(p.z, h)), z1z1), hh);
  lmask pinf = fe_is_zero(p.z);
  lmask hzero = fe_is_zero(h) & ~pinf;
  if (__builtin_expect((pinf | hzero) != 0, 0)) {
    jac qa;
    qa.x = q.x;
    qa.y = q.y;
    qa.z = fe_small(1);
    o = jac_select(o, qa, pinf);
    if (hzero != 0) {  // same x: either P == Q (double) or P == -Q (infinity)
      lmask same = hzero & fe_is_zero(rr);
      jac d = jdouble(qa);
      o = jac_select(o, jac_infinity(), hzero & ~same);
      o = jac_select(o, d, same);
    }
  }
  return jac_select(o, p, skip);
}

// general Jacobian + Jacobian (add-2007-bl), 11M + 5S; used only while building tables
FEC_SDEV jac jadd(const jac& p, const jac& q) {
  fe z1z1 = sqr(p.z), z2z2 = sqr(q.z);
  fe u1 = mul(p.x, z2z2), u2 = mul(q.x, z1z1);
  fe s1 = mul(mul(p.y, q.z), z2z2), s2 = mul(mul(q.y, p.z), z1z1);
  fe h = sub(u2, u1);
  fe i = sqr(dbl(h));
  fe j = mul(h, i);
  fe rr = dbl(sub(s2, s1));
  fe v = mul(u1, i);
  jac o;
  o.x = sub(sub(sqr(rr), j), dbl(v));
  o.y = sub(mul(rr, sub(v, o.x)), dbl(mul(s1, j)));
  o.z = mul(sub(sub(sqr(add(p.z, q.z)), z1z1), z2z2), h);
  lmask pinf = fe_is_zero(p.z), qinf = fe_is_zero(q.z);
  lmask hzero = fe_is_zero(h) & ~pinf & ~qinf;
  lmask same = hzero & fe_is_zero(rr);
  jac d = jdouble(p);
  o = jac_select(o, jac_infinity(), hzero & ~same);
  o = jac_select(o, d, same);
  o = jac_select(o, q, pinf);
  o = jac_select(o, p, qinf & ~pinf);
  return o;
}

// Jacobian + Jacobian for the windowed ladder: q is a table entry (never infinite); `skip` lanes
// keep p (zero digit).  P infinite / P == +-Q are fixed up behind a wave-uniform branch.
FEC_SDEV jac jadd_window(const jac& p, const jac& q, lmask skip) {
  fe z1z1 = sqr(p.z), z2z2 = sqr(q.z);
  fe u1 = mul(p.x, z2z2), u2 = mul(q.x, z1z1);
  fe s1 = mul(mul(p.y, q.z), z2z2), s2 = mul(mul(q.y, p.z), z1z1);
  fe h = sub(u2, u1);
  fe i = sqr(dbl(h));
  fe j = mul(h, i);
  fe rr = dbl(sub(s2, s1));
  fe v = mul(u1, i);
  jac o;
  o.x = sub(sub(sqr(rr), j), dbl(v));
  o.y = sub(mul(rr, sub(v, o.x)), dbl(mul(s1, j)));
  o.z = mul(sub(sub(sqr(add(p.z, q.z)), z1z1), z2z2), h);
  lmask pinf = fe_is_zero(p.z) & ~skip;
  lmask hzero = fe_is_zero(h) & ~pinf & ~skip;
  if (__builtin_expect((pinf | hzero) != 0, 0)) {
    o = jac_select(o, q, pinf);
    if (hzero != 0) {
      lmask same = hzero & fe_is_zero(rr);
      jac d = jdouble(q);
      o = jac_select(o, jac_infinity(), hzero & ~same);
      o = jac_select(o, d, same);
    }
  }
  return jac_select(o, p, skip);
}

// x, y < p and y^2 == x^3 + a x + b
FEC_SDEV lmask on_curve(const aff& q) {
  lmask xlt = ~ge_p(q.x), ylt = ~ge_p(q.y);
  fe x2 = sqr(q.x);
  if (!P::A_IS_ZERO) x2 = sub(x2, fe_small(3));
  fe rhs = add(mul(x2, q.x), P::b());
  return uniform_mask(xlt & ylt & fe_eq(sqr(q.y), rhs));
}

FEC_SDEV jac mul_window(const aff& base, const u32* kw, u32* table /* this lane's 15 x 32 words */) {
  jac t;
  t.x = base.x;
  t.y = base.y;
  t.z = fe_small(1);
  win_store(table, t);
  t = jdouble(t);
  win_store(table + WIN_ENTRY_WORDS, t);
#pragma unroll 1
  for (int j = 3; j <= WIN_ENTRIES; ++j) {
    t = jadd_affine(t, base, 0);
    win_store(table + (j - 1) * WIN_ENTRY_WORDS, t);
  }
  jac acc = jac_infinity();
#pragma unroll 1
  for (int w = 63; w >= 0; --w) {
    u32 digit = (kw[(w >> 3) * KSTRIDE] >> ((w & 7) * 4)) & 15u;
    jac q = win_load(table + ((digit == 0 ? 1u : digit) - 1) * WIN_ENTRY_WORDS);
#pragma unroll 1
    for (int d = 0; d < 4; ++d) acc = jdouble(acc);
    acc = jadd_window(acc, q, lanes_where(digit == 0));
  }
  return acc;
}

FEC_SDEV void normalize_group(u32* xy, const u32* zbuf, unsigned char* status, size_t first, size_t stride,
                               size_t n) {
  normalize_group_t<F, true>(xy, zbuf, status, first, stride, n);
}

FEC_SDEV aff comb_entry(const u32* tab, int window, u32 digit /* 1..15 */) {
  const u32* e = tab + (window * COMB_ENTRIES + (int)digit - 1) * COMB_STRIDE;
  aff q;
  FEC_UNROLL for (int i = 0; i < 8; ++i) {
    q.x.w[i] = e[i];
    q.y.w[i] = e[8 + i];
  }
  return q;
}

// Jacobian -> affine.  inv(0) = 0, so the point at infinity comes out as (0, 0) with the mask set.
FEC_SDEV lmask to_affine(const jac& p, aff& a) {
  fe zi = inv(p.z);
  fe zi2 = sqr(zi);
  a.x = mul(p.x, zi2);
  a.y = mul(mul(p.y, zi2), zi);
  return fe_is_zero(p.z);
}

// the 15 entries of one comb window: j * base for j = 1..15, base = 16^window * G (affine)
FEC_SDEV void comb_fill_window(u32* table, int window, const aff& base) {
  jac acc;
  acc.x = base.x;
  acc.y = base.y;
  acc.z = fe_small(1);
#pragma unroll 1
  for (int j = 1; j <= COMB_ENTRIES; ++j) {
    aff e = base;
    if (j > 1) {
      acc = jadd_affine(acc, base, 0);  // j == 2 goes through the P == Q branch
      to_affine(acc, e);
    }
    u32* dst = table + (size_t)(window * COMB_ENTRIES + j - 1) * COMB_STRIDE;
    FEC_UNROLL for (int w = 0; w < 8; ++w) {
      dst[w] = e.x.w[w];
      dst[8 + w] = e.y.w[w];
    }
    dst[16] = 0;
  }
}

// ---- 8-bit comb (table in global memory / L2) ----
FEC_SDEV aff comb8_entry(const u32* tab, int window, u32 digit /* 1..255 */) {
  const u32* e = tab + ((size_t)window * COMB8_ENTRIES + digit - 1) * 16;
  aff q;
  q.x = ld8(e);
  q.y = ld8(e + 8);
  return q;
}
FEC_SDEV void comb8_store(u32* tab, int window, u32 digit, const aff& a) {
  u32* e = tab + ((size_t)window * COMB8_ENTRIES + digit - 1) * 16;
  st8(e, a.x);
  st8(e + 8, a.y);
}
// j * base for one 8-bit digit j (1..255), base affine: plain double-and-add, MSB first
FEC_SDEV jac small_multiple(const aff& base, u32 j) {
  jac acc = jac_infinity();
#pragma unroll 1
  for (int b = 7; b >= 0; --b) {
    acc = jdouble(acc);
    acc = jadd_affine(acc, base, lanes_where(((j >> b) & 1u) == 0));
  }
  return acc;
}
// k*G: one mixed addition per non-zero byte of k (32 bytes), no doublings; the next entry's gather is
// issued before the current addition so its L2 latency hides behind ~11 field multiplications.
FEC_SDEV jac mul_base_comb8(const u32* tab, const u32* kw) {
  jac acc = jac_infinity();
  u32 d = kw[0] & 255u;
  aff q = comb8_entry(tab, 0, d == 0 ? 1u : d);
#pragma unroll 1
  for (int w = 0; w < COMB8_WINDOWS; ++w) {
    const aff cur = q;
    const lmask skip = lanes_where(d == 0);
    if (w + 1 < COMB8_WINDOWS) {
      d = (kw[((w + 1) >> 2) * KSTRIDE] >> (((w + 1) & 3) * 8)) & 255u;
      q = comb8_entry(tab, w + 1, d == 0 ? 1u : d);
    }
    acc = jadd_affine(acc, cur, skip);
  }
  return acc;
}

// k*G: one mixed addition per non-zero 4-bit digit of k (64 digits), no doublings.
// kw: the lane's scalar in LDS (word j at kw[j * KSTRIDE]); any 256-bit k is accepted (k*G with k
// taken modulo the group order, as the group law gives).
FEC_SDEV jac mul_base_comb(const u32* tab, const u32* kw) {
  jac acc = jac_infinity();
#pragma unroll 1
  for (int w = 0; w < COMB_WINDOWS; ++w) {
    u32 digit = (kw[(w >> 3) * KSTRIDE] >> ((w & 7) * 4)) & 15u;
    lmask skip = lanes_where(digit == 0);
    aff q = comb_entry(tab, w, digit == 0 ? 1u : digit);
    acc = jadd_affine(acc, q, skip);
  }
  return acc;
}
};  // struct wei

struct SecpParams {
  using F = FpSecp;
  static constexpr bool A_IS_ZERO = true;
  static constexpr bool HAS_GLV = true;  // phi(x, y) = (beta x, y): glv_secp below
  FEC_SDEV fe b() { return fe_small(7); }
  FEC_SDEV aff generator() {  // SEC 2, section 2.4.1
    aff g;
    const u32 gx[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu, 0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
    const u32 gy[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u, 0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
    FEC_UNROLL for (int i = 0; i < 8; ++i) {
      g.x.w[i] = gx[i];
      g.y.w[i] = gy[i];
    }
    return g;
  }
};
struct P256Params {
  using F = FpP256;
  static constexpr bool A_IS_ZERO = false;  // a = -3
  static constexpr bool HAS_GLV = false;
  FEC_SDEV fe b() {  // FIPS 186-4 D.1.2.3
    fe r;
    const u32 w[8] = {0x27D2604Bu, 0x3BCE3C3Eu, 0xCC53B0F6u, 0x651D06B0u, 0x769886BCu, 0xB3EBBD55u, 0xAA3A93E7u, 0x5AC635D8u};
    FEC_UNROLL for (int i = 0; i < 8; ++i) r.w[i] = w[i];
    return r;
  }
  FEC_SDEV aff generator() {
    aff g;
    const u32 gx[8] = {0xD898C296u, 0xF4A13945u, 0x2DEB33A0u, 0x77037D81u, 0x63A440F2u, 0xF8BCE6E5u, 0xE12C4247u, 0x6B17D1F2u};
    const u32 gy[8] = {0x37BF51F5u, 0xCBB64068u, 0x6B315ECEu, 0x2BCE3357u, 0x7C0F9E16u, 0x8EE7EB4Au, 0xFE1A7F9Bu, 0x4FE342E2u};
    FEC_UNROLL for (int i = 0; i < 8; ++i) {
      g.x.w[i] = gx[i];
      g.y.w[i] = gy[i];
    }
    return g;
  }
};

// ================================================================================================
// Ed25519:  -x^2 + y^2 = 1 + d x^2 y^2  over p = 2^255 - 19  (RFC 8032), extended coordinates
// ================================================================================================
struct FpEd {
  // the reference's Add / Sub / Neg are the textbook forms and correct modulo p (ed25519.rs:458-520, 547-570)
  FEC_SDEV fe add(const fe& a, const fe& b) { return ed::add(a, b); }
  FEC_SDEV fe sub(const fe& a, const fe& b) { return ed::sub(a, b); }
  FEC_SDEV fe neg(const fe& a) { return ed::neg(a); }
  FEC_SDEV lmask ge_p(const fe& v) {  // v >= p  <=>  v + 19 reaches 2^255
    fe u;
    lmask c = add_word256(u, v, 19u);
    return c | lanes_where((u.w[7] >> 31) != 0);
  }
  // t (512 bits) mod p.  2^256 = 38: column k of lo + 38 hi is one v_mad_u64_u32, the nine columns
  // one carry chain; what is left above bit 255 (the chain's top word and bit 255 itself) folds
  // with 2^255 = 19 in a second chain.  ed::reduce (the reference's canonical reduce, correct for
  // any 256-bit value) finishes the rare lanes that end up >= p.
  FEC_SDEV fe reduce512(const u32 t[16]) {
    fe a, b;
    u32 top;
    {
      u64 q[8];
      FEC_UNROLL for (int k = 0; k < 8; ++k) q[k] = (u64)t[8 + k] * 38u + t[k];
      FEC_UNROLL for (int k = 0; k < 8; ++k) a.w[k] = (u32)q[k];
      b.w[0] = 0;
      FEC_UNROLL for (int k = 1; k < 8; ++k) b.w[k] = (u32)(q[k - 1] >> 32);
      top = (u32)(q[7] >> 32);  // <= 38
    }
    fe r;
    lmask c = add256(r, a, b);
    top += word_select(0u, 1u, c);
    // value = r + top * 2^256 = (r mod 2^255) + 19 * (2 top + bit255(r))
    const u32 fold = 19u * (2u * top + (r.w[7] >> 31));
    r.w[7] &= 0x7FFFFFFFu;
    fe v;
    add_word256(v, r, fold);  // < 2^255 + 1520: no carry out
    u32 ones = v.w[1] & v.w[2] & v.w[3] & v.w[4] & v.w[5] & v.w[6] & (v.w[7] | 0x80000000u);
    if (__builtin_expect(lanes_where(ones == 0xFFFFFFFFu || (v.w[7] >> 31) != 0) != 0, 0)) v = ed::reduce(v);
    return v;
  }
  FEC_SDEV fe mul(const fe& a, const fe& b) {
    u32 t[16];
    mul_wide(t, a, b);
    return reduce512(t);
  }
  FEC_SDEV fe sqr(const fe& a) {
    u32 t[16];
    sqr_wide(t, a);
    return reduce512(t);
  }
  FEC_SDEV fe sqr_n(fe a, int n) {
#pragma unroll 1
    for (int i = 0; i < n; ++i) a = sqr(a);
    return a;
  }
  FEC_SDEV fe inv(const fe& z) {  // z^(2^255 - 21): the usual 254 S + 11 M chain
    fe z2 = sqr(z);
    fe z9 = mul(sqr_n(z2, 2), z);
    fe z11 = mul(z9, z2);
    fe z2_5_0 = mul(sqr(z11), z9);
    fe z2_10_0 = mul(sqr_n(z2_5_0, 5), z2_5_0);
    fe z2_20_0 = mul(sqr_n(z2_10_0, 10), z2_10_0);
    fe z2_40_0 = mul(sqr_n(z2_20_0, 20), z2_20_0);
    fe z2_50_0 = mul(sqr_n(z2_40_0, 10), z2_10_0);
    fe z2_100_0 = mul(sqr_n(z2_50_0, 50), z2_50_0);
    fe z2_200_0 = mul(sqr_n(z2_100_0, 100), z2_100_0);
    fe z2_250_0 = mul(sqr_n(z2_200_0, 50), z2_50_0);
    return mul(sqr_n(z2_250_0, 5), z11);
  }
};

struct ext {
  fe x, y, z, t;
};  // x = X/Z, y = Y/Z, T = XY/Z
struct niels {
  fe ypx, ymx, t2d;
};  // affine: y + x, y - x, 2 d x y
struct pniels {
  fe ypx, ymx, z, t2d;
};  // projective: Y + X, Y - X, Z, 2 d T

// Signed 4-bit comb for k*B: window w holds j * 16^w * B for j = 1..8 as affine Niels points (24
// words, stride 25 against LDS bank conflicts); one extra entry, 2^256 * B, absorbs the carry of the
// signed recoding so that any 256-bit scalar is accepted.
constexpr int ED_COMB_ENTRIES = 8, ED_COMB_STRIDE = 25;
constexpr int ED_COMB_WORDS = (COMB_WINDOWS * ED_COMB_ENTRIES + 1) * ED_COMB_STRIDE;
// Signed 8-bit comb: 32 windows x 128 affine Niels multiples of 256^w * B (+ 2^256 * B), each padded to
// 32 words = one 128-byte line; 512 KiB in global memory, L2-resident.
constexpr int ED_COMB8_ENTRIES = 128;
constexpr size_t ED_COMB8_WORDS = ((size_t)COMB8_WINDOWS * ED_COMB8_ENTRIES + 1) * 32;
// Variable base: 1P..8P as projective Niels points, 32 words = one 128-byte line each
constexpr int ED_WIN_ENTRIES = 8;

struct edw {
  using F = FpEd;
  FEC_SDEV fe add(const fe& a, const fe& b) { return F::add(a, b); }
  FEC_SDEV fe sub(const fe& a, const fe& b) { return F::sub(a, b); }
  FEC_SDEV fe neg(const fe& a) { return F::neg(a); }
  FEC_SDEV fe mul(const fe& a, const fe& b) { return F::mul(a, b); }
  FEC_SDEV fe sqr(const fe& a) { return F::sqr(a); }
  FEC_SDEV fe inv(const fe& a) { return F::inv(a); }
  FEC_SDEV fe fe_words(u32 w0, u32 w1, u32 w2, u32 w3, u32 w4, u32 w5, u32 w6, u32 w7) {
    fe r;
    r.w[0] = w0; r.w[1] = w1; r.w[2] = w2; r.w[3] = w3; r.w[4] = w4; r.w[5] = w5; r.w[6] = w6; r.w[7] = w7;
    return r;
  }
  FEC_SDEV fe d() {  // -121665 / 121666
    return fe_words(0x135978A3u, 0x75EB4DCAu, 0x4141D8ABu, 0x00700A4Du, 0x7779E898u, 0x8CC74079u, 0x2B6FFE73u, 0x52036CEEu);
  }
  FEC_SDEV fe d2() {
    return fe_words(0x26B2F159u, 0xEBD69B94u, 0x8283B156u, 0x00E0149Au, 0xEEF3D130u, 0x198E80F2u, 0x56DFFCE7u, 0x2406D9DCu);
  }
  FEC_SDEV aff generator() {  // RFC 8032 section 5.1: y = 4/5, x even
    aff g;
    g.x = fe_words(0x8F25D51Au, 0xC9562D60u, 0x9525A7B2u, 0x692CC760u, 0xFDD6DC5Cu, 0xC0A4E231u, 0xCD6E53FEu, 0x216936D3u);
    g.y = fe_words(0x66666658u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u);
    return g;
  }
  FEC_SDEV ext identity() {
    ext p;
    p.x = fe_zero();
    p.y = fe_small(1);
    p.z = fe_small(1);
    p.t = fe_zero();
    return p;
  }
  FEC_SDEV ext from_affine(const aff& a) {
    ext p;
    p.x = a.x;
    p.y = a.y;
    p.z = fe_small(1);
    p.t = mul(a.x, a.y);
    return p;
  }
  FEC_SDEV pniels to_pniels(const ext& p) {
    pniels q;
    q.ypx = add(p.y, p.x);
    q.ymx = sub(p.y, p.x);
    q.z = p.z;
    q.t2d = mul(p.t, d2());
    return q;
  }
  // x, y < p and -x^2 + y^2 == 1 + d x^2 y^2
  FEC_SDEV lmask on_curve(const aff& q) {
    lmask xlt = ~F::ge_p(q.x), ylt = ~F::ge_p(q.y);
    fe xx = sqr(q.x), yy = sqr(q.y);
    fe lhs = sub(yy, xx);
    fe rhs = add(fe_small(1), mul(d(), mul(xx, yy)));
    return uniform_mask(xlt & ylt & fe_eq(lhs, rhs));
  }
  // Doubling (dbl-2008-hwcd).  WITH_T = false skips T3 (the next operation is another doubling).
  template <bool WITH_T>
  FEC_SDEV ext dbl(const ext& p) {
    fe xx = sqr(p.x), yy = sqr(p.y);
    fe zz = sqr(p.z);
    fe b2 = add(zz, zz);
    fe aa = sqr(add(p.x, p.y));
    fe ys = add(yy, xx);      // completed Y
    fe zd = sub(yy, xx);      // completed Z
    fe xc = sub(aa, ys);      // completed X
    fe tc = sub(b2, zd);      // completed T
    ext r;
    r.x = mul(xc, tc);
    r.y = mul(ys, zd);
    r.z = mul(zd, tc);
    r.t = WITH_T ? mul(xc, ys) : fe_zero();
    return r;
  }
  // p +- q, q an affine Niels point (madd-2008-hwcd-3, 7M).  `negate` lanes subtract; `skip` lanes keep p.
  FEC_SDEV ext add_niels(const ext& p, const niels& q, lmask negate, lmask skip) {
    fe qp = fe_select(q.ypx, q.ymx, negate), qm = fe_select(q.ymx, q.ypx, negate);
    fe a = mul(add(p.y, p.x), qp);
    fe b = mul(sub(p.y, p.x), qm);
    fe c = mul(q.t2d, p.t);
    fe dd = add(p.z, p.z);
    fe zs = add(dd, c), zm = sub(dd, c);
    fe zc = fe_select(zs, zm, negate);  // completed Z = D + C  (D - C when subtracting)
    fe tc = fe_select(zm, zs, negate);  // completed T = D - C
    fe xc = sub(a, b), yc = add(a, b);
    ext r;
    r.x = mul(xc, tc);
    r.y = mul(yc, zc);
    r.z = mul(zc, tc);
    r.t = mul(xc, yc);
    if (skip != 0) {
      r.x = fe_select(r.x, p.x, skip);
      r.y = fe_select(r.y, p.y, skip);
      r.z = fe_select(r.z, p.z, skip);
      r.t = fe_select(r.t, p.t, skip);
    }
    return r;
  }
  // p +- q, q a projective Niels point (add-2008-hwcd-3, 8M)
  FEC_SDEV ext add_pniels(const ext& p, const pniels& q, lmask negate, lmask skip) {
    fe qp = fe_select(q.ypx, q.ymx, negate), qm = fe_select(q.ymx, q.ypx, negate);
    fe a = mul(add(p.y, p.x), qp);
    fe b = mul(sub(p.y, p.x), qm);
    fe c = mul(q.t2d, p.t);
    fe zz = mul(p.z, q.z);
    fe dd = add(zz, zz);
    fe zs = add(dd, c), zm = sub(dd, c);
    fe zc = fe_select(zs, zm, negate);
    fe tc = fe_select(zm, zs, negate);
    fe xc = sub(a, b), yc = add(a, b);
    ext r;
    r.x = mul(xc, tc);
    r.y = mul(yc, zc);
    r.z = mul(zc, tc);
    r.t = mul(xc, yc);
    if (skip != 0) {
      r.x = fe_select(r.x, p.x, skip);
      r.y = fe_select(r.y, p.y, skip);
      r.z = fe_select(r.z, p.z, skip);
      r.t = fe_select(r.t, p.t, skip);
    }
    return r;
  }
  FEC_SDEV aff to_affine(const ext& p) {
    fe zi = inv(p.z);
    aff a;
    a.x = mul(p.x, zi);
    a.y = mul(p.y, zi);
    return a;
  }
  FEC_SDEV void normalize_group(u32* xy, const u32* zbuf, unsigned char* status, size_t first, size_t stride,
                                 size_t n) {
    normalize_group_t<F, false>(xy, zbuf, status, first, stride, n);
  }

  // ---- fixed base ----
  FEC_SDEV void comb_store(u32* table, int index, const aff& a) {
    u32* dst = table + (size_t)index * ED_COMB_STRIDE;
    fe ypx = add(a.y, a.x), ymx = sub(a.y, a.x), t2d = mul(mul(a.x, a.y), d2());
    FEC_UNROLL for (int w = 0; w < 8; ++w) {
      dst[w] = ypx.w[w];
      dst[8 + w] = ymx.w[w];
      dst[16 + w] = t2d.w[w];
    }
    dst[24] = 0;
  }
  // the 8 entries of one window: j * base for j = 1..8 (base = 16^window * B, affine)
  FEC_SDEV void comb_fill_window(u32* table, int window, const aff& base) {
    const pniels bq = to_pniels(from_affine(base));
    ext acc = from_affine(base);
#pragma unroll 1
    for (int j = 1; j <= ED_COMB_ENTRIES; ++j) {
      aff e = base;
      if (j > 1) {
        acc = add_pniels(acc, bq, 0, 0);
        e = to_affine(acc);
      }
      comb_store(table, window * ED_COMB_ENTRIES + j - 1, e);
    }
  }
  FEC_SDEV niels comb_entry(const u32* tab, int index) {
    const u32* e = tab + (size_t)index * ED_COMB_STRIDE;
    niels q;
    FEC_UNROLL for (int i = 0; i < 8; ++i) {
      q.ypx.w[i] = e[i];
      q.ymx.w[i] = e[8 + i];
      q.t2d.w[i] = e[16 + i];
    }
    return q;
  }
  // k*B for any 256-bit k: signed digits d_w in -8..8 with k = sum d_w 16^w + carry * 2^256
  FEC_SDEV ext mul_base_comb(const u32* tab, const u32* kw) {
    ext acc = identity();
    u32 carry = 0;
#pragma unroll 1
    for (int w = 0; w < COMB_WINDOWS; ++w) {
      u32 v = ((kw[(w >> 3) * KSTRIDE] >> ((w & 7) * 4)) & 15u) + carry;  // 0..16
      carry = v > 8u ? 1u : 0u;
      const u32 mag = carry ? 16u - v : v;                                // 0..8
      niels q = comb_entry(tab, w * ED_COMB_ENTRIES + (int)(mag == 0 ? 1u : mag) - 1);
      acc = add_niels(acc, q, lanes_where(carry != 0 && mag != 0), lanes_where(mag == 0));
    }
    niels top = comb_entry(tab, COMB_WINDOWS * ED_COMB_ENTRIES);
    return add_niels(acc, top, 0, lanes_where(carry == 0));
  }

  // ---- fixed base, signed 8-bit comb in global memory ----
  FEC_SDEV void comb8_store(u32* table, size_t index, const aff& a) {
    u32* dst = table + index * 32;
    st8(dst, add(a.y, a.x));
    st8(dst + 8, sub(a.y, a.x));
    st8(dst + 16, mul(mul(a.x, a.y), d2()));
  }
  FEC_SDEV niels comb8_entry(const u32* tab, size_t index) {
    const u32* e = tab + index * 32;
    niels q;
    q.ypx = ld8(e);
    q.ymx = ld8(e + 8);
    q.t2d = ld8(e + 16);
    return q;
  }
  // j * base for j in 1..128: double-and-add over 8 bits
  FEC_SDEV ext small_multiple(const aff& base, u32 j) {
    const pniels bq = to_pniels(from_affine(base));
    ext acc = identity();
#pragma unroll 1
    for (int b = 7; b >= 0; --b) {
      acc = dbl<true>(acc);
      acc = add_pniels(acc, bq, 0, lanes_where(((j >> b) & 1u) == 0));
    }
    return acc;
  }
  // k*B for any 256-bit k: signed bytes d_w in -128..128, k = sum d_w 256^w + carry * 2^256
  FEC_SDEV ext mul_base_comb8(const u32* tab, const u32* kw) {
    ext acc = identity();
    u32 v = kw[0] & 255u;
    u32 carry = v > 128u ? 1u : 0u;
    u32 mag = carry ? 256u - v : v;
    niels q = comb8_entry(tab, (size_t)(mag == 0 ? 1u : mag) - 1);
#pragma unroll 1
    for (int w = 0; w < COMB8_WINDOWS; ++w) {
      const niels cur = q;
      const lmask negate = lanes_where(carry != 0 && mag != 0), skip = lanes_where(mag == 0);
      if (w + 1 < COMB8_WINDOWS) {
        v = ((kw[((w + 1) >> 2) * KSTRIDE] >> (((w + 1) & 3) * 8)) & 255u) + carry;  // 0..256
        carry = v > 128u ? 1u : 0u;
        mag = carry ? 256u - v : v;
        q = comb8_entry(tab, (size_t)(w + 1) * ED_COMB8_ENTRIES + (mag == 0 ? 1u : mag) - 1);
      } else {
        q = comb8_entry(tab, (size_t)COMB8_WINDOWS * ED_COMB8_ENTRIES);  // 2^256 * B
      }
      acc = add_niels(acc, cur, negate, skip);
    }
    return add_niels(acc, q, 0, lanes_where(carry == 0));
  }

  // ---- variable base: signed 4-bit windows, MSB first; the lane's 1P..8P live in its scratch slice ----
  FEC_SDEV void pn_store(u32* slot, const pniels& q) {
    st8(slot, q.ypx);
    st8(slot + 8, q.ymx);
    st8(slot + 16, q.z);
    st8(slot + 24, q.t2d);
  }
  FEC_SDEV pniels pn_load(const u32* slot) {
    pniels q;
    q.ypx = ld8(slot);
    q.ymx = ld8(slot + 8);
    q.z = ld8(slot + 16);
    q.t2d = ld8(slot + 24);
    return q;
  }
  FEC_SDEV ext mul_window(const aff& base, const u32* kw, u32* table /* this lane's 8 x 32 words */) {
    ext t = from_affine(base);
    const pniels b1 = to_pniels(t);
    pn_store(table, b1);
#pragma unroll 1
    for (int j = 2; j <= ED_WIN_ENTRIES; ++j) {
      t = add_pniels(t, b1, 0, 0);
      pn_store(table + (j - 1) * 32, to_pniels(t));
    }
    // signed recoding needs the carry from below: digits are produced LSB first, consumed MSB first,
    // so recode into a 64 x 5-bit word array first (sign bit 4 + magnitude 0..8)
    u32 dig[8];  // 8 digits per word, 4 bits magnitude... magnitude needs 0..8 -> 4 bits; signs kept apart
    u32 sgn[2];
    sgn[0] = sgn[1] = 0;
    u32 carry = 0;
    FEC_UNROLL for (int wd = 0; wd < 8; ++wd) {
      const u32 word = kw[wd * KSTRIDE];
      u32 packed = 0;
      FEC_UNROLL for (int nb = 0; nb < 8; ++nb) {
        u32 v = ((word >> (4 * nb)) & 15u) + carry;
        carry = v > 8u ? 1u : 0u;
        const u32 mag = carry ? 16u - v : v;  // 0..8; 8 only without carry... (v = 8) -> fits 4 bits
        packed |= mag << (4 * nb);
        sgn[wd >> 2] |= carry << ((wd & 3) * 8 + nb);
      }
      dig[wd] = packed;
    }
    // acc = carry * P (the 2^256 term, doubled 256 times by the loop below)
    ext acc = identity();
    {
      pniels q = pn_load(table);
      acc = add_pniels(acc, q, 0, lanes_where(carry == 0));
    }
#pragma unroll 1
    for (int w = 63; w >= 0; --w) {
      u32 mag = 0, neg = 0;
      FEC_UNROLL for (int wd = 0; wd < 8; ++wd) {   // select the digit word without dynamic register indexing
        if (wd == (w >> 3)) {
          mag = (dig[wd] >> ((w & 7) * 4)) & 15u;
          neg = (sgn[wd >> 2] >> ((wd & 3) * 8 + (w & 7))) & 1u;
        }
      }
      pniels q = pn_load(table + ((mag == 0 ? 1u : mag) - 1) * 32);
#pragma unroll 1
      for (int dd = 0; dd < 3; ++dd) acc = dbl<false>(acc);
      acc = dbl<true>(acc);
      acc = add_pniels(acc, q, lanes_where(neg != 0), lanes_where(mag == 0));
    }
    return acc;
  }
};

// ================================================================================================
// Scalar fields (integers modulo the group order n) for ECDSA verification: Montgomery arithmetic
// with R = 2^256 for a general odd modulus.  N supplies n(), nm2() (n - 2), N0INV = -n^-1 mod 2^32,
// r2() = R^2 mod n.
// ================================================================================================
template <class N>
struct Fn {
  FEC_SDEV lmask ge_n(const fe& a) {
    fe d;
    return ~sub256(d, a, N::n());  // no borrow  <=>  a >= n
  }
  // t (512 bits, < n * 2^256) * R^-1 mod n
  FEC_SDEV fe redc(const u32 tin[16]) {
    const fe nn = N::n();
    u32 t[16];
    FEC_UNROLL for (int i = 0; i < 16; ++i) t[i] = tin[i];
    u32 extra = 0;
    FEC_UNROLL for (int k = 0; k < 8; ++k) {
      const u32 m = t[k] * N::N0INV;
      u64 c = 0;
      FEC_UNROLL for (int j = 0; j < 8; ++j) {
        u64 a = (u64)m * nn.w[j] + t[k + j] + c;  // <= (2^32-1)^2 + 2 (2^32-1): fits
        t[k + j] = (u32)a;
        c = a >> 32;
      }
      u64 a = (u64)t[k + 8] + c + extra;
      t[k + 8] = (u32)a;
      extra = (u32)(a >> 32);
    }
    fe r, d;
    FEC_UNROLL for (int i = 0; i < 8; ++i) r.w[i] = t[8 + i];
    lmask borrow = sub256(d, r, nn);
    // value = extra * 2^256 + r < 2n: subtract n when extra is set or r >= n
    return fe_select(r, d, lanes_where(extra != 0) | ~borrow);
  }
  FEC_SDEV fe mmul(const fe& a, const fe& b) {
    u32 t[16];
    mul_wide(t, a, b);
    return redc(t);
  }
  FEC_SDEV fe msqr(const fe& a) {
    u32 t[16];
    sqr_wide(t, a);
    return redc(t);
  }
  // (sm)^-1 for sm in Montgomery form, result in Montgomery form: sm^(n-2) with Montgomery products
  FEC_SDEV fe inv_mm(const fe& sm) {
    const fe e = N::nm2();
    fe r = sm;
    int i = 255;
    while (i > 0 && !((e.w[i >> 5] >> (i & 31)) & 1u)) --i;  // leading one of the exponent: r = sm
#pragma unroll 1
    for (--i; i >= 0; --i) {
      r = msqr(r);
      if ((e.w[i >> 5] >> (i & 31)) & 1u) r = mmul(r, sm);
    }
    return r;
  }
  // s^-1 in Montgomery form for s plain, in [1, n)
  FEC_SDEV fe inv_mont(const fe& s) { return inv_mm(mmul(s, N::r2())); }
  // a + b, a - b modulo n for a, b in [0, n)
  FEC_SDEV fe addn(const fe& a, const fe& b) {
    fe s, d;
    const lmask carry = add256(s, a, b);
    const lmask borrow = sub256(d, s, N::n());
    return fe_select(s, d, carry | ~borrow);
  }
  FEC_SDEV fe subn(const fe& a, const fe& b) {
    fe d, d2;
    const lmask borrow = sub256(d, a, b);
    add256(d2, d, N::n());
    return fe_select(d, d2, borrow);
  }
};

struct NSecp {
  static constexpr u32 N0INV = 0x5588B13Fu;
  FEC_SDEV fe n() { return edw::fe_words(0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); }
  FEC_SDEV fe nm2() { return edw::fe_words(0xD036413Fu, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); }
  FEC_SDEV fe r2() { return edw::fe_words(0x67D7D140u, 0x896CF214u, 0x0E7CF878u, 0x741496C2u, 0x5BCD07C6u, 0xE697F5E4u, 0x81C69BC5u, 0x9D671CD5u); }
};
struct NEd {  // l = 2^252 + 27742317777372353535851937790883648493, the order of the Ed25519 base point
  static constexpr u32 N0INV = 0x12547E1Bu;
  FEC_SDEV fe n() { return edw::fe_words(0x5CF5D3EDu, 0x5812631Au, 0xA2F79CD6u, 0x14DEF9DEu, 0u, 0u, 0u, 0x10000000u); }
  FEC_SDEV fe nm2() { return edw::fe_words(0x5CF5D3EBu, 0x5812631Au, 0xA2F79CD6u, 0x14DEF9DEu, 0u, 0u, 0u, 0x10000000u); }
  FEC_SDEV fe r2() { return edw::fe_words(0x449C0F01u, 0xA40611E3u, 0x68859347u, 0xD00E1BA7u, 0x17F5BE65u, 0xCEEC73D2u, 0x7C309A3Du, 0x0399411Bu); }
};
struct NP256 {
  static constexpr u32 N0INV = 0xEE00BC4Fu;
  FEC_SDEV fe n() { return edw::fe_words(0xFC632551u, 0xF3B9CAC2u, 0xA7179E84u, 0xBCE6FAADu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0xFFFFFFFFu); }
  FEC_SDEV fe nm2() { return edw::fe_words(0xFC63254Fu, 0xF3B9CAC2u, 0xA7179E84u, 0xBCE6FAADu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0xFFFFFFFFu); }
  FEC_SDEV fe r2() { return edw::fe_words(0xBE79EEA2u, 0x83244C95u, 0x49BD6FA6u, 0x4699799Cu, 0x2B6BEC59u, 0x2845B239u, 0xF3D95620u, 0x66E12D94u); }
};

// ================================================================================================
// GLV for secp256k1 (Gallant-Lambert-Vanstone): phi(x, y) = (beta x, y) = lambda (x, y), so
//   k P = k1 P + k2 phi(P),  k = k1 + k2 lambda (mod n),  |k1|, |k2| < 2^128,
// and the windowed ladder needs 33 windows of (4 doublings + 2 additions) instead of 64 of (4 + 1).
// Decomposition as in Guide to ECC Alg. 3.74 with the precomputed basis (a1, b1), (a2, b2):
//   c1 = round(b2 k / n) = (k g1 + 2^383) >> 384,  c2 = round(-b1 k / n) = (k g2 + 2^383) >> 384,
//   k2 = c1 (-b1) + c2 (-b2),  k1 = k - k2 lambda   (mod n),  each then taken with its sign.
// ================================================================================================
struct glv_secp {
  using W = wei<SecpParams>;
  using F = Fn<NSecp>;
  FEC_SDEV fe beta() { return edw::fe_words(0x719501EEu, 0xC1396C28u, 0x12F58995u, 0x9CF04975u, 0xAC3434E9u, 0x6E64479Eu, 0x657C0710u, 0x7AE96A2Bu); }
  FEC_SDEV fe g1() { return edw::fe_words(0x45DBB031u, 0xE893209Au, 0x71E8CA7Fu, 0x3DAA8A14u, 0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u); }
  FEC_SDEV fe g2() { return edw::fe_words(0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu, 0x0ABFE4C4u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u); }
  // (-b1) R, (-b2) R, lambda R modulo n: plain * Montgomery-form constant = plain product mod n
  FEC_SDEV fe mb1R() { return edw::fe_words(0x0AD9263Cu, 0xC50468D0u, 0xFAA6ED42u, 0x1B1C8205u, 0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu); }
  FEC_SDEV fe mb2R() { return edw::fe_words(0x6A144696u, 0x0CAC5E50u, 0xF3BA5939u, 0x1E8A8DC5u, 0xBA244FCEu, 0x176CDF65u, 0x8E173580u, 0xC25575EBu); }
  FEC_SDEV fe lamR() { return edw::fe_words(0xC9926C9Eu, 0xF07DEB3Du, 0x83C6944Cu, 0x2C93E7ADu, 0x52697D91u, 0x73A96606u, 0x8558D639u, 0x53284017u); }
  FEC_SDEV fe half_n() { return edw::fe_words(0x681B20A0u, 0xDFE92F46u, 0x57A4501Du, 0x5D576E73u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu); }

  // (k g + 2^383) >> 384: the top four words of the 512-bit product, rounded
  FEC_SDEV fe mul_shift384(const fe& k, const fe& g) {
    u32 t[16];
    mul_wide(t, k, g);
    fe c = fe_zero();
    c.w[0] = t[12]; c.w[1] = t[13]; c.w[2] = t[14]; c.w[3] = t[15];
    fe r;
    add_word256(r, c, t[11] >> 31);
    return r;
  }
  // any 256-bit k -> |k1|, |k2| and the lanes where each is negative
  FEC_SDEV void decompose(const fe& kin, fe& k1, lmask& neg1, fe& k2, lmask& neg2) {
    fe k = kin, d;
    const lmask borrow = sub256(d, k, NSecp::n());
    k = fe_select(d, k, borrow);  // k mod n (k < 2^256 < 2n)
    const fe c1 = mul_shift384(k, g1()), c2 = mul_shift384(k, g2());
    const fe r2 = F::addn(F::mmul(c1, mb1R()), F::mmul(c2, mb2R()));
    const fe r1 = F::subn(k, F::mmul(r2, lamR()));
    const fe h = half_n();
    fe t;
    neg1 = uniform_mask(sub256(t, h, r1));  // r1 > n/2
    neg2 = uniform_mask(sub256(t, h, r2));
    fe m1, m2;
    sub256(m1, NSecp::n(), r1);
    sub256(m2, NSecp::n(), r2);
    k1 = fe_select(r1, m1, neg1);
    k2 = fe_select(r2, m2, neg2);
  }
  FEC_SDEV jac entry(const u32* table, u32 digit, lmask negate, bool endo) {
    jac q = win_load(table + ((digit == 0 ? 1u : digit) - 1) * WIN_ENTRY_WORDS);
    if (endo) q.x = W::mul(q.x, beta());
    q.y = fe_select(q.y, W::neg(q.y), negate);
    return q;
  }
  // kw: this lane's scalar column in LDS (8 words at stride KSTRIDE); it is overwritten with the low
  // four words of |k1| (words 0..3) and |k2| (words 4..7).
  FEC_SDEV jac mul_window(const aff& base, u32* kw, u32* table) {
    jac t;
    t.x = base.x;
    t.y = base.y;
    t.z = fe_small(1);
    win_store(table, t);
    t = W::jdouble(t);
    win_store(table + WIN_ENTRY_WORDS, t);
#pragma unroll 1
    for (int j = 3; j <= WIN_ENTRIES; ++j) {
      t = W::jadd_affine(t, base, 0);
      win_store(table + (j - 1) * WIN_ENTRY_WORDS, t);
    }
    fe k, k1, k2;
    FEC_UNROLL for (int i = 0; i < 8; ++i) k.w[i] = kw[i * KSTRIDE];
    lmask neg1, neg2;
    decompose(k, k1, neg1, k2, neg2);
    FEC_UNROLL for (int i = 0; i < 4; ++i) {
      kw[i * KSTRIDE] = k1.w[i];
      kw[(4 + i) * KSTRIDE] = k2.w[i];
    }
    // window 32: the bits above 2^128 (zero in practice; kept for the proven bound |k_i| < 2^128 + small)
    jac acc = jac_infinity();
    {
      const u32 d1 = k1.w[4] & 15u, d2 = k2.w[4] & 15u;
      if (__builtin_expect(lanes_where((d1 | d2) != 0) != 0, 0)) {
        acc = W::jadd_window(acc, entry(table, d1, neg1, false), lanes_where(d1 == 0));
        acc = W::jadd_window(acc, entry(table, d2, neg2, true), lanes_where(d2 == 0));
      }
    }
#pragma unroll 1
    for (int w = 31; w >= 0; --w) {
      const u32 d1 = (kw[(w >> 3) * KSTRIDE] >> ((w & 7) * 4)) & 15u;
      const u32 d2 = (kw[(4 + (w >> 3)) * KSTRIDE] >> ((w & 7) * 4)) & 15u;
#pragma unroll 1
      for (int d = 0; d < 4; ++d) acc = W::jdouble(acc);
      acc = W::jadd_window(acc, entry(table, d1, neg1, false), lanes_where(d1 == 0));
      acc = W::jadd_window(acc, entry(table, d2, neg2, true), lanes_where(d2 == 0));
    }
    return acc;
  }
};

// Scalar-field operations for callers (signing, key derivation): any 256-bit inputs, results in [0, n).
//   op 0: a * b + c (mod n)      op 1: a^-1 (mod n), 0 for a = 0 (mod n)
template <class N>
FEC_DEV fe scalar_op(int op, const fe& a, const fe& b, const fe& c) {
  using F = Fn<N>;
  const fe am = F::mmul(a, N::r2());            // a R mod n: the product a * R2 is below n * 2^256 for any a
  if (op == 0) {
    const fe ab = F::mmul(am, b);               // (a R) b R^-1 = a b mod n, b any 256-bit value
    const fe cr = F::mmul(F::mmul(c, N::r2()), fe_small(1));  // c mod n
    return F::addn(ab, cr);
  }
  return F::mmul(F::inv_mm(am), fe_small(1));   // (a R)^-1 R, out of Montgomery form
}

// ECDSA verification (FIPS 186-4 section 6.4 / SEC 1 section 4.1.4), the scalar half:
//   r, s in [1, n-1]; w = s^-1; u1 = z w, u2 = r w  (mod n).   Returns the lanes that pass the range check.
template <class N>
FEC_DEV lmask ecdsa_scalars(const fe& z, const fe& r, const fe& s, fe& u1, fe& u2) {
  using F = Fn<N>;
  const lmask ok = ~fe_is_zero(r) & ~fe_is_zero(s) & ~F::ge_n(r) & ~F::ge_n(s);
  const fe wm = F::inv_mont(s);   // s = 0 gives 0: harmless, the lane is rejected anyway
  u1 = F::mmul(z, wm);            // plain * Montgomery = plain; z >= n is reduced by the product
  u2 = F::mmul(r, wm);
  return uniform_mask(ok);
}
// The same for a group of NORM_GROUP signatures per lane with ONE inversion (Montgomery's trick on
// the s values): elements first, first + stride, ...; rejected (out-of-range) signatures take s = 1
// in the chain.  ok[i] = 1 where r, s are in [1, n-1].
template <class N>
FEC_DEV void ecdsa_scalars_group(const u32* zs, const u32* rs, const u32* ss, u32* u1, u32* u2, unsigned char* ok,
                                 size_t first, size_t stride, size_t n) {
  using F = Fn<N>;
  fe c[NORM_GROUP];
  fe run = fe_small(1);
  FEC_UNROLL for (int j = 0; j < NORM_GROUP; ++j) {
    const size_t i = first + (size_t)j * stride;
    fe s = fe_small(1);
    if (i < n) s = ld8(ss + i * 8);
    const lmask bad = uniform_mask(fe_is_zero(s) | F::ge_n(s));
    s = fe_select(s, fe_small(1), bad);
    const fe sm = F::mmul(s, N::r2());
    run = j == 0 ? sm : F::mmul(run, sm);
    c[j] = run;
  }
  fe u = F::inv_mm(run);
#pragma unroll 1
  for (int j = NORM_GROUP - 1; j >= 0; --j) {
    const size_t i = first + (size_t)j * stride;
    const bool live = i < n;
    fe s = fe_small(1), r = fe_small(1), z = fe_zero();
    if (live) {
      s = ld8(ss + i * 8);
      r = ld8(rs + i * 8);
      z = ld8(zs + i * 8);
    }
    const lmask sbad = uniform_mask(fe_is_zero(s) | F::ge_n(s));
    const lmask good = uniform_mask(~sbad & ~fe_is_zero(r) & ~F::ge_n(r));
    s = fe_select(s, fe_small(1), sbad);
    fe prev = F::mmul(fe_small(1), N::r2());  // 1 in Montgomery form (j = 0)
    FEC_UNROLL for (int t = 0; t < NORM_GROUP - 1; ++t) prev = fe_select(prev, c[t], lanes_where(t == j - 1));
    const fe wm = F::mmul(u, prev);          // s_j^-1, Montgomery form
    u = F::mmul(u, F::mmul(s, N::r2()));
    if (live) {
      st8(u1 + i * 8, F::mmul(z, wm));       // plain * Montgomery = plain; z >= n is reduced by the product
      st8(u2 + i * 8, F::mmul(r, wm));
      ok[i] = lane_of(good) ? 1 : 0;
    }
  }
}
// ... and the final comparison: x(R) mod n == r  (x < p < 2n)
template <class N>
FEC_DEV lmask ecdsa_x_matches(const fe& x, const fe& r) {
  using F = Fn<N>;
  fe d;
  sub256(d, x, N::n());
  const fe xr = fe_select(x, d, F::ge_n(x));
  return fe_eq(xr, r);
}

// ================================================================================================
// Signature verification beyond ECDSA: BIP-340 Schnorr (secp256k1) and EdDSA (Ed25519, RFC 8032)
// ================================================================================================
// a^((p+1)/4) for secp256k1: the square root when one exists (p = 3 mod 4); 253 S + 13 M
FEC_DEV fe secp_sqrt_candidate(const fe& a) {
  using F = FpSecp;
  fe x2 = F::mul(F::sqr(a), a);
  fe x3 = F::mul(F::sqr(x2), a);
  fe x6 = F::mul(F::sqr_n(x3, 3), x3);
  fe x9 = F::mul(F::sqr_n(x6, 3), x3);
  fe x11 = F::mul(F::sqr_n(x9, 2), x2);
  fe x22 = F::mul(F::sqr_n(x11, 11), x11);
  fe x44 = F::mul(F::sqr_n(x22, 22), x22);
  fe x88 = F::mul(F::sqr_n(x44, 44), x44);
  fe x176 = F::mul(F::sqr_n(x88, 88), x88);
  fe x220 = F::mul(F::sqr_n(x176, 44), x44);
  fe x223 = F::mul(F::sqr_n(x220, 3), x3);
  fe t = F::mul(F::sqr_n(x223, 23), x22);
  t = F::mul(F::sqr_n(t, 6), x2);
  return F::sqr_n(t, 2);
}
// BIP-340 lift_x: the point with this x and even y; mask = lanes where x < p and x^3 + 7 is a square
FEC_DEV lmask bip340_lift_x(const fe& x, aff& out) {
  using F = FpSecp;
  const lmask in_range = ~F::ge_p(x);
  const fe c = F::add(F::mul(F::sqr(x), x), fe_small(7));
  fe y = secp_sqrt_candidate(c);
  const lmask is_root = fe_eq(F::sqr(y), c);
  y = fe_select(y, F::neg(y), lanes_where((y.w[0] & 1u) != 0));
  out.x = x;
  out.y = y;
  return uniform_mask(in_range & is_root);
}
// BIP-340 verification, the scalar / decoding half:  P = lift_x(pk), u1 = s, u2 = n - e (mod n);
// ok = pk liftable, r < p, s < n.   The point half is u1 G + u2 P; the final test is bip340_accept.
FEC_DEV lmask bip340_prepare(const fe& pkx, const fe& r, const fe& s, const fe& e, aff& P, fe& u2) {
  using Fs = Fn<NSecp>;
  const lmask lifted = bip340_lift_x(pkx, P);
  fe d, er = e;
  const lmask borrow = sub256(d, e, NSecp::n());
  er = fe_select(d, e, borrow);                        // e mod n (e < 2^256 < 2n)
  fe neg;
  sub256(neg, NSecp::n(), er);
  u2 = fe_select(neg, fe_zero(), fe_is_zero(er));      // n - e, and 0 for e = 0
  return uniform_mask(lifted & ~FpSecp::ge_p(r) & ~Fs::ge_n(s));
}
// R finite, y(R) even, x(R) == r
FEC_DEV lmask bip340_accept(const fe& x, const fe& y, const fe& r) {
  return uniform_mask(fe_eq(x, r) & lanes_where((y.w[0] & 1u) == 0));
}

// z^((p-5)/8) = z^(2^252 - 3) for p = 2^255 - 19 (the usual 251 S + 11 M chain)
FEC_DEV fe ed_pow22523(const fe& z) {
  using F = FpEd;
  fe z2 = F::sqr(z);
  fe z9 = F::mul(F::sqr_n(z2, 2), z);
  fe z11 = F::mul(z9, z2);
  fe z2_5_0 = F::mul(F::sqr(z11), z9);
  fe z2_10_0 = F::mul(F::sqr_n(z2_5_0, 5), z2_5_0);
  fe z2_20_0 = F::mul(F::sqr_n(z2_10_0, 10), z2_10_0);
  fe z2_40_0 = F::mul(F::sqr_n(z2_20_0, 20), z2_20_0);
  fe z2_50_0 = F::mul(F::sqr_n(z2_40_0, 10), z2_10_0);
  fe z2_100_0 = F::mul(F::sqr_n(z2_50_0, 50), z2_50_0);
  fe z2_200_0 = F::mul(F::sqr_n(z2_100_0, 100), z2_100_0);
  fe z2_250_0 = F::mul(F::sqr_n(z2_200_0, 50), z2_50_0);
  return F::mul(F::sqr_n(z2_250_0, 2), z);
}
// RFC 8032 section 5.1.3: decode a 32-byte point encoding (given as its little-endian 256-bit integer).
FEC_DEV lmask ed_decode(const fe& enc, aff& out) {
  using F = FpEd;
  const fe sqrtm1 = edw::fe_words(0x4A0EA0B0u, 0xC4EE1B27u, 0xAD2FE478u, 0x2F431806u, 0x3DFBD7A7u, 0x2B4D0099u, 0x4FC1DF0Bu, 0x2B832480u);
  fe y = enc;
  const u32 sign = enc.w[7] >> 31;
  y.w[7] &= 0x7FFFFFFFu;
  const lmask y_ok = ~F::ge_p(y);
  const fe yy = F::sqr(y);
  const fe u = F::sub(yy, fe_small(1));
  const fe v = F::add(F::mul(edw::d(), yy), fe_small(1));
  const fe v3 = F::mul(F::sqr(v), v);
  const fe v7 = F::mul(F::sqr(v3), v);
  fe x = F::mul(F::mul(u, v3), ed_pow22523(F::mul(u, v7)));   // candidate root of u / v
  const fe vxx = F::mul(v, F::sqr(x));
  const lmask direct = fe_eq(vxx, u), twisted = fe_eq(vxx, F::neg(u));
  x = fe_select(x, F::mul(x, sqrtm1), twisted & ~direct);
  const lmask x_zero = fe_is_zero(x);
  const lmask bad_sign = x_zero & lanes_where(sign != 0);
  x = fe_select(x, F::neg(x), lanes_where((x.w[0] & 1u) != sign));
  out.x = x;
  out.y = y;
  return uniform_mask(y_ok & (direct | twisted) & ~bad_sign);
}
// EdDSA verification, decoding half: A, R decoded; S < l; u2 = l - h (mod l)
FEC_DEV lmask eddsa_prepare(const fe& a_enc, const fe& r_enc, const fe& s, const fe& h, aff& A, aff& R, fe& u2) {
  const fe l = edw::fe_words(0x5CF5D3EDu, 0x5812631Au, 0xA2F79CD6u, 0x14DEF9DEu, 0u, 0u, 0u, 0x10000000u);
  const lmask a_ok = ed_decode(a_enc, A), r_ok = ed_decode(r_enc, R);
  fe t;
  const lmask s_lt = sub256(t, s, l);           // borrow  <=>  S < l
  const lmask h_lt = sub256(t, h, l);
  fe neg;
  sub256(neg, l, h);
  u2 = fe_select(neg, fe_zero(), fe_is_zero(h));
  return uniform_mask(a_ok & r_ok & s_lt & h_lt);
}

}  // namespace canon
using csecp = canon::wei<canon::SecpParams>;
using cp256 = canon::wei<canon::P256Params>;
using ced = canon::edw;
using cglv = canon::glv_secp;
}  // namespace fecgpu
