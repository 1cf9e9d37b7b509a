/*
 * forge_ec_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT A PRODUCT PATH)
 *
 * See forge_ec_oracle.h for scope, provenance and the parity-pinning statement.
 * Citations are relative to /root/reference/forge-ec-curves/src/.
 *
 * Style: each function restates the reference's integer operation sequence in
 * C with `unsigned __int128` standing for Rust's u128, `u64` wrap-around for
 * wrapping_* / overflowing_* and explicit carries.  Nothing is algebraically
 * simplified: the quirks ARE the specification.
 */
#include "forge_ec_oracle.h"

#include <pthread.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;
typedef __int128 i128;

typedef struct { u64 v[4]; } fe;                 /* field element, LE limbs   */
typedef struct { fe x, y, z; } jpt;              /* Jacobian X,Y,Z            */
typedef struct { fe x, y, z, t; } ept;           /* Ed25519 extended X,Y,Z,T  */

static int fe_eq(const fe* a, const fe* b) {
  return a->v[0] == b->v[0] && a->v[1] == b->v[1] && a->v[2] == b->v[2] && a->v[3] == b->v[3];
}
static int fe_is_zero(const fe* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static fe fe_small(u64 w) { fe r = {{w, 0, 0, 0}}; return r; }
static fe fe_select(const fe* a, const fe* b, int choice) { return choice ? *b : *a; } /* subtle: c ? b : a */

/* =====================================================================================
 * secp256k1  (secp256k1.rs)
 * ===================================================================================== */
static const u64 K_P[4] = {0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL,
                           0xFFFFFFFFFFFFFFFFULL};                       /* secp256k1.rs:22-23 */

/* secp256k1.rs:47-75  compare_with_p: -1 / 0 / +1, most significant limb first */
static int k_cmp_p(const u64 l[4]) {
  int result = 0;
  for (int i = 3; i >= 0; --i) {
    int decided = result != 0;
    int nr = (l[i] < K_P[i]) ? -1 : ((l[i] > K_P[i]) ? 1 : 0);
    result = decided ? result : nr;
  }
  return result;
}

/* secp256k1.rs:78-102  reduce: one conditional subtraction of p */
static void k_reduce(fe* s) {
  fe red = *s;
  u64 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u64 d1 = red.v[i] - K_P[i];
    u64 b1 = red.v[i] < K_P[i];
    u64 d2 = d1 - borrow;
    u64 b2 = d1 < borrow;
    red.v[i] = d2;
    borrow = b1 | b2;
  }
  if (k_cmp_p(s->v) >= 0) *s = red;
}

/* secp256k1.rs:353-393  Add */
static fe k_add(fe a, fe b) {
  fe r = a;
  u64 carry = 0;
  for (int i = 0; i < 4; ++i) {
    u64 sum1 = r.v[i] + b.v[i];
    u64 sum2 = sum1 + carry;
    u64 c1 = r.v[i] > ~b.v[i];
    u64 c2 = sum1 > (~(u64)0 - carry);
    r.v[i] = sum2;
    carry = c1 | c2;
  }
  fe red = r;
  u64 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u64 d1 = red.v[i] - K_P[i];
    u64 b1 = red.v[i] < K_P[i];
    u64 d2 = d1 - borrow;
    u64 b2 = d1 < borrow;
    red.v[i] = d2;
    borrow = b1 | b2;
  }
  int should = (carry > 0) || (k_cmp_p(r.v) >= 0);
  return fe_select(&r, &red, should);
}

/* secp256k1.rs:395-440  Sub */
static fe k_sub(fe a, fe b) {
  fe r = a;
  u64 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u64 d1 = r.v[i] - b.v[i];
    u64 d2 = d1 - borrow;
    u64 b1 = r.v[i] < b.v[i];
    u64 b2 = d1 < borrow;
    r.v[i] = d2;
    borrow = b1 | b2;
  }
  fe wp = r;
  u64 carry = 0;
  for (int i = 0; i < 4; ++i) {
    u64 sum1 = wp.v[i] + K_P[i];
    u64 sum2 = sum1 + carry;
    u64 c1 = wp.v[i] > ~K_P[i];
    u64 c2 = sum1 > (~(u64)0 - carry);
    wp.v[i] = sum2;
    carry = c1 | c2;
  }
  return fe_select(&r, &wp, borrow > 0);
}

/* secp256k1.rs:442-507  Mul: schoolbook + 4-round Montgomery reduction, `carry` declared
 * OUTSIDE the round loop (470) and a carry leaving t[7] is lost. */
static fe k_mul(fe a, fe b) {
  u64 t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u64 carry = 0;
    for (int j = 0; j < 4; ++j) {
      u128 prod = (u128)a.v[i] * (u128)b.v[j] + (u128)t[i + j] + (u128)carry;
      t[i + j] = (u64)prod;
      carry = (u64)(prod >> 64);
    }
    t[i + 4] = carry;
  }
  const u64 N0 = 0xD838091DD2253531ULL;                                  /* :468 */
  u64 carry = 0;                                                        /* :470 */
  for (int i = 0; i < 4; ++i) {
    u64 m = t[i] * N0;
    u128 sum = (u128)t[i] + (u128)m * (u128)K_P[0] + (u128)carry;
    carry = (u64)(sum >> 64);
    for (int j = 1; j < 4; ++j) {
      sum = (u128)t[i + j] + (u128)m * (u128)K_P[j] + (u128)carry;
      t[i + j] = (u64)sum;
      carry = (u64)(sum >> 64);
    }
    int j = i + 4;
    while (j < 8 && carry > 0) {
      u128 s2 = (u128)t[j] + (u128)carry;
      t[j] = (u64)s2;
      carry = (u64)(s2 >> 64);
      ++j;
    }
  }
  fe r = {{t[4], t[5], t[6], t[7]}};
  if (k_cmp_p(r.v) >= 0) k_reduce(&r);
  return r;
}

/* secp256k1.rs:509-539  Neg */
static fe k_neg(fe a) {
  fe r;
  u64 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u64 d1 = K_P[i] - a.v[i];
    u64 d2 = d1 - borrow;
    u64 b1 = K_P[i] < a.v[i];
    u64 b2 = d1 < borrow;
    r.v[i] = d2;
    borrow = b1 | b2;
  }
  return fe_select(&r, &a, fe_is_zero(&a));
}

/* secp256k1.rs:634-713  square: NOT Montgomery, NOT mul(x,x) */
static fe k_sqr(fe a) {
  u64 product[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {                                        /* :643-649 */
    u128 sq = (u128)a.v[i] * (u128)a.v[i];
    product[i * 2] = (u64)sq;
    product[i * 2 + 1] = (u64)(sq >> 64);
  }
  for (int i = 0; i < 4; ++i) {                                        /* :652-679 */
    for (int j = i + 1; j < 4; ++j) {
      u128 cross = (u128)a.v[i] * (u128)a.v[j];
      cross = cross * 2;                       /* u128 wrapping_mul(2): bit 127 lost */
      u64 lo = (u64)cross;
      u64 hi = (u64)(cross >> 64);
      u64 sum = product[i + j] + lo;
      int carry = sum < lo;
      product[i + j] = sum;
      u64 sumh = product[i + j + 1] + hi;      /* `carry` is NOT added here */
      int carry2 = sumh < hi;
      product[i + j + 1] = sumh;
      if (carry || carry2) {                   /* a single +1, two positions up */
        int k = i + j + 2;
        while (k < 8) {
          product[k] = product[k] + 1;
          if (product[k] != 0) break;
          ++k;
        }
      }
    }
  }
  u64 result[4] = {product[0], product[1], product[2], product[3]};
  u64 carry = 0;                                                        /* :692 */
  for (int i = 4; i < 8; ++i) {
    u64 m = product[i] * 0x1000003D1ULL;       /* low 64 bits only */
    u64 t = result[0] + m;                     /* always into result[0] */
    t = t + carry;
    result[0] = t;
    carry = (u64)(t < m) | ((u64)(t < carry) & (u64)(m != 0));          /* :698-699 */
    for (int j = 1; j < 4; ++j) {
      u64 t2 = result[j] + carry;
      result[j] = t2;
      carry = (u64)(t2 < carry);
    }
  }
  fe r = {{result[0], result[1], result[2], result[3]}};
  k_reduce(&r);
  return r;
}

/* secp256k1.rs:599-632  invert: square-and-multiply over p-2, limbs visited LS->MS, bits MS->LS */
static fe k_inv(fe a) {
  if (fe_is_zero(&a)) return fe_small(0);      /* CtOption none; value is zero */
  static const u64 e[4] = {0xFFFFFFFEFFFFFC2DULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL,
                           0xFFFFFFFFFFFFFFFFULL};
  fe result = fe_small(1);                     /* one() = raw 1, :585-593 */
  for (int i = 0; i < 4; ++i) {
    for (int j = 63; j >= 0; --j) {
      result = k_sqr(result);
      if ((e[i] >> j) & 1) result = k_mul(result, a);
    }
  }
  return result;
}

static jpt k_identity(void) {                  /* :1322-1324 */
  jpt p = {fe_small(0), fe_small(1), fe_small(0)};
  return p;
}
static int k_is_identity(const jpt* p) {       /* :1326-1340 */
  if (fe_is_zero(&p->x) && fe_is_zero(&p->y) && fe_is_zero(&p->z)) return 1;
  return fe_is_zero(&p->z);
}

/* secp256k1.rs:1502-1540  inherent ProjectivePoint::double -- the one the ladder reaches */
static jpt k_double(const jpt* p) {
  if (k_is_identity(p)) return k_identity();
  fe a = k_sqr(p->x);
  fe b = k_sqr(p->y);
  fe c = k_sqr(b);
  fe xpb = k_add(p->x, b);
  fe xpb2 = k_sqr(xpb);
  fe dd = k_sub(k_sub(xpb2, a), c);
  fe d = k_add(dd, dd);                        /* field double(): s + s, :105-109 */
  fe e = k_mul(a, fe_small(3));                /* raw 3 through the Montgomery Mul */
  fe f = k_sqr(e);
  fe x3 = k_sub(f, k_add(d, d));
  fe y3 = k_sub(k_mul(e, k_sub(d, x3)), k_mul(c, fe_small(8)));
  fe yz = k_mul(p->y, p->z);
  fe z3 = k_add(yz, yz);
  jpt r = {x3, y3, z3};
  return r;
}

/* secp256k1.rs:1375-1418  trait PointProjective::double (not on the ladder path) */
static jpt k_double_trait(const jpt* p) {
  if (k_is_identity(p)) return k_identity();
  fe xx = k_sqr(p->x);
  fe yy = k_sqr(p->y);
  fe yyyy = k_sqr(yy);
  fe xy2 = k_sqr(k_add(p->x, yy));
  fe w = k_sub(k_sub(xy2, xx), yyyy);
  fe d = k_add(w, w);
  fe e = k_mul(fe_small(3), xx);
  fe ee = k_sqr(e);
  fe x3 = k_sub(k_sub(ee, d), d);
  fe eight_yyyy = k_mul(fe_small(8), yyyy);
  fe y3 = k_sub(k_mul(e, k_sub(d, x3)), eight_yyyy);
  fe z3 = k_add(p->y, p->y);
  fe one = fe_small(1);
  if (!fe_eq(&p->z, &one)) z3 = k_mul(z3, p->z);
  jpt r = {x3, y3, z3};
  return r;
}

/* secp256k1.rs:1444-1498  Add for ProjectivePoint */
static jpt k_padd(const jpt* p, const jpt* q) {
  if (k_is_identity(p)) return *q;
  if (k_is_identity(q)) return *p;
  fe z1s = k_sqr(p->z);
  fe z2s = k_sqr(q->z);
  fe u1 = k_mul(p->x, z2s);
  fe u2 = k_mul(q->x, z1s);
  fe z1c = k_mul(z1s, p->z);
  fe z2c = k_mul(z2s, q->z);
  fe s1 = k_mul(p->y, z2c);
  fe s2 = k_mul(q->y, z1c);
  if (fe_eq(&u1, &u2)) {
    if (fe_eq(&s1, &s2)) return k_double(p);   /* resolves to the inherent double */
    return k_identity();
  }
  fe h = k_sub(u2, u1);
  fe r = k_sub(s2, s1);
  fe h2 = k_sqr(h);
  fe h3 = k_mul(h2, h);
  fe u1h2 = k_mul(u1, h2);
  fe x3 = k_sub(k_sub(k_sub(k_sqr(r), h3), u1h2), u1h2);
  fe y3 = k_sub(k_mul(r, k_sub(u1h2, x3)), k_mul(s1, h3));
  fe z3 = k_mul(k_mul(h, p->z), q->z);
  jpt o = {x3, y3, z3};
  return o;
}

/* secp256k1.rs:219-235 to_montgomery with the reference's R_SQUARED; 2608-2625 generator */
static jpt k_generator(void) {
  fe r2 = {{0x000E9F61ULL, 0x07A20000ULL, 0x00000100ULL, 0}};
  fe gx = {{0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL, 0x79BE667EF9DCBBACULL}};
  fe gy = {{0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL, 0x483ADA7726A3C465ULL}};
  jpt g = {k_mul(gx, r2), k_mul(gy, r2), fe_small(1)};
  return g;
}

/* secp256k1.rs:1342-1363 to_affine */
static int k_to_affine(const jpt* p, fe* x, fe* y) {
  if (k_is_identity(p)) { *x = fe_small(0); *y = fe_small(0); return 1; }
  fe zi = k_inv(p->z);
  fe zi2 = k_sqr(zi);
  fe zi3 = k_mul(zi2, zi);
  *x = k_mul(p->x, zi2);
  *y = k_mul(p->y, zi3);
  return 0;
}

/* secp256k1.rs:2635-2692  Curve::multiply: Montgomery ladder over the inherent (little-endian)
 * Scalar::to_bytes (1924-1933), bits MSB-first inside each byte => scalar consumed byte-reversed. */
static jpt k_multiply(const jpt* point, const u64 k[4]) {
  if (k_is_identity(point) || (k[0] | k[1] | k[2] | k[3]) == 0) return k_identity();
  unsigned char bytes[32];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) bytes[i * 8 + j] = (unsigned char)(k[i] >> (j * 8));
  jpt r0 = k_identity();
  jpt r1 = *point;
  for (int i = 0; i < 256; ++i) {
    int bit = (bytes[i / 8] >> (7 - (i % 8))) & 1;
    jpt s = k_padd(&r0, &r1);
    jpt d0 = k_double(&r0);
    jpt d1 = k_double(&r1);
    r0 = bit ? s : d0;
    r1 = bit ? d1 : s;
  }
  return r0;
}

/* ---- secp256k1 scalar field as the reference implements it, and ECDSA verify ---------------- */
/* secp256k1.rs:27-28, literally: note that the two top limbs are swapped relative to the true
 * group order (true n has 0xFFFFFFFFFFFFFFFE in limb 2); the reference uses this constant everywhere */
static const u64 K_N[4] = {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFFULL,
                           0xFFFFFFFFFFFFFFFEULL};

/* the >= n comparison written out at 1955-1958 / 2443-2449 */
static int ks_ge_n(const u64 a[4]) {
  return a[3] > K_N[3] || (a[3] == K_N[3] && a[2] > K_N[2]) ||
         (a[3] == K_N[3] && a[2] == K_N[2] && a[1] > K_N[1]) ||
         (a[3] == K_N[3] && a[2] == K_N[2] && a[1] == K_N[1] && a[0] >= K_N[0]);
}
/* secp256k1.rs:1953-1969 Scalar::reduce: one subtraction of n if >= n */
static void ks_reduce(u64 a[4]) {
  if (ks_ge_n(a)) {
    u64 borrow = 0;
    for (int i = 0; i < 4; ++i) {
      u64 d1 = a[i] - K_N[i];
      u64 b1 = a[i] < K_N[i];
      u64 d2 = d1 - borrow;
      u64 b2 = d1 < borrow;
      a[i] = d2;
      borrow = (b1 || b2) ? 1 : 0;
    }
  }
}
/* secp256k1.rs:2410-2456 Mul for Scalar: exact 512-bit product, of which ONLY the low 256 bits are
 * kept, then reduce() and a while-loop of further reduce() */
static void ks_mul(const u64 a[4], const u64 b[4], u64 r[4]) {
  u64 t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u64 carry = 0;
    for (int j = 0; j < 4; ++j) {
      u128 product = (u128)a[i] * (u128)b[j];
      u64 lo = (u64)product, hi = (u64)(product >> 64);
      u64 res1 = t[i + j] + lo;
      u64 c1 = res1 < lo;
      u64 res2 = res1 + carry;
      u64 c2 = res2 < carry;
      t[i + j] = res2;
      carry = hi + c1 + c2;
      if (j == 3) t[i + j + 1] = carry;
    }
  }
  r[0] = t[0]; r[1] = t[1]; r[2] = t[2]; r[3] = t[3];
  ks_reduce(r);
  while (ks_ge_n(r)) ks_reduce(r);
}
/* secp256k1.rs:2162-2195 invert: a^(n-2), limbs LS->MS, bits MS->LS; zero -> none */
static int ks_inv(const u64 a[4], u64 r[4]) {
  if ((a[0] | a[1] | a[2] | a[3]) == 0) { r[0] = r[1] = r[2] = r[3] = 0; return 0; }
  static const u64 e[4] = {0xBFD25E8CD036413FULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFFULL,
                           0xFFFFFFFFFFFFFFFEULL};
  u64 result[4] = {1, 0, 0, 0};
  for (int i = 0; i < 4; ++i)
    for (int j = 63; j >= 0; --j) {
      u64 sq[4];
      ks_mul(result, result, sq);               /* square() = s * s (2197-2200) */
      memcpy(result, sq, sizeof sq);
      if ((e[i] >> j) & 1) { ks_mul(result, a, sq); memcpy(result, sq, sizeof sq); }
    }
  memcpy(r, result, sizeof result);
  return 1;
}
/* trait Scalar::from_bytes (2270-2297): big-endian; valid iff < n */
static int ks_from_bytes_be(const unsigned char b[32], u64 l[4]) {
  l[0] = l[1] = l[2] = l[3] = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) l[i] |= (u64)b[31 - (i * 8 + j)] << (j * 8);
  return !ks_ge_n(l);
}

/* forge-ec-signature/src/ecdsa.rs:213-281 Ecdsa::<Secp256k1, D>::verify with the digest given.
 * Returns 1 valid, 0 invalid, 2 where the reference panics (CtOption::unwrap on None: the digest
 * or the affine x, read as a scalar, is >= n). */
int fo_secp256k1_ecdsa_verify(const unsigned char digest[32], const u64 r[4], const u64 s[4],
                              const u64 pk_xy[8], int pk_inf) {
  if ((r[0] | r[1] | r[2] | r[3]) == 0 || (s[0] | s[1] | s[2] | s[3]) == 0) return 0;   /* 215-217 */
  if (ks_ge_n(r) || ks_ge_n(s)) return 0;                   /* 223-228, ct_lt override 2321-2347 */
  u64 h[4];
  if (!ks_from_bytes_be(digest, h)) return 2;               /* 239 unwrap */
  u64 s_inv[4];
  if (!ks_inv(s, s_inv)) return 0;                          /* 243-247 */
  u64 u1[4], u2[4];
  ks_mul(h, s_inv, u1);                                      /* 250 */
  ks_mul(r, s_inv, u2);                                      /* 251 */
  jpt g = k_generator();
  jpt q = k_identity();                                     /* from_affine 1365-1373 */
  if (!pk_inf) {
    for (int i = 0; i < 4; ++i) { q.x.v[i] = pk_xy[i]; q.y.v[i] = pk_xy[4 + i]; }
    q.z = fe_small(1);
  }
  jpt r1 = k_multiply(&g, u1);
  jpt r2 = k_multiply(&q, u2);
  jpt rp = k_padd(&r1, &r2);                                 /* 254-256 */
  if (k_is_identity(&rp)) return 0;                          /* 259-262 */
  fe x, y;
  k_to_affine(&rp, &x, &y);                                  /* 264 */
  /* field_to_bytes -> FieldElement::to_bytes (138-178): mont_reduce(x) = Mul(x, raw 1), big-endian;
   * then Scalar::from_bytes of those bytes: the same limbs, valid iff < n (271 unwrap) */
  fe xr = k_mul(x, fe_small(1));
  if (ks_ge_n(xr.v)) return 2;
  return xr.v[0] == r[0] && xr.v[1] == r[1] && xr.v[2] == r[2] && xr.v[3] == r[3];   /* 274 */
}

/* scalar-field ops for tests: op in {"mul","inv"} */
static void ks_add(const u64 a[4], const u64 b[4], u64 r[4]);
static void ns_add(const u64 a[4], const u64 b[4], u64 r[4]);
int fo_secp256k1_scalar_op(const char* op, const u64 a[4], const u64 b[4], u64 r[4]) {
  if (!strcmp(op, "mul")) { ks_mul(a, b, r); return 0; }
  if (!strcmp(op, "inv")) { return ks_inv(a, r) ? 0 : 1; }
  if (!strcmp(op, "add")) { ks_add(a, b, r); return 0; }
  return -2;
}

/* =====================================================================================
 * P-256  (p256.rs)
 * ===================================================================================== */
static const u64 N_P[4] = {0xFFFFFFFFFFFFFFFFULL, 0x00000000FFFFFFFFULL, 0x0000000000000000ULL,
                           0xFFFFFFFF00000001ULL};                       /* p256.rs:18-19 */

/* p256.rs:70-80 compare */
static int n_cmp(const u64 a[4], const u64 b[4]) {
  for (int i = 3; i >= 0; --i) {
    if (a[i] < b[i]) return -1;
    if (a[i] > b[i]) return 1;
  }
  return 0;
}

/* p256.rs:88-99 reduce: while >= p subtract p */
static void n_reduce(fe* s) {
  while (n_cmp(s->v, N_P) >= 0) {
    u64 borrow = 0;
    for (int i = 0; i < 4; ++i) {
      u64 d1 = s->v[i] - N_P[i];
      u64 b1 = s->v[i] < N_P[i];
      u64 d2 = d1 - borrow;
      u64 b2 = d1 < borrow;
      s->v[i] = d2;
      borrow = b1 + b2;
    }
  }
}

/* p256.rs:416-468 Add */
static fe n_add(fe a, fe b) {
  u64 r[4] = {a.v[0], a.v[1], a.v[2], a.v[3]};
  u64 carry = 0;
  for (int i = 0; i < 4; ++i) {
    u64 s1 = r[i] + b.v[i];
    u64 o1 = s1 < b.v[i];
    u64 s2 = s1 + carry;
    u64 o2 = s2 < carry;
    r[i] = s2;
    carry = o1 + o2;
  }
  while (carry > 0) {                                                   /* :436-452 */
    static const u64 red[4] = {0x0000000000000001ULL, 0xFFFFFFFF00000000ULL, 0xFFFFFFFFFFFFFFFFULL,
                               0x00000000FFFFFFFEULL};
    u64 ac = 0;
    for (int i = 0; i < 4; ++i) {
      u64 s1 = r[i] + red[i];
      u64 o1 = s1 < red[i];
      u64 s2 = s1 + ac;
      u64 o2 = s2 < ac;
      r[i] = s2;
      ac = o1 + o2;
    }
    carry = carry - 1 + ac;
  }
  fe out = {{r[0], r[1], r[2], r[3]}};
  while (n_cmp(out.v, N_P) >= 0) {                                      /* :456-464 */
    u64 borrow = 0;
    for (int i = 0; i < 4; ++i) {
      u64 d1 = out.v[i] - N_P[i];
      u64 b1 = out.v[i] < N_P[i];
      u64 d2 = d1 - borrow;
      u64 b2 = d1 < borrow;
      out.v[i] = d2;
      borrow = b1 + b2;
    }
  }
  return out;
}

/* p256.rs:470-496 Sub: `result += P` is undone by Add's own reduction; then a wrapping
 * 256-bit subtraction => (a-b) mod 2^256 when a < b (possibly >= p). */
static fe n_sub(fe a, fe b) {
  fe r = a;
  if (n_cmp(a.v, b.v) < 0) {
    fe pp = {{N_P[0], N_P[1], N_P[2], N_P[3]}};
    r = n_add(r, pp);
  }
  u64 borrow = 0;
  fe d;
  for (int i = 0; i < 4; ++i) {
    u64 d1 = r.v[i] - b.v[i];
    u64 b1 = r.v[i] < b.v[i];
    u64 d2 = d1 - borrow;
    u64 b2 = d1 < borrow;
    d.v[i] = d2;
    borrow = b1 + b2;
  }
  return d;
}

/* p256.rs:544-704 reduce_wide_p256 */
static fe n_reduce_wide(const u64 wide[8]) {
  u64 c[16];
  for (int i = 0; i < 8; ++i) {
    c[2 * i] = wide[i] & 0xFFFFFFFFULL;
    c[2 * i + 1] = wide[i] >> 32;
  }
  i128 acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  /* s1 */
  for (int i = 0; i < 8; ++i) acc[i] += (i128)c[i];
  /* 2*s2 */
  acc[3] += 2 * (i128)c[11]; acc[4] += 2 * (i128)c[12]; acc[5] += 2 * (i128)c[13];
  acc[6] += 2 * (i128)c[14]; acc[7] += 2 * (i128)c[15];
  /* 2*s3 */
  acc[3] += 2 * (i128)c[12]; acc[4] += 2 * (i128)c[13]; acc[5] += 2 * (i128)c[14];
  acc[6] += 2 * (i128)c[15];
  /* s4 */
  acc[0] += (i128)c[8]; acc[1] += (i128)c[9]; acc[2] += (i128)c[10];
  acc[6] += (i128)c[14]; acc[7] += (i128)c[15];
  /* s5 */
  acc[0] += (i128)c[9]; acc[1] += (i128)c[10]; acc[2] += (i128)c[11]; acc[3] += (i128)c[13];
  acc[4] += (i128)c[14]; acc[5] += (i128)c[15]; acc[6] += (i128)c[13]; acc[7] += (i128)c[8];
  /* -s6 */
  acc[0] -= (i128)c[11]; acc[1] -= (i128)c[12]; acc[2] -= (i128)c[13];
  acc[6] -= (i128)c[8]; acc[7] -= (i128)c[10];
  /* -s7 */
  acc[0] -= (i128)c[12]; acc[1] -= (i128)c[13]; acc[2] -= (i128)c[14]; acc[3] -= (i128)c[15];
  acc[6] -= (i128)c[9]; acc[7] -= (i128)c[11];
  /* -s8 */
  acc[0] -= (i128)c[13]; acc[1] -= (i128)c[14]; acc[2] -= (i128)c[15]; acc[3] -= (i128)c[8];
  acc[4] -= (i128)c[9]; acc[5] -= (i128)c[10]; acc[7] -= (i128)c[12];
  /* -s9 */
  acc[0] -= (i128)c[14]; acc[1] -= (i128)c[15]; acc[3] -= (i128)c[9]; acc[4] -= (i128)c[10];
  acc[5] -= (i128)c[11]; acc[7] -= (i128)c[13];

  for (int i = 0; i < 7; ++i) {                 /* :657-661, arithmetic >> on i128 */
    i128 carry = acc[i] >> 32;
    acc[i] &= (i128)0xFFFFFFFFULL;
    acc[i + 1] += carry;
  }
  i128 carry = acc[7] >> 32;
  acc[7] &= (i128)0xFFFFFFFFULL;

  fe r;
  r.v[0] = (u64)acc[0] | ((u64)acc[1] << 32);
  r.v[1] = (u64)acc[2] | ((u64)acc[3] << 32);
  r.v[2] = (u64)acc[4] | ((u64)acc[5] << 32);
  r.v[3] = (u64)acc[6] | ((u64)acc[7] << 32);

  while (carry > 0) {                           /* :677-686 wrapping subtract p */
    u64 borrow = 0;
    for (int i = 0; i < 4; ++i) {
      u64 d1 = r.v[i] - N_P[i];
      u64 b1 = r.v[i] < N_P[i];
      u64 d2 = d1 - borrow;
      u64 b2 = d1 < borrow;
      r.v[i] = d2;
      borrow = b1 + b2;
    }
    carry -= 1;
  }
  while (carry < 0) {                           /* :689-698 wrapping add p */
    u64 cc = 0;
    for (int i = 0; i < 4; ++i) {
      u64 s1 = r.v[i] + N_P[i];
      u64 o1 = s1 < N_P[i];
      u64 s2 = s1 + cc;
      u64 o2 = s2 < cc;
      r.v[i] = s2;
      cc = o1 + o2;
    }
    carry += 1;
  }
  n_reduce(&r);
  return r;
}

/* p256.rs:498-534 Mul */
static fe n_mul(fe a, fe b) {
  u64 wide[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 carry = 0;
    for (int j = 0; j < 4; ++j) {
      u128 prod = (u128)a.v[i] * (u128)b.v[j] + (u128)wide[i + j] + carry;
      wide[i + j] = (u64)prod;
      carry = prod >> 64;
    }
    u128 sum = (u128)wide[i + 4] + carry;
    wide[i + 4] = (u64)sum;
    if ((sum >> 64) != 0) {
      for (int k = i + 5; k < 8; ++k) {
        u64 nv = wide[k] + 1;
        int ovf = nv == 0;
        wide[k] = nv;
        if (!ovf) break;
      }
    }
  }
  return n_reduce_wide(wide);
}
static fe n_sqr(fe a) { return n_mul(a, a); }   /* p256.rs:772-776 */

/* p256.rs:707-729 Neg */
static fe n_neg(fe a) {
  if (fe_is_zero(&a)) return a;
  fe r;
  u64 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u64 d1 = N_P[i] - a.v[i];
    u64 b1 = N_P[i] < a.v[i];
    u64 d2 = d1 - borrow;
    u64 b2 = d1 < borrow;
    r.v[i] = d2;
    borrow = b1 + b2;
  }
  return r;
}

/* p256.rs:376-393 pow (LSB first), 343-370 invert (exponent p-2 by limb-wise borrow) */
static fe n_pow(fe a, const u64 e[4]) {
  fe result = fe_small(1);
  fe base = a;
  for (int w = 0; w < 4; ++w) {
    u64 x = e[w];
    for (int i = 0; i < 64; ++i) {
      if (x & 1) result = n_mul(result, base);
      base = n_sqr(base);
      x >>= 1;
    }
  }
  return result;
}
static fe n_inv(fe a) {
  if (fe_is_zero(&a)) return fe_small(0);
  u64 e[4] = {N_P[0], N_P[1], N_P[2], N_P[3]};
  u64 borrow = 2;
  for (int i = 0; i < 4; ++i) {
    u64 val = e[i] - borrow;
    int did = e[i] < borrow;
    e[i] = val;
    if (!did) { borrow = 0; break; }
    borrow = 1;
  }
  return n_pow(a, e);
}

static jpt n_identity(void) {                   /* :1827-1829 */
  jpt p = {fe_small(0), fe_small(1), fe_small(0)};
  return p;
}
static int n_is_identity(const jpt* p) { return fe_is_zero(&p->z); }   /* :1831-1833 */

/* p256.rs:1869-1912 double (a = 0 formula although a = -3; Z==1 shortcut) */
static jpt n_double(const jpt* p) {
  if (n_is_identity(p)) return n_identity();
  fe xx = n_sqr(p->x);
  fe yy = n_sqr(p->y);
  fe yyyy = n_sqr(yy);
  fe xy2 = n_sqr(n_add(p->x, yy));
  fe w = n_sub(n_sub(xy2, xx), yyyy);
  fe d = n_add(w, w);
  fe e = n_mul(fe_small(3), xx);
  fe ee = n_sqr(e);
  fe x3 = n_sub(n_sub(ee, d), d);
  fe eight_yyyy = n_mul(fe_small(8), yyyy);
  fe y3 = n_sub(n_mul(e, n_sub(d, x3)), eight_yyyy);
  fe z3 = n_add(p->y, p->y);
  fe one = fe_small(1);
  if (!fe_eq(&p->z, &one)) z3 = n_mul(z3, p->z);
  jpt r = {x3, y3, z3};
  return r;
}

/* p256.rs:2034-2068 ConstantTimeEq for ProjectivePoint */
static int n_pt_eq(const jpt* p, const jpt* q) {
  if (n_is_identity(p) && n_is_identity(q)) return 1;
  if (n_is_identity(p) || n_is_identity(q)) return 0;
  fe z1z1 = n_sqr(p->z);
  fe z2z2 = n_sqr(q->z);
  fe u1 = n_mul(p->x, z2z2);
  fe u2 = n_mul(q->x, z1z1);
  fe s1 = n_mul(n_mul(p->y, q->z), z2z2);
  fe s2 = n_mul(n_mul(q->y, p->z), z1z1);
  return fe_eq(&u1, &u2) & fe_eq(&s1, &s2);
}

/* p256.rs:1938-2007 Add */
static jpt n_padd(const jpt* p, const jpt* q) {
  if (n_is_identity(p)) return *q;
  if (n_is_identity(q)) return *p;
  if (n_pt_eq(p, q)) return n_double(p);
  fe z1z1 = n_sqr(p->z);
  fe z2z2 = n_sqr(q->z);
  fe u1 = n_mul(p->x, z2z2);
  fe u2 = n_mul(q->x, z1z1);
  fe s1 = n_mul(n_mul(p->y, q->z), z2z2);
  fe s2 = n_mul(n_mul(q->y, p->z), z1z1);
  fe ns2 = n_neg(s2);
  if (fe_eq(&u1, &u2) && fe_eq(&s1, &ns2)) return n_identity();        /* :1977 */
  fe h = n_sub(u2, u1);
  fe i = n_sqr(n_add(h, h));
  fe j = n_mul(h, i);
  fe r = n_add(n_sub(s2, s1), n_sub(s2, s1));
  fe v = n_mul(u1, i);
  fe x3 = n_sub(n_sub(n_sub(n_sqr(r), j), v), v);
  fe y3 = n_sub(n_mul(r, n_sub(v, x3)), n_mul(n_add(s1, s1), j));
  fe z3 = n_mul(n_sub(n_sub(n_sqr(n_add(p->z, q->z)), z1z1), z2z2), h);
  jpt o = {x3, y3, z3};
  return o;
}

static jpt n_generator(void) {                  /* :2092-2110 */
  jpt g = {{{0xF4A13945D898C296ULL, 0x77037D812DEB33A0ULL, 0xF8BCE6E563A440F2ULL, 0x6B17D1F2E12C4247ULL}},
           {{0xCBB6406837BF51F5ULL, 0x2BCE33576B315ECEULL, 0x8EE7EB4A7C0F9E16ULL, 0x4FE342E2FE1A7F9BULL}},
           {{1, 0, 0, 0}}};
  return g;
}

static int n_to_affine(const jpt* p, fe* x, fe* y) {                    /* :1835-1857 */
  if (n_is_identity(p)) { *x = fe_small(0); *y = fe_small(0); return 1; }
  fe zi = n_inv(p->z);
  fe zi2 = n_sqr(zi);
  fe zi3 = n_mul(zi2, zi);
  *x = n_mul(p->x, zi2);
  *y = n_mul(p->y, zi3);
  return 0;
}

/* p256.rs:2120-2156 Curve::multiply: MSB-first double-and-add over the inherent big-endian
 * Scalar::to_bytes (1026-1038); the add is data-dependent. */
static jpt n_multiply(const jpt* point, const u64 k[4]) {
  if (n_is_identity(point) || (k[0] | k[1] | k[2] | k[3]) == 0) return n_identity();
  unsigned char bytes[32];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) bytes[31 - (i * 8 + j)] = (unsigned char)((k[i] >> (j * 8)) & 0xFF);
  jpt result = n_identity();
  for (int i = 0; i < 256; ++i) {
    int bit = (bytes[i / 8] >> (7 - (i % 8))) & 1;
    result = n_double(&result);
    if (bit == 1) result = n_padd(&result, point);
  }
  return result;
}

/* ---- P-256 scalar field (p256.rs:875-1038, 1409-1432) and Ecdsa::<P256, D>::verify -------------
 * The scalar Mul is an exact schoolbook product followed by reduce_wide (924-1020), which is NOT a
 * reduction modulo n: the second folding round adds only the low four limbs of high2 * (2^256 - n)
 * (993-998) and drops the rest.  Restated limb for limb. */
static const u64 NS_N[4] = {0xF3B9CAC2FC632551ULL, 0xBCE6FAADA7179E84ULL, 0xFFFFFFFFFFFFFFFFULL,
                            0xFFFFFFFF00000000ULL};                                  /* p256.rs:23-24 */
static const u64 NS_C[4] = {0x0C46353D039CDAAFULL, 0x4319055258E8617BULL, 0x0000000000000000ULL,
                            0x00000000FFFFFFFFULL};                                  /* TWO_256_MINUS_N 932-937 */
static void ns_sub_n_while_ge(u64 v[4]) {                                          /* 1007-1017 / 911-920 */
  while (n_cmp(v, NS_N) >= 0) {
    u64 borrow = 0;
    for (int i = 0; i < 4; ++i) {
      u64 d1 = v[i] - NS_N[i];
      u64 b1 = v[i] < NS_N[i];
      u64 d2 = d1 - borrow;
      u64 b2 = d1 < borrow;
      v[i] = d2;
      borrow = b1 + b2;
    }
  }
}
static void ns_times_c(const u64 h[4], u128 prod[8]) {                             /* 943-954 / 979-990 */
  for (int i = 0; i < 8; ++i) prod[i] = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) prod[i + j] += (u128)h[i] * (u128)NS_C[j];
  for (int i = 0; i < 7; ++i) {
    prod[i + 1] += prod[i] >> 64;
    prod[i] &= (u128)0xFFFFFFFFFFFFFFFFULL;
  }
}
static void ns_reduce_wide(const u64 wide[8], u64 out[4]) {                        /* 924-1020 */
  u128 product[8], result[8];
  ns_times_c(wide + 4, product);
  for (int i = 0; i < 4; ++i) result[i] = product[i] + (u128)wide[i];
  for (int i = 4; i < 8; ++i) result[i] = product[i];
  for (int i = 0; i < 7; ++i) {
    result[i + 1] += result[i] >> 64;
    result[i] &= (u128)0xFFFFFFFFFFFFFFFFULL;
  }
  u64 low2[4] = {(u64)result[0], (u64)result[1], (u64)result[2], (u64)result[3]};
  u64 high2[4] = {(u64)result[4], (u64)result[5], (u64)result[6], (u64)result[7]};
  if (high2[0] | high2[1] | high2[2] | high2[3]) {
    u128 product2[8];
    ns_times_c(high2, product2);
    u128 carry = 0;
    for (int i = 0; i < 4; ++i) {                  /* only product2[0..4] is used (993-998) */
      u128 sum = (u128)low2[i] + product2[i] + carry;
      low2[i] = (u64)sum;
      carry = sum >> 64;
    }
    if (carry > 0) {                               /* 1000-1007 */
      u128 c = carry;
      for (int i = 0; i < 4; ++i) {
        u128 sum = (u128)low2[i] + c * (u128)NS_C[i];
        low2[i] = (u64)sum;
        c = sum >> 64;
      }
    }
  }
  ns_sub_n_while_ge(low2);
  memcpy(out, low2, 32);
}
static void ns_mul(const u64 a[4], const u64 b[4], u64 r[4]) {                     /* 1409-1432 */
  u64 res[8] = {0};
  for (int i = 0; i < 4; ++i) {
    u64 carry = 0;
    for (int j = 0; j < 4; ++j) {
      u128 p = (u128)a[i] * (u128)b[j] + (u128)res[i + j] + (u128)carry;
      res[i + j] = (u64)p;
      carry = (u64)(p >> 64);
    }
    res[i + 4] = carry;
  }
  ns_reduce_wide(res, r);
}
static int ns_inv(const u64 a[4], u64 r[4]) {                                      /* 1057-1080, pow 1083-1100 */
  if ((a[0] | a[1] | a[2] | a[3]) == 0) { memset(r, 0, 32); return 0; }
  static const u64 e[4] = {0xF3B9CAC2FC63254FULL, 0xBCE6FAADA7179E84ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFF00000000ULL};
  u64 result[4] = {1, 0, 0, 0}, base[4], t[4];
  memcpy(base, a, 32);
  for (int i = 0; i < 4; ++i) {
    u64 x = e[i];
    for (int k = 0; k < 64; ++k) {
      if (x & 1) { ns_mul(result, base, t); memcpy(result, t, 32); }
      ns_mul(base, base, t);                       /* square() = s * s (1103-1106) */
      memcpy(base, t, 32);
      x >>= 1;
    }
  }
  memcpy(r, result, 32);
  return 1;
}
static int ns_from_bytes_be(const unsigned char b[32], u64 l[4]) {                  /* 1041-1055 */
  l[0] = l[1] = l[2] = l[3] = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) l[i] |= (u64)b[31 - (i * 8 + j)] << (j * 8);
  return n_cmp(l, NS_N) < 0;
}
/* Scalar::ct_lt, the trait default (forge-ec-core/src/lib.rs:497-531; P-256 does not override it): on
 * big-endian bytes, with `is_lt = !borrow(other - self)`, i.e. self_byte <= other_byte -- so the verdict is
 * decided by the first byte pair alone: true iff self's top byte <= other's (equal bytes set it, a greater
 * byte clears the equality chain).  Restated as written. */
static int ns_ct_lt(const u64 a[4], const u64 b[4]) {
  int result = 0, eq_so_far = 1;
  for (int i = 0; i < 32; ++i) {
    unsigned char sb = (unsigned char)(a[3 - i / 8] >> (56 - 8 * (i % 8)));
    unsigned char ob = (unsigned char)(b[3 - i / 8] >> (56 - 8 * (i % 8)));
    int is_lt = !(ob < sb);
    int is_eq = sb == ob;
    result |= eq_so_far & is_lt;
    eq_so_far &= is_eq;
  }
  return result;
}
/* forge-ec-signature/src/ecdsa.rs:213-281 for C = P256 with the digest given: 1 valid, 0 invalid,
 * 2 where the reference panics (CtOption::unwrap on None) */
int fo_p256_ecdsa_verify(const unsigned char digest[32], const u64 r[4], const u64 s[4], const u64 pk_xy[8],
                         int pk_inf) {
  if ((r[0] | r[1] | r[2] | r[3]) == 0 || (s[0] | s[1] | s[2] | s[3]) == 0) return 0;
  if (!(ns_ct_lt(r, NS_N) & ns_ct_lt(s, NS_N))) return 0;
  u64 h[4];
  if (!ns_from_bytes_be(digest, h)) return 2;               /* 239 unwrap */
  u64 s_inv[4];
  if (!ns_inv(s, s_inv)) return 0;
  u64 u1[4], u2[4];
  ns_mul(h, s_inv, u1);
  ns_mul(r, s_inv, u2);
  jpt g = n_generator();
  jpt q = n_identity();                                     /* from_affine 1859-1867 */
  if (!pk_inf) {
    for (int i = 0; i < 4; ++i) { q.x.v[i] = pk_xy[i]; q.y.v[i] = pk_xy[4 + i]; }
    q.z = fe_small(1);
  }
  jpt r1 = n_multiply(&g, u1);
  jpt r2 = n_multiply(&q, u2);
  jpt rp = n_padd(&r1, &r2);
  if (n_is_identity(&rp)) return 0;
  fe x, y;
  n_to_affine(&rp, &x, &y);
  /* field_to_bytes = FieldElement::to_bytes (288-300): the raw limbs big-endian; Scalar::from_bytes of
   * them: the same limbs, valid iff < n (271 unwrap) */
  if (n_cmp(x.v, NS_N) >= 0) return 2;
  return x.v[0] == r[0] && x.v[1] == r[1] && x.v[2] == r[2] && x.v[3] == r[3];
}
int fo_p256_scalar_op(const char* op, const u64 a[4], const u64 b[4], u64 r[4]) {
  if (!strcmp(op, "mul")) { ns_mul(a, b, r); return 0; }
  if (!strcmp(op, "inv")) { return ns_inv(a, r) ? 0 : 1; }
  if (!strcmp(op, "add")) { ns_add(a, b, r); return 0; }
  return -2;
}

/* =====================================================================================
 * Ed25519  (ed25519.rs)
 * ===================================================================================== */
static const u64 E_P[4] = {0xFFFFFFFFFFFFFFEDULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL,
                           0x7FFFFFFFFFFFFFFFULL};                       /* ed25519.rs:64-69 */
static const u64 E_D[4] = {0x75EB4DCA135EDEFFULL, 0x00E0149A8283B156ULL, 0x198E80F2EEF3D130ULL,
                           0x2406875CC61A8E3CULL};                       /* ed25519.rs:86-91 */

/* ed25519.rs:214-247 reduce */
static void e_reduce(fe* s) {
  u64 top = s->v[3] >> 63;
  s->v[3] &= 0x7FFFFFFFFFFFFFFFULL;
  u64 carry = top * 19;
  for (int i = 0; i < 4; ++i) {
    u128 sum = (u128)s->v[i] + (u128)carry;
    s->v[i] = (u64)sum;
    carry = (u64)(sum >> 64);
  }
  u64 diff[4];
  u64 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u64 d1 = s->v[i] - E_P[i];
    u64 b1 = s->v[i] < E_P[i];
    u64 d2 = d1 - borrow;
    u64 b2 = d1 < borrow;
    diff[i] = d2;
    borrow = b1 + b2;
  }
  if (borrow == 0)
    for (int i = 0; i < 4; ++i) s->v[i] = diff[i];
}

/* ed25519.rs:260-289 reduce_wide: low + 38*high; the final carry is folded with x19 (not 38) */
static fe e_reduce_wide(const u64 l[8]) {
  u64 low[4] = {l[0], l[1], l[2], l[3]};
  u128 carry = 0;
  for (int i = 0; i < 4; ++i) {
    u128 prod = (u128)l[4 + i] * (u128)38 + (u128)low[i] + carry;
    low[i] = (u64)prod;
    carry = prod >> 64;
  }
  u64 fc = (u64)carry * 19;
  for (int i = 0; i < 4; ++i) {
    u128 sum = (u128)low[i] + (u128)fc;
    low[i] = (u64)sum;
    fc = (u64)(sum >> 64);
  }
  fe r = {{low[0], low[1], low[2], low[3]}};
  e_reduce(&r);
  return r;
}

/* ed25519.rs:458-488 Add */
static fe e_add(fe a, fe b) {
  u64 r[4];
  u64 carry = 0;
  for (int i = 0; i < 4; ++i) {
    u128 sum = (u128)a.v[i] + (u128)b.v[i] + (u128)carry;
    r[i] = (u64)sum;
    carry = (u64)(sum >> 64);
  }
  if (carry > 0) {
    u64 ec = carry * 19;
    for (int i = 0; i < 4; ++i) {
      u128 sum = (u128)r[i] + (u128)ec;
      r[i] = (u64)sum;
      ec = (u64)(sum >> 64);
    }
  }
  fe o = {{r[0], r[1], r[2], r[3]}};
  e_reduce(&o);
  return o;
}

/* ed25519.rs:490-520 Sub (no final reduce) */
static fe e_sub(fe a, fe b) {
  u64 r[4];
  u64 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u64 d1 = a.v[i] - b.v[i];
    u64 b1 = a.v[i] < b.v[i];
    u64 d2 = d1 - borrow;
    u64 b2 = d1 < borrow;
    r[i] = d2;
    borrow = b1 + b2;
  }
  if (borrow > 0) {
    u64 carry = 0;
    for (int i = 0; i < 4; ++i) {
      u128 sum = (u128)r[i] + (u128)E_P[i] + (u128)carry;
      r[i] = (u64)sum;
      carry = (u64)(sum >> 64);
    }
  }
  fe o = {{r[0], r[1], r[2], r[3]}};
  return o;
}

/* ed25519.rs:522-545 Mul */
static fe e_mul(fe a, fe b) {
  u64 product[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u64 carry = 0;
    for (int j = 0; j < 4; ++j) {
      u128 prod = (u128)a.v[i] * (u128)b.v[j] + (u128)product[i + j] + (u128)carry;
      product[i + j] = (u64)prod;
      carry = (u64)(prod >> 64);
    }
    product[i + 4] = carry;
  }
  return e_reduce_wide(product);
}
static fe e_sqr(fe a) { return e_mul(a, a); }   /* ed25519.rs:623-625 */

/* ed25519.rs:547-570 Neg */
static fe e_neg(fe a) {
  fe r;
  u64 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u64 d1 = E_P[i] - a.v[i];
    u64 b1 = E_P[i] < a.v[i];
    u64 d2 = d1 - borrow;
    u64 b2 = d1 < borrow;
    r.v[i] = d2;
    borrow = b1 + b2;
  }
  if (fe_is_zero(&a)) return fe_small(0);
  return r;
}

/* ed25519.rs:410-431 pow (mul every bit, select), 603-621 invert */
static fe e_pow(fe a, const u64 e[4]) {
  fe result = fe_small(1);
  fe base = a;
  for (int w = 0; w < 4; ++w) {
    for (int i = 0; i < 64; ++i) {
      int bit = (int)((e[w] >> i) & 1);
      fe nr = e_mul(result, base);
      result = fe_select(&result, &nr, bit);
      base = e_sqr(base);
    }
  }
  return result;
}
static fe e_inv(fe a) {
  static const u64 e[4] = {0xFFFFFFFFFFFFFFEBULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL,
                           0x7FFFFFFFFFFFFFFFULL};
  return e_pow(a, e);                          /* CtOption is none for zero; value is pow(0) */
}

static ept e_identity(void) {                   /* :1776-1783 */
  ept p = {fe_small(0), fe_small(1), fe_small(1), fe_small(0)};
  return p;
}
static int e_is_identity(const ept* p) {        /* :1785-1791 */
  return fe_is_zero(&p->x) & fe_eq(&p->y, &p->z) & fe_is_zero(&p->t);
}

/* ed25519.rs:1864-1928 Add (C = T1*T2*d with d not 2d; D = Z1*Z2 not 2*Z1*Z2) */
static ept e_padd(const ept* p, const ept* q) {
  if (e_is_identity(p)) return *q;
  if (e_is_identity(q)) return *p;
  fe nqx = e_neg(q->x);
  if (fe_eq(&p->x, &nqx) && fe_eq(&p->y, &q->y)) return e_identity();  /* :1878 raw coords */
  fe d = {{E_D[0], E_D[1], E_D[2], E_D[3]}};
  fe a = e_mul(e_sub(p->y, p->x), e_sub(q->y, q->x));
  fe b = e_mul(e_add(p->y, p->x), e_add(q->y, q->x));
  fe c = e_mul(e_mul(p->t, q->t), d);
  fe dd = e_mul(p->z, q->z);
  fe e = e_sub(b, a);
  fe f = e_sub(dd, c);
  fe g = e_add(dd, c);
  fe h = e_add(b, a);
  ept o = {e_mul(e, f), e_mul(g, h), e_mul(f, g), e_mul(e, h)};         /* x, y, z, t */
  return o;
}
static ept e_double(const ept* p) { return e_padd(p, p); }              /* :1828-1832 */

/* ed25519.rs:2015-2052 generator: hard-coded x,y; t = x*y (from_affine 1813-1826) */
static ept e_generator(void) {
  fe y = {{0x2DFC9311D90045F9ULL, 0x0A71C760BF38C6A7ULL, 0xA6FB8EEBCEAA2C8DULL, 0x5FD9C9E6CC3CCCCCULL}};
  fe x = {{0x1A1462FAFB9683F2ULL, 0xD2E8A68B8B30C404ULL, 0xA0C0F3A1E9E71B63ULL, 0x216936D3CD6E53FEULL}};
  ept g = {x, y, fe_small(1), e_mul(x, y)};
  return g;
}

static int e_to_affine(const ept* p, fe* x, fe* y) {                    /* :1793-1811 */
  if (e_is_identity(p)) { *x = fe_small(0); *y = fe_small(0); return 1; }
  fe zi = e_inv(p->z);
  *x = e_mul(p->x, zi);
  *y = e_mul(p->y, zi);
  return 0;
}

/* ed25519.rs:2062-2097 Curve::multiply: LSB-first over scalar.to_raw(), add every bit + select */
static ept e_multiply(const ept* point, const u64 k[4]) {
  if (e_is_identity(point) || (k[0] | k[1] | k[2] | k[3]) == 0) return e_identity();
  ept result = e_identity();
  ept addend = *point;
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 64; ++j) {
      int bit = (k[i] & ((u64)1 << j)) != 0;
      ept rpa = e_padd(&result, &addend);
      result = bit ? rpa : result;
      addend = e_double(&addend);
    }
  }
  return result;
}

/* =====================================================================================
 * C entry points
 * ===================================================================================== */
int fo_point_limbs(int curve) {
  return (curve == FO_SECP256K1 || curve == FO_P256) ? 12 : (curve == FO_ED25519 ? 16 : 0);
}

static fe ld(const u64* p) { fe r = {{p[0], p[1], p[2], p[3]}}; return r; }
static void st(u64* p, fe a) { p[0] = a.v[0]; p[1] = a.v[1]; p[2] = a.v[2]; p[3] = a.v[3]; }
static jpt ldj(const u64* p) { jpt r = {ld(p), ld(p + 4), ld(p + 8)}; return r; }
static void stj(u64* p, jpt a) { st(p, a.x); st(p + 4, a.y); st(p + 8, a.z); }
static ept lde(const u64* p) { ept r = {ld(p), ld(p + 4), ld(p + 8), ld(p + 12)}; return r; }
static void ste(u64* p, ept a) { st(p, a.x); st(p + 4, a.y); st(p + 8, a.z); st(p + 12, a.t); }

int fo_field_op(int curve, const char* op, const u64 a[4], const u64 b[4], u64 r[4]) {
  fe x = ld(a), y = b ? ld(b) : fe_small(0), o;
  int c = curve;
  if (c != FO_SECP256K1 && c != FO_P256 && c != FO_ED25519) return -1;
  if (!strcmp(op, "add")) o = c == 0 ? k_add(x, y) : c == 1 ? n_add(x, y) : e_add(x, y);
  else if (!strcmp(op, "sub")) o = c == 0 ? k_sub(x, y) : c == 1 ? n_sub(x, y) : e_sub(x, y);
  else if (!strcmp(op, "mul")) o = c == 0 ? k_mul(x, y) : c == 1 ? n_mul(x, y) : e_mul(x, y);
  else if (!strcmp(op, "sqr")) o = c == 0 ? k_sqr(x) : c == 1 ? n_sqr(x) : e_sqr(x);
  else if (!strcmp(op, "neg")) o = c == 0 ? k_neg(x) : c == 1 ? n_neg(x) : e_neg(x);
  else if (!strcmp(op, "inv")) o = c == 0 ? k_inv(x) : c == 1 ? n_inv(x) : e_inv(x);
  else return -2;
  st(r, o);
  return 0;
}

void fo_identity(int curve, u64* p) {
  if (curve == FO_SECP256K1) stj(p, k_identity());
  else if (curve == FO_P256) stj(p, n_identity());
  else ste(p, e_identity());
}
void fo_generator(int curve, u64* p) {
  if (curve == FO_SECP256K1) stj(p, k_generator());
  else if (curve == FO_P256) stj(p, n_generator());
  else ste(p, e_generator());
}
int fo_is_identity(int curve, const u64* p) {
  if (curve == FO_SECP256K1) { jpt a = ldj(p); return k_is_identity(&a); }
  if (curve == FO_P256) { jpt a = ldj(p); return n_is_identity(&a); }
  ept a = lde(p);
  return e_is_identity(&a);
}
void fo_point_add(int curve, const u64* p, const u64* q, u64* r) {
  if (curve == FO_SECP256K1) { jpt a = ldj(p), b = ldj(q); stj(r, k_padd(&a, &b)); }
  else if (curve == FO_P256) { jpt a = ldj(p), b = ldj(q); stj(r, n_padd(&a, &b)); }
  else { ept a = lde(p), b = lde(q); ste(r, e_padd(&a, &b)); }
}
void fo_point_double(int curve, const u64* p, u64* r) {
  if (curve == FO_SECP256K1) { jpt a = ldj(p); stj(r, k_double(&a)); }
  else if (curve == FO_P256) { jpt a = ldj(p); stj(r, n_double(&a)); }
  else { ept a = lde(p); ste(r, e_double(&a)); }
}
void fo_secp256k1_point_double_trait(const u64* p, u64* r) {
  jpt a = ldj(p);
  stj(r, k_double_trait(&a));
}
void fo_point_negate(int curve, const u64* p, u64* r) {
  if (curve == FO_SECP256K1) { jpt a = ldj(p); a.y = k_neg(a.y); stj(r, a); }        /* :1420-1422 */
  else if (curve == FO_P256) { jpt a = ldj(p); a.y = n_neg(a.y); stj(r, a); }        /* p256 :1914-1916 */
  else { ept a = lde(p); a.x = e_neg(a.x); a.t = e_neg(a.t); ste(r, a); }            /* ed :1834-1841 */
}
int fo_to_affine(int curve, const u64* p, u64* xy) {
  fe x, y;
  int inf;
  if (curve == FO_SECP256K1) { jpt a = ldj(p); inf = k_to_affine(&a, &x, &y); }
  else if (curve == FO_P256) { jpt a = ldj(p); inf = n_to_affine(&a, &x, &y); }
  else { ept a = lde(p); inf = e_to_affine(&a, &x, &y); }
  st(xy, x);
  st(xy + 4, y);
  return inf;
}
void fo_multiply(int curve, const u64* point, const u64 scalar[4], u64* out) {
  if (curve == FO_SECP256K1) { jpt a = ldj(point); stj(out, k_multiply(&a, scalar)); }
  else if (curve == FO_P256) { jpt a = ldj(point); stj(out, n_multiply(&a, scalar)); }
  else { ept a = lde(point); ste(out, e_multiply(&a, scalar)); }
}

/* ---- threaded batch drivers ---- */
typedef struct {
  int kind;               /* 0 mul, 1 fixed, 2 double-mul, 3 to_affine */
  int curve;
  const u64 *s, *s2, *pts;
  u64* out;
  uint8_t* inf;
  size_t lo, hi;
} job_t;

static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  int pl = fo_point_limbs(j->curve);
  u64 g[16], a[16], b[16];
  if (j->kind == 2) fo_generator(j->curve, g);
  for (size_t i = j->lo; i < j->hi; ++i) {
    switch (j->kind) {
      case 0: fo_multiply(j->curve, j->pts + i * pl, j->s + i * 4, j->out + i * pl); break;
      case 1: fo_multiply(j->curve, j->pts, j->s + i * 4, j->out + i * pl); break;
      case 2:
        fo_multiply(j->curve, g, j->s + i * 4, a);
        fo_multiply(j->curve, j->pts + i * pl, j->s2 + i * 4, b);
        fo_point_add(j->curve, a, b, j->out + i * pl);
        break;
      default: j->inf[i] = (uint8_t)fo_to_affine(j->curve, j->pts + i * pl, j->out + i * 8); break;
    }
  }
  return NULL;
}

static void run_jobs(job_t proto, size_t n, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
  pthread_t th[256];
  job_t jobs[256];
  for (int t = 0; t < nthreads; ++t) {
    jobs[t] = proto;
    jobs[t].lo = n * (size_t)t / (size_t)nthreads;
    jobs[t].hi = n * (size_t)(t + 1) / (size_t)nthreads;
  }
  if (nthreads == 1) { worker(&jobs[0]); return; }
  for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, worker, &jobs[t]);
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}

void fo_batch_mul(int curve, const u64* scalars, const u64* points, u64* out, size_t n, int nthreads) {
  job_t j = {0, curve, scalars, NULL, points, out, NULL, 0, 0};
  run_jobs(j, n, nthreads);
}
void fo_batch_mul_fixed(int curve, const u64* scalars, const u64* base, u64* out, size_t n, int nthreads) {
  job_t j = {1, curve, scalars, NULL, base, out, NULL, 0, 0};
  run_jobs(j, n, nthreads);
}
void fo_batch_double_mul(int curve, const u64* u1, const u64* u2, const u64* q, u64* out, size_t n,
                         int nthreads) {
  job_t j = {2, curve, u1, u2, q, out, NULL, 0, 0};
  run_jobs(j, n, nthreads);
}
typedef struct { const unsigned char* dg; const u64 *r, *s, *pk; const uint8_t* inf; uint8_t* out; size_t lo, hi; } vjob_t;
static void* vworker(void* arg) {
  vjob_t* j = (vjob_t*)arg;
  for (size_t i = j->lo; i < j->hi; ++i)
    j->out[i] = (uint8_t)fo_secp256k1_ecdsa_verify(j->dg + 32 * i, j->r + 4 * i, j->s + 4 * i, j->pk + 8 * i,
                                                   j->inf ? j->inf[i] : 0);
  return NULL;
}
/* ---- Eddsa::<Ed25519, D>::verify / Ed25519::verify from the point computation on
 * (forge-ec-signature/src/eddsa.rs:174-211 and 430-447; the hashing and the message special cases at
 * 157-170 / 361-374 stay with the caller): k = from_bytes_reduced(hash) and s are given as scalars.
 * 1 = true, 0 = false, 2 = the reference panics (to_affine's z.invert().unwrap() on z == 0 of a point
 * that is not the identity, ed25519.rs:1805 with invert 603-621). */
static ept e_from_affine(const u64 xy[8], int inf) {                    /* ed25519.rs:1813-1826 */
  if (inf) return e_identity();
  fe x = ld(xy), y = ld(xy + 4);
  ept p = {x, y, fe_small(1), e_mul(x, y)};
  return p;
}
static ept e_negate(const ept* p) {                                      /* 1834-1841 */
  ept r = {e_neg(p->x), p->y, p->z, e_neg(p->t)};
  return r;
}
int fo_ed25519_eddsa_verify(const u64 r_xy[8], int r_inf, const u64 pk_xy[8], int pk_inf, const u64 s[4],
                            const u64 k[4]) {
  if (r_inf) return 0;                                                   /* eddsa.rs:174-177 */
  ept g = e_generator();
  ept s_g = e_multiply(&g, s);                                           /* 196 / 431 */
  ept a = e_from_affine(pk_xy, pk_inf);
  ept k_a = e_multiply(&a, k);                                           /* 199 / 434 */
  ept r = e_from_affine(r_xy, 0);
  ept rk = e_padd(&r, &k_a);                                             /* 200 / 435 */
  if ((!e_is_identity(&s_g) && fe_is_zero(&s_g.z)) || (!e_is_identity(&rk) && fe_is_zero(&rk.z))) return 2;
  fe x1, y1, x2, y2;
  int i1 = e_to_affine(&s_g, &x1, &y1), i2 = e_to_affine(&rk, &x2, &y2); /* 204-205 / 439-440 */
  u64 a1[8], a2[8];
  st(a1, x1); st(a1 + 4, y1); st(a2, x2); st(a2 + 4, y2);
  ept p1 = e_from_affine(a1, i1), p2 = e_from_affine(a2, i2);
  ept n2 = e_negate(&p2);                                                /* Sub = self + rhs.negate(), 1936-1947 */
  ept diff = e_padd(&p1, &n2);                                           /* 210 / 446 */
  return e_is_identity(&diff);
}
/* ---- Ecdsa::<C, D>::batch_verify (forge-ec-signature/src/ecdsa.rs:287-391) for C = Secp256k1 (curve 0) and
 * P256 (curve 1), digests and the weights a_i (302-306: Scalar::random of the reference's OsRng) supplied.
 * 1 true, 0 false, 2 = the reference panics (unwrap at 334 or 381).  The loop returns at the FIRST signature
 * that fails a check; r_sum and the scalar sum are folded strictly in index order.  detail (may be NULL):
 * [0..12) r_sum (Jacobian limbs), [12..16) r_scalar_sum -- zero when the loop returned early. */
static void ks_add(const u64 a[4], const u64 b[4], u64 r[4]) {            /* secp256k1.rs:2358-2378 */
  u64 carry = 0;
  for (int i = 0; i < 4; ++i) {
    u64 s1 = a[i] + b[i];
    int c1 = s1 < a[i];
    u64 s2 = s1 + carry;
    int c2 = s2 < s1;
    r[i] = s2;
    carry = (c1 || c2) ? 1 : 0;
  }
  ks_reduce(r);                                                            /* the carry out is dropped */
}
static void ns_add(const u64 a[4], const u64 b[4], u64 r[4]) {            /* p256.rs:1352-1375 */
  u64 carry = 0;
  for (int i = 0; i < 4; ++i) {
    u64 s1 = a[i] + b[i];
    u64 o1 = s1 < a[i];
    u64 s2 = s1 + carry;
    u64 o2 = s2 < s1;
    r[i] = s2;
    carry = o1 + o2;
  }
  if (carry > 0 || n_cmp(r, NS_N) >= 0) ns_sub_n_while_ge(r);             /* reduce() sees only the low 256 bits */
}
int fo_ecdsa_batch_verify(int curve, const unsigned char* digests, const u64* r, const u64* s, const u64* pk_xy,
                          const uint8_t* pk_inf, const u64* a, size_t n, u64* detail) {
  if (detail) memset(detail, 0, 16 * sizeof(u64));
  if (curve != 0 && curve != 1) return -1;
  if (n == 0) return 0;                                                    /* 289-291 */
  jpt r_sum = curve == 0 ? k_identity() : n_identity();                    /* 310 */
  for (size_t i = 0; i < n; ++i) {
    const u64 *ri = r + 4 * i, *si = s + 4 * i, *ai = a + 4 * i;
    if ((ri[0] | ri[1] | ri[2] | ri[3]) == 0 || (si[0] | si[1] | si[2] | si[3]) == 0) return 0;   /* 317-319 */
    u64 h[4], s_inv[4], u1[4], u2[4], au1[4], au2[4];
    jpt q, r1, r2, ri_pt;
    if (curve == 0) {
      if (ks_ge_n(ri) || ks_ge_n(si)) return 0;                            /* 322-327 */
      if (!ks_from_bytes_be(digests + 32 * i, h)) return 2;                /* 334 */
      if (!ks_inv(si, s_inv)) return 0;                                    /* 338-342 */
      ks_mul(h, s_inv, u1); ks_mul(ri, s_inv, u2);                         /* 345-346 */
      ks_mul(ai, u1, au1); ks_mul(ai, u2, au2);                            /* 349-350 */
      jpt g = k_generator();
      q = k_identity();
      if (!(pk_inf && pk_inf[i])) { q.x = ld(pk_xy + 8 * i); q.y = ld(pk_xy + 8 * i + 4); q.z = fe_small(1); }
      r1 = k_multiply(&g, au1); r2 = k_multiply(&q, au2);                  /* 353-354 */
      ri_pt = k_padd(&r1, &r2);                                            /* 355 */
      r_sum = k_padd(&r_sum, &ri_pt);                                      /* 358 */
    } else {
      if (!(ns_ct_lt(ri, NS_N) & ns_ct_lt(si, NS_N))) return 0;
      if (!ns_from_bytes_be(digests + 32 * i, h)) return 2;
      if (!ns_inv(si, s_inv)) return 0;
      ns_mul(h, s_inv, u1); ns_mul(ri, s_inv, u2);
      ns_mul(ai, u1, au1); ns_mul(ai, u2, au2);
      jpt g = n_generator();
      q = n_identity();
      if (!(pk_inf && pk_inf[i])) { q.x = ld(pk_xy + 8 * i); q.y = ld(pk_xy + 8 * i + 4); q.z = fe_small(1); }
      r1 = n_multiply(&g, au1); r2 = n_multiply(&q, au2);
      ri_pt = n_padd(&r1, &r2);
      r_sum = n_padd(&r_sum, &ri_pt);
    }
  }
  u64 sum[4] = {0, 0, 0, 0}, t[4], ar[4];                                  /* 368-372 */
  for (size_t i = 0; i < n; ++i) {
    if (curve == 0) { ks_mul(a + 4 * i, r + 4 * i, ar); ks_add(sum, ar, t); }
    else { ns_mul(a + 4 * i, r + 4 * i, ar); ns_add(sum, ar, t); }
    memcpy(sum, t, 32);
  }
  if (detail) { stj(detail, r_sum); memcpy(detail + 12, sum, 32); }
  if (curve == 0 ? k_is_identity(&r_sum) : n_is_identity(&r_sum)) return 0;  /* 361-364 */
  fe x, y;
  u64 xs[4];
  if (curve == 0) {
    k_to_affine(&r_sum, &x, &y);                                           /* 376 */
    fe xr = k_mul(x, fe_small(1));                                         /* to_bytes = mont_reduce */
    if (ks_ge_n(xr.v)) return 2;                                           /* 381 */
    memcpy(xs, xr.v, 32);
  } else {
    n_to_affine(&r_sum, &x, &y);
    if (n_cmp(x.v, NS_N) >= 0) return 2;
    memcpy(xs, x.v, 32);
  }
  return xs[0] == sum[0] && xs[1] == sum[1] && xs[2] == sum[2] && xs[3] == sum[3];   /* 384 */
}

typedef struct { const u64 *r, *pk, *s, *k; const uint8_t *rinf, *pinf; uint8_t* out; size_t lo, hi; } ev_t;
static void* evworker(void* arg) {
  ev_t* j = (ev_t*)arg;
  for (size_t i = j->lo; i < j->hi; ++i)
    j->out[i] = (uint8_t)fo_ed25519_eddsa_verify(j->r + 8 * i, j->rinf ? j->rinf[i] : 0, j->pk + 8 * i,
                                                  j->pinf ? j->pinf[i] : 0, j->s + 4 * i, j->k + 4 * i);
  return NULL;
}
void fo_batch_ed25519_eddsa_verify(const u64* r_xy, const uint8_t* r_inf, const u64* pk_xy, const uint8_t* pk_inf,
                                   const u64* s, const u64* k, uint8_t* out, size_t n, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  pthread_t th[64];
  ev_t jobs[64];
  for (int t = 0; t < nthreads; ++t) {
    ev_t j = {r_xy, pk_xy, s, k, r_inf, pk_inf, out, n * t / nthreads, n * (t + 1) / nthreads};
    jobs[t] = j;
    pthread_create(&th[t], NULL, evworker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}

typedef struct { const unsigned char* d; const u64 *r, *s, *pk; const uint8_t* inf; uint8_t* out; size_t lo, hi; } pv_t;
static void* pvworker(void* arg) {
  pv_t* j = (pv_t*)arg;
  for (size_t i = j->lo; i < j->hi; ++i)
    j->out[i] = (uint8_t)fo_p256_ecdsa_verify(j->d + 32 * i, j->r + 4 * i, j->s + 4 * i, j->pk + 8 * i,
                                               j->inf ? j->inf[i] : 0);
  return NULL;
}
void fo_batch_p256_ecdsa_verify(const unsigned char* digests, const u64* r, const u64* s, const u64* pk_xy,
                                const uint8_t* pk_inf, uint8_t* out, size_t n, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  pthread_t th[64];
  pv_t jobs[64];
  for (int t = 0; t < nthreads; ++t) {
    pv_t j = {digests, r, s, pk_xy, pk_inf, out, n * t / nthreads, n * (t + 1) / nthreads};
    jobs[t] = j;
    pthread_create(&th[t], NULL, pvworker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}

void fo_batch_secp256k1_ecdsa_verify(const unsigned char* digests, const u64* r, const u64* s, const u64* pk_xy,
                                     const uint8_t* pk_inf, uint8_t* out, size_t n, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
  pthread_t th[256];
  vjob_t jobs[256];
  for (int t = 0; t < nthreads; ++t) {
    vjob_t v = {digests, r, s, pk_xy, pk_inf, out, n * (size_t)t / (size_t)nthreads, n * (size_t)(t + 1) / (size_t)nthreads};
    jobs[t] = v;
  }
  if (nthreads == 1) { vworker(&jobs[0]); return; }
  for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, vworker, &jobs[t]);
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}

/* ---- Ed25519 Scalar arithmetic for schnorr::batch_verify::<Ed25519, D> (schnorr.rs:264: signatures[i].s * a[i]) ----
 * impl Add for Scalar, ed25519.rs:1193-1239: the 256-bit sum (the carry out of the top limb is dropped), one
 * conditional subtraction of ORDER. */
static const u64 ES_ORDER[4] = {0x5812631A5CF5D3EDULL, 0x14DEF9DEA2F79CD6ULL, 0ULL, 0x1000000000000000ULL};
static int es_ge_order(const u64 r[4]) {                     /* the loops at 1215-1226, 1303-1312, 1354-1363 */
  for (int i = 3; i >= 0; --i) {
    if (r[i] < ES_ORDER[i]) return 0;
    if (r[i] > ES_ORDER[i]) return 1;
  }
  return 1;
}
static void es_sub_order(u64 r[4]) {                         /* 1229-1236, 1365-1372 */
  u64 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    const __int128 diff = (__int128)r[i] - (__int128)ES_ORDER[i] - (__int128)borrow;
    r[i] = (u64)diff;
    borrow = diff < 0 ? 1 : 0;
  }
}
static void es_add(const u64 a[4], const u64 b[4], u64 out[4]) {
  u64 r[4], carry = 0;
  for (int i = 0; i < 4; ++i) {
    const u128 sum = (u128)a[i] + (u128)b[i] + (u128)carry;
    r[i] = (u64)sum;
    carry = (u64)(sum >> 64);
  }
  if (es_ge_order(r)) es_sub_order(r);
  memcpy(out, r, sizeof r);
}
/* impl Mul for Scalar, ed25519.rs:1256-1376, AS THE RELEASE PROFILE RUNS IT: /root/reference/Cargo.toml:53-58
 * ([profile.release]: opt-level 3, thin lto, codegen-units 1 -- no `overflow-checks` key, so u128 `+=` wraps; a debug
 * build panics instead).  1268-1272 sum up to four 128-bit products into a u128 and 1278 adds the carry on top: both can
 * pass 2^128.  *overflowed = 1 when one of those additions wrapped (the operands on which a debug build panics). */
static void es_mul_release(const u64 a[4], const u64 b[4], u64 out[4], int* overflowed) {
  u128 product[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int ovf = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const u128 term = (u128)a[i] * (u128)b[j];
      const u128 sum = product[i + j] + term;                /* wraps modulo 2^128 */
      if (sum < term) ovf = 1;
      product[i + j] = sum;
    }
  u128 carry = 0;
  for (int i = 0; i < 8; ++i) {
    const u128 sum = product[i] + carry;
    if (sum < carry) ovf = 1;
    carry = sum >> 64;
    product[i] = sum & (u128)0xFFFFFFFFFFFFFFFFULL;
  }
  if (overflowed) *overflowed = ovf;
  u64 result[4], high[4];
  for (int i = 0; i < 4; ++i) { result[i] = (u64)product[i]; high[i] = (u64)product[i + 4]; }
  const int high_nonzero = (high[0] | high[1] | high[2] | high[3]) != 0;
  if (high_nonzero || es_ge_order(result)) {                 /* 1299-1316 */
    if (high_nonzero)
      for (int k = 0; k < 256; ++k) es_add(result, high, result);   /* 1347-1349: result += high_bits, 256 times */
    if (es_ge_order(result)) es_sub_order(result);           /* 1352-1373 */
  }
  memcpy(out, result, sizeof result);
}
int fo_ed25519_scalar_mul_release(const u64 a[4], const u64 b[4], u64 out[4]) {
  int ovf = 0;
  es_mul_release(a, b, out, &ovf);
  return ovf;
}

/* schnorr::batch_verify::<Ed25519, D> (schnorr.rs:194-290), release profile.  Arguments as for the two functions
 * below; 1 true, 0 false, 2 = the reference panics in to_affine (286: z.invert().unwrap() on a zero z of a point that
 * is not the identity, ed25519.rs:1805).  *debug_build_panics (optional) = 1 when some s_i * a_i wrapped a u128. */
int fo_ed25519_schnorr_batch_verify(const u64* pk_xy, const uint8_t* pk_inf, const u64* r_xy, const uint8_t* r_inf,
                                    const u64* s, const u64* a, const u64* e, size_t n, u64* sides, uint8_t* sides_inf,
                                    uint8_t* debug_build_panics) {
  if (sides) memset(sides, 0, 16 * sizeof(u64));
  if (sides_inf) sides_inf[0] = sides_inf[1] = 0;
  if (debug_build_panics) *debug_build_panics = 0;
  if (n == 0) return 0;
  for (size_t i = 0; i < n; ++i) {
    if (pk_inf && pk_inf[i]) return 0;
    if (r_inf && r_inf[i]) return 0;
  }
  ept g = e_generator();
  ept s_g = e_identity(), r_e_p = e_identity();
  for (size_t i = 0; i < n; ++i) {
    u64 sa[4];
    int ovf = 0;
    es_mul_release(s + 4 * i, a + 4 * i, sa, &ovf);
    if (ovf && debug_build_panics) *debug_build_panics = 1;
    ept t = e_multiply(&g, sa);
    s_g = e_padd(&s_g, &t);
    ept P = e_from_affine(pk_xy + 8 * i, 0);
    ept ep = e_multiply(&P, e + 4 * i);
    ept R = e_from_affine(r_xy + 8 * i, 0);
    ept rp = e_padd(&R, &ep);
    ept arp = e_multiply(&rp, a + 4 * i);
    r_e_p = e_padd(&r_e_p, &arp);
  }
  if (!e_is_identity(&s_g) && fe_is_zero(&s_g.z)) return 2;      /* to_affine(s_g) is evaluated first */
  if (!e_is_identity(&r_e_p) && fe_is_zero(&r_e_p.z)) return 2;
  fe x1, y1, x2, y2;
  int i1 = e_to_affine(&s_g, &x1, &y1), i2 = e_to_affine(&r_e_p, &x2, &y2);
  if (sides) { st(sides, x1); st(sides + 4, y1); st(sides + 8, x2); st(sides + 12, y2); }
  if (sides_inf) { sides_inf[0] = (uint8_t)i1; sides_inf[1] = (uint8_t)i2; }
  return (fe_eq(&x1, &x2) && fe_eq(&y1, &y2)) || (i1 && i2);     /* AffinePoint::ct_eq 1759-1763 */
}

/* forge-ec-signature/src/schnorr.rs:194-290  schnorr::batch_verify::<Secp256k1, D>, with the
 * per-signature challenges e_i = from_bytes_reduced(H(R || P || m)) (236-256) and the random
 * weights a_i (228-233, OsRng) supplied by the caller as raw scalar limbs.
 *   - 197-199: n == 0 -> false.
 *   - 204-225: the two is_on_curve tests are written `!x.unwrap_u8() == 1`; `!` on a u8 is a
 *     bitwise NOT (254 or 255), never 1, so they can never reject.  The two is_identity tests do.
 *   - 262-281: s_g += multiply(G, s_i * a_i);  r_e_p += multiply(from_affine(R_i) +
 *     multiply(from_affine(P_i), e_i), a_i), both folds strictly in index order from identity().
 *   - 286: to_affine(s_g).ct_eq(to_affine(r_e_p)) with AffinePoint::ct_eq (1292-1296).
 * sides (optional, 16 limbs): x,y of to_affine(s_g) then of to_affine(r_e_p); sides_inf (optional, 2). */
int fo_secp256k1_schnorr_batch_verify(const u64* pk_xy, const uint8_t* pk_inf, const u64* r_xy,
                                      const uint8_t* r_inf, const u64* s, const u64* a, const u64* e, size_t n,
                                      u64* sides, uint8_t* sides_inf) {
  if (sides) memset(sides, 0, 16 * sizeof(u64));
  if (sides_inf) sides_inf[0] = sides_inf[1] = 0;
  if (n == 0) return 0;
  for (size_t i = 0; i < n; ++i) {
    if (pk_inf && pk_inf[i]) return 0;
    if (r_inf && r_inf[i]) return 0;
  }
  jpt g = k_generator();
  jpt s_g = k_identity(), r_e_p = k_identity();
  for (size_t i = 0; i < n; ++i) {
    u64 sa[4];
    ks_mul(s + 4 * i, a + 4 * i, sa);                          /* impl Mul for Scalar 2410-2456 */
    jpt t = k_multiply(&g, sa);
    s_g = k_padd(&s_g, &t);                                    /* AddAssign 1543-1547 */
    jpt P = {ld(pk_xy + 8 * i), ld(pk_xy + 8 * i + 4), fe_small(1)};   /* from_affine 1365-1373 */
    jpt ep = k_multiply(&P, e + 4 * i);
    jpt R = {ld(r_xy + 8 * i), ld(r_xy + 8 * i + 4), fe_small(1)};
    jpt rp = k_padd(&R, &ep);
    jpt arp = k_multiply(&rp, a + 4 * i);
    r_e_p = k_padd(&r_e_p, &arp);
  }
  fe x1, y1, x2, y2;
  int i1 = k_to_affine(&s_g, &x1, &y1), i2 = k_to_affine(&r_e_p, &x2, &y2);
  if (sides) { st(sides, x1); st(sides + 4, y1); st(sides + 8, x2); st(sides + 12, y2); }
  if (sides_inf) { sides_inf[0] = (uint8_t)i1; sides_inf[1] = (uint8_t)i2; }
  return (fe_eq(&x1, &x2) && fe_eq(&y1, &y2)) || (i1 && i2);
}

/* PointAffine::to_bytes -> [u8; 33], the compressed SEC1-style encoding every curve implements the
 * same way (secp256k1.rs:875-896, p256.rs:1558-1578, ed25519.rs:1505-1525): 0x00 + zeros for the
 * identity, else 0x02 | (y.to_bytes()[31] & 1) followed by x.to_bytes().  FieldElement::to_bytes:
 *   secp256k1 (138-178): mont_reduce (= Mul by raw 1), big-endian;
 *   P-256 (288-300): the raw limbs, big-endian;
 *   Ed25519 (295-310): reduce(), LITTLE-endian -- so byte 31 is the most significant byte and the
 *     "parity" bit the reference takes is bit 248 of y, and x is emitted little-endian.
 * forge-ec-encoding/src/point.rs:38-67 (CompressedPoint::from_affine) builds the same bytes. */
static void field_to_bytes(int curve, fe a, unsigned char out[32]) {
  if (curve == FO_SECP256K1) a = k_mul(a, fe_small(1));
  else if (curve == FO_ED25519) e_reduce(&a);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      unsigned char b = (unsigned char)(a.v[i] >> (8 * j));
      if (curve == FO_ED25519) out[i * 8 + j] = b; else out[31 - (i * 8 + j)] = b;
    }
}
void fo_batch_compress(int curve, const u64* xy, const uint8_t* inf, unsigned char* out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    unsigned char* o = out + 33 * i;
    memset(o, 0, 33);
    if (inf && inf[i]) continue;
    unsigned char yb[32];
    field_to_bytes(curve, ld(xy + 8 * i + 4), yb);
    o[0] = (yb[31] & 1) ? 0x03 : 0x02;
    field_to_bytes(curve, ld(xy + 8 * i), o + 1);
  }
}

/* ================================================================================================
 * Point DECODING (the half round 1 left out).
 *
 *  fo_batch_decompress          PointAffine::from_bytes(&[u8; 33]) of each curve:
 *                               secp256k1.rs:896-976, p256.rs:1580-1639, ed25519.rs:1526-1582
 *  fo_batch_encode_uncompressed UncompressedPoint::from_affine, forge-ec-encoding/src/point.rs:186-211
 *  fo_batch_decode_uncompressed UncompressedPoint::to_affine, point.rs:214-281 (generic over the curve:
 *                               C::Field::from_bytes, Mul, get_a/get_b, then C::PointAffine::new)
 *
 * ok[i] = 1 when the reference returns Some(point), 0 for None (xy and inf are then zero).  The
 * arithmetic is the reference's own, quirks included:
 *  - secp256k1 FieldElement::sqrt (112-131) raises to the four-limb exponent [0xFF0C, 0xFFFF, 0xFFFE,
 *    0x3FFF] -- (p+1)/4 written as 16-bit words into 64-bit limbs -- so it returns None for essentially
 *    every input; the x^3 + 7 it is applied to mixes a non-Montgomery square() and a raw 7 into
 *    Montgomery-form values; whatever survives is re-checked by is_on_curve (978-1004).
 *  - P-256: the inherent sqrt (320-339) with its own exponent; no on-curve check at the end.
 *  - Ed25519 FieldElement::from_bytes (315-357) rejects a value as soon as ANY limb exceeds the same limb
 *    of p (an early `return` inside the comparison loop), not only values >= p; from_bytes evaluates the
 *    MONTGOMERY-curve equation x^3 + a x^2 + x with a = 0x7FFFFFDA on what it calls x.
 * ================================================================================================ */
static fe k_one(void) { return fe_small(1); }
/* secp256k1.rs:715-735 trait pow: LSB first, `result *= base` when the bit is set, base = base.square() */
static fe k_pow(fe a, const u64* e, int limbs) {
  fe result = k_one(), base = a;
  for (int w = 0; w < limbs; ++w)
    for (int j = 0; j < 64; ++j) {
      if ((e[w] >> j) & 1) result = k_mul(result, base);
      base = k_sqr(base);
    }
  return result;
}
static int k_sqrt_inherent(fe a, fe* out) {      /* 112-131 */
  static const u64 e[4] = {0xFF0C, 0xFFFF, 0xFFFE, 0x3FFF};
  fe s = k_pow(a, e, 4);
  fe s2 = k_sqr(s);
  *out = s;
  return fe_eq(&s2, &a);
}
static fe k_to_montgomery(fe a) {                 /* 219-235 */
  fe r2 = {{0x000E9F61ULL, 0x07A20000ULL, 0x00000100ULL, 0}};
  return k_mul(a, r2);
}
static int k_from_bytes(const unsigned char b[32], fe* out) {   /* 182-212: big-endian, < p, then to_montgomery */
  u64 l[4] = {0, 0, 0, 0};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) l[3 - i] |= (u64)b[i * 8 + j] << (56 - j * 8);
  int valid = k_cmp_p(l) < 0;
  fe raw = {{l[0], l[1], l[2], l[3]}};
  fe m = k_to_montgomery(raw);
  *out = valid ? m : fe_small(0);
  return valid;
}
static int k_affine_new(fe x, fe y) {             /* PointAffine::new 856-869 (= is_on_curve 978-1004) */
  fe x3 = k_mul(k_sqr(x), x);
  fe rhs = k_add(x3, k_to_montgomery(fe_small(7)));
  fe y2 = k_sqr(y);
  return fe_eq(&y2, &rhs);
}
static int k_decompress(const unsigned char* in, fe* x, fe* y, int* inf) {   /* 896-976 */
  *inf = 0;
  if (in[0] == 0x00) { *x = fe_small(0); *y = fe_small(0); *inf = 1; return 1; }
  if (in[0] != 0x02 && in[0] != 0x03) return 0;
  if (!k_from_bytes(in + 1, x)) return 0;
  fe xs = k_sqr(*x);
  fe xc = k_mul(xs, *x);
  fe y2 = k_add(xc, fe_small(7));               /* `seven` is the RAW 7 here (935) */
  fe ye;
  if (!k_sqrt_inherent(y2, &ye)) return 0;
  fe yo = k_neg(ye);
  fe red = k_mul(ye, fe_small(1));              /* to_bytes: mont_reduce; byte 31 = least significant */
  int parity = (int)(red.v[0] & 1);
  int want_odd = in[0] == 0x03;
  *y = (want_odd ^ parity) ? yo : ye;
  return k_affine_new(*x, *y);                  /* is_on_curve */
}

static int n_from_bytes(const unsigned char b[32], fe* out) {   /* p256.rs:303-317 */
  u64 l[4] = {0, 0, 0, 0};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) l[i] |= (u64)b[31 - (i * 8 + j)] << (j * 8);
  fe r = {{l[0], l[1], l[2], l[3]}};
  *out = r;
  return n_cmp(l, N_P) < 0;
}
static const u64 N_B[4] = {0x3BCE3C3E27D2604BULL, 0x651D06B0CC53B0F6ULL, 0xB3EBBD55769886BCULL, 0x5AC635D8AA3A93E7ULL};
static fe n_rhs(fe x) {                           /* x^3 - 3x + b as 1536-1543 / 1612-1618 spell it */
  fe x2 = n_sqr(x);
  fe x3 = n_mul(x2, x);
  fe three_x = n_mul(fe_small(3), x);
  fe b = {{N_B[0], N_B[1], N_B[2], N_B[3]}};
  return n_add(n_sub(x3, three_x), b);
}
static int n_sqrt_inherent(fe a, fe* out) {       /* 320-339 */
  static const u64 e[4] = {0xC0000000ULL, 0x40000000ULL, 0x4000000000000000ULL, 0x40000000C0000000ULL};
  fe s = n_pow(a, e);
  fe s2 = n_sqr(s);
  *out = s;
  return fe_eq(&s2, &a);
}
static int n_decompress(const unsigned char* in, fe* x, fe* y, int* inf) {   /* 1580-1639 */
  *inf = 0;
  if (in[0] == 0x00) { *x = fe_small(0); *y = fe_small(0); *inf = 1; return 1; }
  if (in[0] != 0x02 && in[0] != 0x03) return 0;
  if (!n_from_bytes(in + 1, x)) return 0;
  fe y2 = n_rhs(*x);
  fe r;
  if (!n_sqrt_inherent(y2, &r)) return 0;
  int parity = (int)(r.v[0] & 1);               /* to_bytes (288-300) is the raw limbs, big-endian */
  if (parity != (in[0] == 0x03)) r = n_neg(r);
  *y = r;
  return 1;                                      /* no curve check here (1638) */
}
static int n_affine_new(fe x, fe y) {             /* 1535-1552 */
  fe rhs = n_rhs(x);
  fe y2 = n_sqr(y);
  return fe_eq(&y2, &rhs);
}

static const u64 E_SQRT_M1[4] = {0xC4EE1B274A0EA0B0ULL, 0x2F431806AD2FE478ULL, 0x2B4D00993DFBD7A7ULL, 0x2B8324804FC1DF0BULL};
static int e_from_bytes(const unsigned char b[32], fe* out) {   /* ed25519.rs:315-357 */
  u64 l[4];
  for (int i = 0; i < 4; ++i) {
    l[i] = 0;
    for (int j = 0; j < 8; ++j) l[i] |= (u64)b[i * 8 + j] << (8 * j);
  }
  int is_less = 0, is_equal = 1;
  for (int i = 3; i >= 0; --i) {
    int lt = l[i] < E_P[i], eq = l[i] == E_P[i], gt = l[i] > E_P[i];
    is_less |= is_equal & lt;
    is_equal &= eq;
    if (gt) { *out = fe_small(0); return 0; }   /* 346-348: returns None whatever the higher limbs said */
  }
  fe r = {{l[0], l[1], l[2], l[3]}};
  *out = r;
  return is_less;
}
static int e_sqrt(fe a, fe* out) {                /* 359-402 */
  static const u64 e1[4] = {0x7FFFFFFFFFFFFFF6ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x3FFFFFFFFFFFFFFFULL};
  static const u64 e2[4] = {0x1FFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x0FFFFFFFFFFFFFFFULL};
  fe leg = e_pow(a, e1);
  fe one = fe_small(1), zero = fe_small(0);
  if (!(fe_eq(&leg, &one) | fe_eq(&leg, &zero))) return 0;
  fe cand = e_pow(a, e2);
  fe c2 = e_sqr(cand);
  int ok1 = fe_eq(&c2, &a);
  fe sm1 = {{E_SQRT_M1[0], E_SQRT_M1[1], E_SQRT_M1[2], E_SQRT_M1[3]}};
  fe alt = e_mul(cand, sm1);
  fe a2 = e_sqr(alt);
  int ok2 = fe_eq(&a2, &a);
  *out = ok2 ? alt : cand;
  return ok1 | ok2;
}
static int e_decompress(const unsigned char* in, fe* x, fe* y, int* inf) {   /* 1526-1582 */
  *inf = 0;
  if (in[0] == 0x00) { *x = fe_small(0); *y = fe_small(0); *inf = 1; return 1; }
  if (in[0] != 0x02 && in[0] != 0x03) return 0;
  if (!e_from_bytes(in + 1, x)) return 0;
  fe x2 = e_sqr(*x);
  fe x3 = e_mul(x2, *x);
  fe a = fe_small(0x7FFFFFDAULL);
  fe y2 = e_add(e_add(x3, e_mul(a, x2)), *x);
  fe r;
  if (!e_sqrt(y2, &r)) return 0;
  fe red = r;
  e_reduce(&red);                                /* to_bytes (295-310): reduce(), little-endian: byte 31 is the TOP byte */
  int parity = (int)((red.v[3] >> 56) & 1);
  if (parity != (in[0] == 0x03)) r = e_neg(r);
  *y = r;
  return 1;
}
static int e_affine_new(fe x, fe y) {             /* 1476-1498 */
  fe x2 = e_sqr(x), y2 = e_sqr(y);
  fe x2y2 = e_mul(x2, y2);
  fe d = {{E_D[0], E_D[1], E_D[2], E_D[3]}};
  fe lhs = e_add(e_neg(x2), y2);
  fe rhs = e_add(fe_small(1), e_mul(d, x2y2));
  return fe_eq(&lhs, &rhs);
}

void fo_batch_decompress(int curve, const unsigned char* in, u64* xy, uint8_t* inf, uint8_t* ok, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    fe x = fe_small(0), y = fe_small(0);
    int f = 0, v;
    const unsigned char* b = in + 33 * i;
    if (curve == FO_SECP256K1) v = k_decompress(b, &x, &y, &f);
    else if (curve == FO_P256) v = n_decompress(b, &x, &y, &f);
    else v = e_decompress(b, &x, &y, &f);
    if (!v) { x = fe_small(0); y = fe_small(0); f = 0; }
    st(xy + 8 * i, x);
    st(xy + 8 * i + 4, y);
    inf[i] = (uint8_t)f;
    ok[i] = (uint8_t)v;
  }
}

/* UncompressedPoint::from_affine (point.rs:186-211): 0x00 + 64 zero bytes for the identity, else
 * 0x04 || x.to_bytes() || y.to_bytes() */
void fo_batch_encode_uncompressed(int curve, const u64* xy, const uint8_t* inf, unsigned char* out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    unsigned char* o = out + 65 * i;
    memset(o, 0, 65);
    if (inf && inf[i]) continue;
    o[0] = 0x04;
    field_to_bytes(curve, ld(xy + 8 * i), o + 1);
    field_to_bytes(curve, ld(xy + 8 * i + 4), o + 33);
  }
}

/* UncompressedPoint::to_affine (point.rs:214-281), generic over C */
void fo_batch_decode_uncompressed(int curve, const unsigned char* in, u64* xy, uint8_t* inf, uint8_t* ok, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const unsigned char* b = in + 65 * i;
    fe x = fe_small(0), y = fe_small(0);
    int f = 0, v = 0;
    if (b[0] == 0x00) {                          /* 221-226: C::to_affine(&C::identity()) */
      f = 1;
      v = 1;
    } else if (b[0] == 0x04) {
      int vx, vy, on = 0;
      if (curve == FO_SECP256K1) {
        vx = k_from_bytes(b + 1, &x); vy = k_from_bytes(b + 33, &y);
        if (vx && vy) {
          fe x3 = k_mul(k_mul(x, x), x), y2 = k_mul(y, y);          /* `x * x`, not square() (251-253) */
          fe ax = k_mul(fe_small(0), x);                              /* get_a() = zero (2712-2715) */
          fe rhs = k_add(k_add(x3, ax), fe_small(7));                 /* get_b() = raw 7 (2717-2720) */
          on = fe_eq(&y2, &rhs) && k_affine_new(x, y);
        }
      } else if (curve == FO_P256) {
        vx = n_from_bytes(b + 1, &x); vy = n_from_bytes(b + 33, &y);
        if (vx && vy) {
          fe x3 = n_mul(n_mul(x, x), x), y2 = n_mul(y, y);
          fe a = {{0xFFFFFFFCULL, 0xFFFFFFFFULL, 0xFFFFFFFEULL, 0xFFFFFFFFULL}};   /* get_a() as written (2177-2180) */
          fe bb = {{N_B[0], N_B[1], N_B[2], N_B[3]}};
          fe rhs = n_add(n_add(x3, n_mul(a, x)), bb);
          on = fe_eq(&y2, &rhs) && n_affine_new(x, y);
        }
      } else {
        vx = e_from_bytes(b + 1, &x); vy = e_from_bytes(b + 33, &y);
        if (vx && vy) {
          fe x3 = e_mul(e_mul(x, x), x), y2 = e_mul(y, y);
          fe a = {{0x7FFFFFFFFFFFFFEDULL, 0x7FFFFFFFFFFFFULL, 0, 0}};             /* get_a() as written (2107-2110) */
          fe rhs = e_add(e_add(x3, e_mul(a, x)), fe_small(0));                     /* get_b() = zero */
          on = fe_eq(&y2, &rhs) && e_affine_new(x, y);
        }
      }
      v = vx && vy && on;
    }
    if (!v || f) { x = fe_small(0); y = fe_small(0); }
    if (!v) f = 0;
    st(xy + 8 * i, x);
    st(xy + 8 * i + 4, y);
    inf[i] = (uint8_t)f;
    ok[i] = (uint8_t)v;
  }
}

void fo_batch_to_affine(int curve, const u64* points, u64* xy, uint8_t* inf, size_t n, int nthreads) {
  job_t j = {3, curve, NULL, NULL, points, xy, inf, 0, 0};
  run_jobs(j, n, nthreads);
}

/* ---- KeyExchange::derive_shared_secret (secp256k1.rs:1884-1904, p256.rs:2281-2302) ---------------------------
 * curve 0 (Secp256k1): multiply(from_affine(pk), sk), to_affine; identity -> Err(InvalidEncoding) (status 2);
 *   else Ok(x.to_bytes()).  No validation of the public key.
 * curve 1 (P256): validate_public_key (2304-2312) = !is_identity & validate_point (2187-2191) = is_on_curve
 *   (1636-1656), else Err(InvalidPublicKey) (status 1); then as above with Err(KeyExchangeError) (status 2).
 * Ed25519 does not implement KeyExchange.  out: 32 bytes per element (zero unless status 0). */
int fo_ecdh(int curve, const u64 sk[4], const u64 pk_xy[8], int pk_inf, unsigned char out[32]) {
  memset(out, 0, 32);
  fe x, y;
  if (curve == 0) {
    jpt q = k_identity();
    if (!pk_inf) { q.x = ld(pk_xy); q.y = ld(pk_xy + 4); q.z = fe_small(1); }
    jpt sp = k_multiply(&q, sk);
    if (k_to_affine(&sp, &x, &y)) return 2;
    field_to_bytes(0, x, out);
    return 0;
  }
  if (curve != 1) return -1;
  fe px = ld(pk_xy), py = ld(pk_xy + 4);
  fe lhs = n_mul(py, py);                       /* y.square() = self * self */
  fe rhs = n_rhs(px);
  int on_curve = pk_inf ? 1 : (lhs.v[0] == rhs.v[0] && lhs.v[1] == rhs.v[1] && lhs.v[2] == rhs.v[2] && lhs.v[3] == rhs.v[3]);
  if (pk_inf || !on_curve) return 1;
  jpt q = {px, py, fe_small(1)};
  jpt sp = n_multiply(&q, sk);
  if (n_to_affine(&sp, &x, &y)) return 2;
  field_to_bytes(1, x, out);
  return 0;
}
typedef struct { int curve; const u64 *sk, *pk; const uint8_t* inf; unsigned char* out; uint8_t* st; size_t lo, hi; } dh_t;
static void* dhworker(void* arg) {
  dh_t* j = (dh_t*)arg;
  for (size_t i = j->lo; i < j->hi; ++i)
    j->st[i] = (uint8_t)fo_ecdh(j->curve, j->sk + 4 * i, j->pk + 8 * i, j->inf ? j->inf[i] : 0, j->out + 32 * i);
  return NULL;
}
void fo_batch_ecdh(int curve, const u64* sk, const u64* pk_xy, const uint8_t* pk_inf, unsigned char* out, uint8_t* status,
                   size_t n, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  pthread_t th[64];
  dh_t jobs[64];
  for (int t = 0; t < nthreads; ++t) {
    dh_t j = {curve, sk, pk_xy, pk_inf, out, status, n * t / nthreads, n * (t + 1) / nthreads};
    jobs[t] = j;
    pthread_create(&th[t], NULL, dhworker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}

/* ---- Curve::validate_point -----------------------------------------------------------------------------------
 * Secp256k1 (secp256k1.rs:2722-2726) and P256 (p256.rs:2187-2191) override it with PointAffine::is_on_curve
 * (978-1004 / 1636-1656; the infinity flag counts as on the curve).  Ed25519 keeps the trait default
 * (forge-ec-core/src/lib.rs:905-925): is_on_curve (ed25519.rs:1719-1744) AND
 * multiply(clear_cofactor(from_affine(p)), order()).is_identity(), with the default clear_cofactor (885-897) =
 * multiply(p, Scalar::from(8)) and order() = L (ed25519.rs:75-80, 2099-2101). */
int fo_validate_point(int curve, const u64 xy[8], int inf) {
  fe x = ld(xy), y = ld(xy + 4);
  if (curve == 0) return inf ? 1 : k_affine_new(x, y);
  if (curve == 1) {
    if (inf) return 1;
    fe lhs = n_mul(y, y), rhs = n_rhs(x);
    return lhs.v[0] == rhs.v[0] && lhs.v[1] == rhs.v[1] && lhs.v[2] == rhs.v[2] && lhs.v[3] == rhs.v[3];
  }
  if (curve != 2) return -1;
  int on_curve = inf ? 1 : e_affine_new(x, y);
  static const u64 eight[4] = {8, 0, 0, 0};
  static const u64 order[4] = {0x5812631A5CF5D3EDULL, 0x14DEF9DEA2F79CD6ULL, 0, 0x1000000000000000ULL};
  ept p = e_from_affine(xy, inf);
  ept cleared = e_multiply(&p, eight);
  ept sp = e_multiply(&cleared, order);
  return on_curve & e_is_identity(&sp);
}
typedef struct { int curve; const u64* xy; const uint8_t* inf; uint8_t* ok; size_t lo, hi; } vp_t;
static void* vpworker(void* arg) {
  vp_t* j = (vp_t*)arg;
  for (size_t i = j->lo; i < j->hi; ++i) j->ok[i] = (uint8_t)fo_validate_point(j->curve, j->xy + 8 * i, j->inf ? j->inf[i] : 0);
  return NULL;
}
void fo_batch_validate_point(int curve, const u64* xy, const uint8_t* inf, uint8_t* ok, size_t n, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  pthread_t th[64];
  vp_t jobs[64];
  for (int t = 0; t < nthreads; ++t) {
    vp_t j = {curve, xy, inf, ok, n * t / nthreads, n * (t + 1) / nthreads};
    jobs[t] = j;
    pthread_create(&th[t], NULL, vpworker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}

/* ---- forge-ec-signature/src/schnorr.rs:90-140  Schnorr::<C, D>::verify per signature, for the three curves,
 * from the point computation on: the caller hashes (107-123: e = from_bytes_reduced(H(R || P || m))) and keeps the
 * two message special cases (92-99).  1 true, 0 false, 2 = the reference panics (Ed25519 only: to_affine unwraps
 * the inverse of a zero z of a point that is not the identity, ed25519.rs:1805).
 *   103-105  an infinite signature point is rejected
 *   125-126  s_g = multiply(G, s), e_p = multiply(from_affine(pk), e)
 *   129-134  e_p_affine = to_affine(e_p); PointAffine::new(x, -y) re-validates the curve equation under the
 *            reference's own arithmetic (secp256k1.rs:856-869, p256.rs:1535-1552, ed25519.rs:1477-1498) and its None
 *            is `return false` -- which is what happens to practically every real input on secp256k1 and P-256
 *   136-139  r' = s_g + from_affine(neg), to_affine
 *   142      AffinePoint::ct_eq: (x == x && y == y) | (infinity & infinity) */
int fo_schnorr_verify(int curve, const u64 pk_xy[8], int pk_inf, const u64 r_xy[8], int r_inf, const u64 s[4],
                      const u64 e[4]) {
  if (r_inf) return 0;
  fe sx = ld(r_xy), sy = ld(r_xy + 4), rx, ry;
  int ri;
  if (curve == FO_SECP256K1 || curve == FO_P256) {
    const int k = curve == FO_SECP256K1;
    jpt g = k ? k_generator() : n_generator();
    jpt s_g = k ? k_multiply(&g, s) : n_multiply(&g, s);
    jpt P = {ld(pk_xy), ld(pk_xy + 4), fe_small(1)};
    if (pk_inf) P = k ? k_identity() : n_identity();                       /* from_affine of the identity */
    jpt e_p = k ? k_multiply(&P, e) : n_multiply(&P, e);
    fe x, y;
    if (k) (void)k_to_affine(&e_p, &x, &y); else (void)n_to_affine(&e_p, &x, &y);   /* (0, 0) for the identity */
    fe ny = k ? k_neg(y) : n_neg(y);
    if (!(k ? k_affine_new(x, ny) : n_affine_new(x, ny))) return 0;
    jpt neg = {x, ny, fe_small(1)};                                        /* new() builds a finite point */
    jpt rp = k ? k_padd(&s_g, &neg) : n_padd(&s_g, &neg);
    ri = k ? k_to_affine(&rp, &rx, &ry) : n_to_affine(&rp, &rx, &ry);
  } else if (curve == FO_ED25519) {
    ept g = e_generator();
    ept s_g = e_multiply(&g, s);
    ept P = e_from_affine(pk_xy, pk_inf);
    ept e_p = e_multiply(&P, e);
    if (!e_is_identity(&e_p) && fe_is_zero(&e_p.z)) return 2;
    fe x, y;
    (void)e_to_affine(&e_p, &x, &y);
    fe ny = e_neg(y);
    if (!e_affine_new(x, ny)) return 0;
    u64 nxy[8];
    st(nxy, x); st(nxy + 4, ny);
    ept neg = e_from_affine(nxy, 0);
    ept rp = e_padd(&s_g, &neg);
    if (!e_is_identity(&rp) && fe_is_zero(&rp.z)) return 2;
    ri = e_to_affine(&rp, &rx, &ry);
  } else {
    return -1;
  }
  return (fe_eq(&rx, &sx) && fe_eq(&ry, &sy)) || (ri && r_inf);
}
typedef struct { int curve; const u64 *pk, *r, *s, *e; const uint8_t *pk_inf, *r_inf; uint8_t* st; size_t lo, hi; } sv_t;
static void* svworker(void* arg) {
  sv_t* j = (sv_t*)arg;
  for (size_t i = j->lo; i < j->hi; ++i)
    j->st[i] = (uint8_t)fo_schnorr_verify(j->curve, j->pk + 8 * i, j->pk_inf ? j->pk_inf[i] : 0, j->r + 8 * i,
                                          j->r_inf ? j->r_inf[i] : 0, j->s + 4 * i, j->e + 4 * i);
  return NULL;
}
void fo_batch_schnorr_verify(int curve, const u64* pk_xy, const uint8_t* pk_inf, const u64* r_xy, const uint8_t* r_inf,
                             const u64* s, const u64* e, uint8_t* status, size_t n, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  pthread_t th[64];
  sv_t jobs[64];
  for (int t = 0; t < nthreads; ++t) {
    sv_t j = {curve, pk_xy, r_xy, s, e, pk_inf, r_inf, status, n * t / nthreads, n * (t + 1) / nthreads};
    jobs[t] = j;
    pthread_create(&th[t], NULL, svworker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}

/* schnorr::batch_verify::<C, D> (schnorr.rs:194-290) for C = P256: the secp256k1 form above with the P-256 point and
 * scalar arithmetic (Scalar Mul p256.rs:1409-1432).  Same argument meaning, same outputs. */
int fo_p256_schnorr_batch_verify(const u64* pk_xy, const uint8_t* pk_inf, const u64* r_xy, const uint8_t* r_inf,
                                 const u64* s, const u64* a, const u64* e, size_t n, u64* sides, uint8_t* sides_inf) {
  if (sides) memset(sides, 0, 16 * sizeof(u64));
  if (sides_inf) sides_inf[0] = sides_inf[1] = 0;
  if (n == 0) return 0;
  for (size_t i = 0; i < n; ++i) {
    if (pk_inf && pk_inf[i]) return 0;
    if (r_inf && r_inf[i]) return 0;
  }
  jpt g = n_generator();
  jpt s_g = n_identity(), r_e_p = n_identity();
  for (size_t i = 0; i < n; ++i) {
    u64 sa[4];
    ns_mul(s + 4 * i, a + 4 * i, sa);
    jpt t = n_multiply(&g, sa);
    s_g = n_padd(&s_g, &t);
    jpt P = {ld(pk_xy + 8 * i), ld(pk_xy + 8 * i + 4), fe_small(1)};
    jpt ep = n_multiply(&P, e + 4 * i);
    jpt R = {ld(r_xy + 8 * i), ld(r_xy + 8 * i + 4), fe_small(1)};
    jpt rp = n_padd(&R, &ep);
    jpt arp = n_multiply(&rp, a + 4 * i);
    r_e_p = n_padd(&r_e_p, &arp);
  }
  fe x1, y1, x2, y2;
  int i1 = n_to_affine(&s_g, &x1, &y1), i2 = n_to_affine(&r_e_p, &x2, &y2);
  if (sides) { st(sides, x1); st(sides + 4, y1); st(sides + 8, x2); st(sides + 12, y2); }
  if (sides_inf) { sides_inf[0] = (uint8_t)i1; sides_inf[1] = (uint8_t)i2; }
  return (fe_eq(&x1, &x2) && fe_eq(&y1, &y2)) || (i1 && i2);
}
