// Host build of the device headers: exposes the same field/point/multiply routines the kernels
// inline, as a C shared library, so tests can diff them against the oracle WITHOUT a GPU and
// under -fsanitize=address,undefined.  Test infrastructure only.
#define FEC_HOST_EMUL 1
#include "../forge_ec_amd/csrc/secp256k1.hpp"
#include "../forge_ec_amd/csrc/p256.hpp"
#include "../forge_ec_amd/csrc/ed25519.hpp"
#include <string.h>
using namespace fecgpu;

static fe ld(const uint64_t* a) { fe r; for (int i = 0; i < 4; ++i) { r.w[2*i] = (u32)a[i]; r.w[2*i+1] = (u32)(a[i] >> 32); } return r; }
static void st(uint64_t* o, const fe& a) { for (int i = 0; i < 4; ++i) o[i] = (u64)a.w[2*i] | ((u64)a.w[2*i+1] << 32); }

extern "C" unsigned long he_rare_sqr_count() { return secp::fec_host_rare_sqr; }
extern "C" int he_field_op(int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  fe x = ld(a), y = b ? ld(b) : fe_zero(), r;
  if (curve == 0) r = op == 0 ? secp::add(x, y) : op == 1 ? secp::sub(x, y) : op == 2 ? secp::mul(x, y) : op == 3 ? secp::sqr(x) : secp::neg(x);
  else if (curve == 1) r = op == 0 ? p256::add(x, y) : op == 1 ? p256::sub(x, y) : op == 2 ? p256::mul(x, y) : op == 3 ? p256::sqr(x) : p256::neg(x);
  else r = op == 0 ? ed::add(x, y) : op == 1 ? ed::sub(x, y) : op == 2 ? ed::mul(x, y) : op == 3 ? ed::mul(x, x) : ed::neg(x);
  st(out, r);
  return 0;
}
template <class PT> static PT ldp3(const uint64_t* p) { PT r; r.x = ld(p); r.y = ld(p + 4); r.z = ld(p + 8); return r; }
template <class PT> static void stp3(uint64_t* o, const PT& p) { st(o, p.x); st(o + 4, p.y); st(o + 8, p.z); }
static ed::pt ldp4(const uint64_t* p) { ed::pt r; r.x = ld(p); r.y = ld(p + 4); r.z = ld(p + 8); r.t = ld(p + 12); return r; }
static void stp4(uint64_t* o, const ed::pt& p) { st(o, p.x); st(o + 4, p.y); st(o + 8, p.z); st(o + 12, p.t); }

// op: 0 add, 1 double
extern "C" int he_point_op(int curve, int op, const uint64_t* p, const uint64_t* q, uint64_t* out) {
  if (curve == 0) { auto a = ldp3<secp::pt>(p); stp3(out, op == 0 ? secp::padd(a, ldp3<secp::pt>(q)) : secp::pdouble(a)); }
  else if (curve == 1) { auto a = ldp3<p256::pt>(p); stp3(out, op == 0 ? p256::padd(a, ldp3<p256::pt>(q)) : p256::pdouble(a)); }
  else { auto a = ldp4(p); stp4(out, op == 0 ? ed::padd(a, ldp4(q)) : ed::padd(a, a)); }
  return 0;
}
extern "C" int he_multiply(int curve, const uint64_t* point, const uint64_t* scalar, uint64_t* out) {
  static thread_local u32 kw[8 * KSTRIDE];
  for (int i = 0; i < 4; ++i) { kw[(2*i) * KSTRIDE] = (u32)scalar[i]; kw[(2*i+1) * KSTRIDE] = (u32)(scalar[i] >> 32); }
  if (curve == 0) stp3(out, secp::multiply(ldp3<secp::pt>(point), kw));
  else if (curve == 1) stp3(out, p256::multiply(ldp3<p256::pt>(point), kw));
  else stp4(out, ed::multiply(ldp4(point), kw));
  return 0;
}

// Ed25519 fixed-base through the same table walk the kernel uses (table built by the same chain)
extern "C" int he_ed_multiply_fixed(const uint64_t* base, const uint64_t* scalar, uint64_t* out) {
  static thread_local u32 kw[8 * KSTRIDE];
  static thread_local u32 tab[256 * ed::ED_TSTRIDE];
  for (int i = 0; i < 4; ++i) { kw[(2*i) * KSTRIDE] = (u32)scalar[i]; kw[(2*i+1) * KSTRIDE] = (u32)(scalar[i] >> 32); }
  ed::pt a = ldp4(base);
  for (int j = 0; j < 256; ++j) {
    u32* e = tab + j * ed::ED_TSTRIDE;
    for (int i = 0; i < 8; ++i) { e[i] = a.x.w[i]; e[8+i] = a.y.w[i]; e[16+i] = a.z.w[i]; e[24+i] = a.t.w[i]; }
    a = ed::padd(a, a);
  }
  stp4(out, ed::multiply_fixed(ldp4(base), tab, kw));
  return 0;
}

extern "C" int he_to_affine(int curve, const uint64_t* p, uint64_t* xy) {
  fe x, y; lmask inf;
  if (curve == 0) inf = secp::to_affine(ldp3<secp::pt>(p), x, y);
  else if (curve == 1) inf = p256::to_affine(ldp3<p256::pt>(p), x, y);
  else inf = ed::to_affine(ldp4(p), x, y);
  st(xy, x); st(xy + 4, y);
  return inf ? 1 : 0;
}

// secp256k1 scalar field (mod n) as the reference implements it: op 0 = mul, 1 = inv, 4 = add
extern "C" int he_secp_scalar_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  fe x = ld(a), r;
  if (op == 0) r = secp::sc_mul(x, ld(b)); else if (op == 4) r = secp::sc_add(x, ld(b)); else r = secp::sc_inv(x);
  st(out, r);
  return 0;
}

// P-256 scalar field as the reference implements it (p256.rs:924-1020, 1409-1432): op 0 = mul, 1 = inv,
// 2 = the default ct_lt (a vs b), 3 = a >= n, 4 = add
extern "C" int he_p256_scalar_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  p256::sc x = p256::sc_of(ld(a));
  if (op == 2) return p256::sc_ct_lt_default(x, p256::sc_of(ld(b))) ? 1 : 0;
  if (op == 3) return p256::sc_ge_n(x) ? 1 : 0;
  st(out, p256::sc_fe(op == 0 ? p256::sc_mul(x, p256::sc_of(ld(b))) : op == 4 ? p256::sc_add(x, p256::sc_of(ld(b))) : p256::sc_inv(x)));
  return 0;
}

// ---- canonical-math mode (canon_curves.hpp): checked against oracle/canon_model.py ----
// curve: 0 = secp256k1, 1 = P-256
#include "../forge_ec_amd/csrc/canon_curves.hpp"
#define CANON_DISPATCH(curve, CALL) ((curve) == 0 ? CALL(csecp) : CALL(cp256))

template <class W>
static int t_field_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  fe x = ld(a), y = b ? ld(b) : fe_zero(), r;
  r = op == 0 ? W::add(x, y) : op == 1 ? W::sub(x, y) : op == 2 ? W::mul(x, y)
    : op == 3 ? W::sqr(x) : op == 4 ? W::neg(x) : W::inv(x);
  st(out, r);
  return 0;
}
// op: 0 add, 1 sub, 2 mul, 3 sqr, 4 neg, 5 inv
extern "C" int he_canon_field_op(int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
#define CALL(W) t_field_op<W>(op, a, b, out)
  if (curve == 2) return CALL(ced);
  return CANON_DISPATCH(curve, CALL);
#undef CALL
}
template <class W>
static u32* t_comb_table() {
  static u32* tab = nullptr;
  if (!tab) {
    tab = new u32[canon::COMB_WORDS];
    canon::aff g = W::generator();
    canon::jac b; b.x = g.x; b.y = g.y; b.z = fe_small(1);
    for (int i = 0; i < canon::COMB_WINDOWS; ++i) {
      canon::aff base;
      W::to_affine(b, base);
      W::comb_fill_window(tab, i, base);
      for (int d = 0; d < 4; ++d) b = W::jdouble(b);
    }
  }
  return tab;
}
extern "C" const u32* he_canon_comb_table(int curve) {
#define CALL(W) t_comb_table<W>()
  return CANON_DISPATCH(curve, CALL);
#undef CALL
}
static void canon_kw(u32* kw, const uint64_t* scalar) {
  for (int i = 0; i < 4; ++i) { kw[(2*i) * KSTRIDE] = (u32)scalar[i]; kw[(2*i+1) * KSTRIDE] = (u32)(scalar[i] >> 32); }
}
template <class W>
static int t_mul_base(const uint64_t* scalar, uint64_t* xy) {
  static thread_local u32 kw[8 * KSTRIDE];
  canon_kw(kw, scalar);
  canon::jac r = W::mul_base_comb(t_comb_table<W>(), kw);
  canon::aff a;
  lmask inf = W::to_affine(r, a);
  st(xy, a.x); st(xy + 4, a.y);
  return inf ? 1 : 0;
}
// returns status: 0 finite, 1 infinity
extern "C" int he_canon_mul_base(int curve, const uint64_t* scalar, uint64_t* xy) {
#define CALL(W) t_mul_base<W>(scalar, xy)
  return CANON_DISPATCH(curve, CALL);
#undef CALL
}
template <class W>
static int t_mul(const uint64_t* scalar, const uint64_t* pxy, uint64_t* xy) {
  static thread_local u32 kw[8 * KSTRIDE];
  static thread_local u32 table[canon::WIN_ENTRIES * canon::WIN_ENTRY_WORDS];
  canon_kw(kw, scalar);
  canon::aff base; base.x = ld(pxy); base.y = ld(pxy + 4);
  lmask ok = W::on_curve(base);
  canon::jac r = W::mul_window(base, kw, table);
  canon::aff a;
  lmask inf = W::to_affine(r, a);
  if (!ok) { a.x = fe_zero(); a.y = fe_zero(); }
  st(xy, a.x); st(xy + 4, a.y);
  return !ok ? 2 : inf ? 1 : 0;
}
// returns status: 0 finite, 1 infinity, 2 bad point
extern "C" int he_canon_mul(int curve, const uint64_t* scalar, const uint64_t* pxy, uint64_t* xy) {
#define CALL(W) t_mul<W>(scalar, pxy, xy)
  return CANON_DISPATCH(curve, CALL);
#undef CALL
}
template <class W>
static int t_point_op(int op, const uint64_t* p, const uint64_t* q, uint64_t* out_xy) {
  canon::jac a; a.x = ld(p); a.y = ld(p + 4); a.z = ld(p + 8);
  canon::jac b; if (q) { b.x = ld(q); b.y = ld(q + 4); b.z = ld(q + 8); }
  canon::jac r;
  if (op == 0) r = W::jdouble(a);
  else if (op == 1) r = W::jadd(a, b);
  else if (op == 2) { canon::aff qa; qa.x = b.x; qa.y = b.y; r = W::jadd_affine(a, qa, 0); }
  else r = W::jadd_window(a, b, 0);
  canon::aff o;
  lmask inf = W::to_affine(r, o);
  st(out_xy, o.x); st(out_xy + 4, o.y);
  return inf ? 1 : 0;
}
// Jacobian ops for exceptional-case tests: op 0 = jdouble, 1 = jadd (general), 2 = jadd_affine(p, q.xy), 3 = jadd_window
extern "C" int he_canon_point_op(int curve, int op, const uint64_t* p, const uint64_t* q, uint64_t* out_xy) {
#define CALL(W) t_point_op<W>(op, p, q, out_xy)
  return CANON_DISPATCH(curve, CALL);
#undef CALL
}
template <class W>
static int t_normalize(uint32_t* xy, const uint32_t* zbuf, unsigned char* status, size_t n) {
  const size_t lanes = (n + canon::NORM_GROUP - 1) / canon::NORM_GROUP;
  const size_t stride = (lanes + 63) / 64 * 64;
  for (size_t g = 0; g < stride; ++g) W::normalize_group(xy, zbuf, status, g, stride, n);
  return 0;
}
// the batched normalisation exactly as k_canon_normalize drives it
extern "C" int he_canon_normalize(int curve, uint32_t* xy, const uint32_t* zbuf, unsigned char* status, size_t n) {
#define CALL(W) t_normalize<W>(xy, zbuf, status, n)
  return CANON_DISPATCH(curve, CALL);
#undef CALL
}

// ---- canonical Ed25519 (curve id 2) ----
static u32* ed_comb_table() {
  static u32* tab = nullptr;
  if (!tab) {
    tab = new u32[canon::ED_COMB_WORDS];
    canon::ext b = ced::from_affine(ced::generator());
    for (int i = 0; i <= canon::COMB_WINDOWS; ++i) {
      canon::aff base = ced::to_affine(b);
      if (i < canon::COMB_WINDOWS) ced::comb_fill_window(tab, i, base);
      else ced::comb_store(tab, canon::COMB_WINDOWS * canon::ED_COMB_ENTRIES, base);
      for (int d = 0; d < 4; ++d) b = ced::dbl<true>(b);
    }
  }
  return tab;
}
extern "C" int he_ced_mul_base(const uint64_t* scalar, uint64_t* xy) {
  static thread_local u32 kw[8 * KSTRIDE];
  canon_kw(kw, scalar);
  canon::aff a = ced::to_affine(ced::mul_base_comb(ed_comb_table(), kw));
  st(xy, a.x); st(xy + 4, a.y);
  return 0;
}
extern "C" int he_ced_mul(const uint64_t* scalar, const uint64_t* pxy, uint64_t* xy) {
  static thread_local u32 kw[8 * KSTRIDE];
  static thread_local u32 table[canon::ED_WIN_ENTRIES * 32];
  canon_kw(kw, scalar);
  canon::aff base; base.x = ld(pxy); base.y = ld(pxy + 4);
  lmask ok = ced::on_curve(base);
  canon::aff a = ced::to_affine(ced::mul_window(base, kw, table));
  if (!ok) { a.x = fe_zero(); a.y = fe_zero(); }
  st(xy, a.x); st(xy + 4, a.y);
  return ok ? 0 : 2;
}
// op 0: double, 1: p + q (q through projective Niels), 2: p - q, 3: p + q with q affine Niels, 4: p - q affine Niels
extern "C" int he_ced_point_op(int op, const uint64_t* p, const uint64_t* q, uint64_t* out_xy) {
  canon::ext a; a.x = ld(p); a.y = ld(p + 4); a.z = ld(p + 8); a.t = ld(p + 12);
  canon::ext r;
  if (op == 0) r = ced::dbl<true>(a);
  else {
    canon::ext b; b.x = ld(q); b.y = ld(q + 4); b.z = ld(q + 8); b.t = ld(q + 12);
    if (op <= 2) r = ced::add_pniels(a, ced::to_pniels(b), op == 2 ? ~0ull : 0, 0);
    else {
      canon::aff ba = ced::to_affine(b);
      canon::niels nq; nq.ypx = ced::add(ba.y, ba.x); nq.ymx = ced::sub(ba.y, ba.x);
      nq.t2d = ced::mul(ced::mul(ba.x, ba.y), ced::d2());
      r = ced::add_niels(a, nq, op == 4 ? ~0ull : 0, 0);
    }
  }
  canon::aff o = ced::to_affine(r);
  // also check T: T * Z == X * Y
  fe lhs = ced::mul(r.t, r.z), rhs = ced::mul(r.x, r.y);
  st(out_xy, o.x); st(out_xy + 4, o.y);
  return fe_eq(lhs, rhs) ? 0 : 9;
}
extern "C" int he_ced_normalize(uint32_t* xy, const uint32_t* zbuf, unsigned char* status, size_t n) {
  const size_t lanes = (n + canon::NORM_GROUP - 1) / canon::NORM_GROUP;
  const size_t stride = (lanes + 63) / 64 * 64;
  for (size_t g = 0; g < stride; ++g) ced::normalize_group(xy, zbuf, status, g, stride, n);
  return 0;
}
// ---- scalar fields for canonical ECDSA: curve 0 secp256k1, 1 P-256 ----
// op 0: mmul(a, b) (a * b * R^-1 mod n); 1: inv_mont(a) (a^-1 * R mod n)
extern "C" int he_canon_scalar_op(int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  fe x = ld(a), y = b ? ld(b) : fe_zero(), r;
  if (curve == 0) r = op == 0 ? canon::Fn<canon::NSecp>::mmul(x, y) : canon::Fn<canon::NSecp>::inv_mont(x);
  else r = op == 0 ? canon::Fn<canon::NP256>::mmul(x, y) : canon::Fn<canon::NP256>::inv_mont(x);
  st(out, r);
  return 0;
}
// returns 1 if the range check passes; u1, u2 out
extern "C" int he_canon_ecdsa_scalars(int curve, const uint64_t* z, const uint64_t* r, const uint64_t* s, uint64_t* u1, uint64_t* u2) {
  fe a, b;
  lmask ok = curve == 0 ? canon::ecdsa_scalars<canon::NSecp>(ld(z), ld(r), ld(s), a, b)
                        : canon::ecdsa_scalars<canon::NP256>(ld(z), ld(r), ld(s), a, b);
  st(u1, a); st(u2, b);
  return ok ? 1 : 0;
}
extern "C" int he_canon_ecdsa_x_matches(int curve, const uint64_t* x, const uint64_t* r) {
  lmask m = curve == 0 ? canon::ecdsa_x_matches<canon::NSecp>(ld(x), ld(r)) : canon::ecdsa_x_matches<canon::NP256>(ld(x), ld(r));
  return m ? 1 : 0;
}
// the grouped scalar half exactly as k_canon_ecdsa_scalars drives it
extern "C" int he_canon_ecdsa_scalars_batch(int curve, const uint32_t* z, const uint32_t* r, const uint32_t* s, uint32_t* u1,
                                            uint32_t* u2, unsigned char* ok, size_t n) {
  const size_t lanes = (n + canon::NORM_GROUP - 1) / canon::NORM_GROUP;
  const size_t stride = (lanes + 63) / 64 * 64;
  for (size_t g = 0; g < stride; ++g) {
    if (curve == 0) canon::ecdsa_scalars_group<canon::NSecp>(z, r, s, u1, u2, ok, g, stride, n);
    else canon::ecdsa_scalars_group<canon::NP256>(z, r, s, u1, u2, ok, g, stride, n);
  }
  return 0;
}
// ---- GLV (secp256k1) ----
extern "C" int he_glv_decompose(const uint64_t* k, uint64_t* k1, uint64_t* k2) {
  fe a, b; lmask n1, n2;
  cglv::decompose(ld(k), a, n1, b, n2);
  st(k1, a); st(k2, b);
  return (n1 ? 1 : 0) | (n2 ? 2 : 0);
}
extern "C" int he_glv_mul(const uint64_t* scalar, const uint64_t* pxy, uint64_t* xy) {
  static thread_local u32 kw[8 * KSTRIDE];
  static thread_local u32 table[canon::WIN_ENTRIES * canon::WIN_ENTRY_WORDS];
  canon_kw(kw, scalar);
  canon::aff base; base.x = ld(pxy); base.y = ld(pxy + 4);
  canon::jac r = cglv::mul_window(base, kw, table);
  canon::aff a;
  lmask inf = csecp::to_affine(r, a);
  st(xy, a.x); st(xy + 4, a.y);
  return inf ? 1 : 0;
}
// ---- 8-bit comb (table built exactly as k_canon_build_comb8 does) ----
template <class W>
static u32* t_comb8_table() {
  static u32* tab = nullptr;
  if (!tab) {
    tab = new u32[canon::COMB8_WORDS];
    canon::aff g = W::generator();
    canon::jac b; b.x = g.x; b.y = g.y; b.z = fe_small(1);
    for (int w = 0; w < canon::COMB8_WINDOWS; ++w) {
      canon::aff base;
      W::to_affine(b, base);
      for (u32 j = 1; j <= (u32)canon::COMB8_ENTRIES; ++j) {
        canon::aff e;
        W::to_affine(W::small_multiple(base, j), e);
        W::comb8_store(tab, w, j, e);
      }
      for (int d = 0; d < 8; ++d) b = W::jdouble(b);
    }
  }
  return tab;
}
template <class W>
static int t_mul_base8(const uint64_t* scalar, uint64_t* xy) {
  static thread_local u32 kw[8 * KSTRIDE];
  canon_kw(kw, scalar);
  canon::jac r = W::mul_base_comb8(t_comb8_table<W>(), kw);
  canon::aff a;
  lmask inf = W::to_affine(r, a);
  st(xy, a.x); st(xy + 4, a.y);
  return inf ? 1 : 0;
}
extern "C" int he_canon_mul_base8(int curve, const uint64_t* scalar, uint64_t* xy) {
#define CALL(W) t_mul_base8<W>(scalar, xy)
  return CANON_DISPATCH(curve, CALL);
#undef CALL
}
static u32* ed_comb8_table() {
  static u32* tab = nullptr;
  if (!tab) {
    tab = new u32[canon::ED_COMB8_WORDS];
    canon::ext b = ced::from_affine(ced::generator());
    for (int w = 0; w <= canon::COMB8_WINDOWS; ++w) {
      canon::aff base = ced::to_affine(b);
      const u32 last = w == canon::COMB8_WINDOWS ? 1u : (u32)canon::ED_COMB8_ENTRIES;
      for (u32 j = 1; j <= last; ++j)
        ced::comb8_store(tab, (size_t)w * canon::ED_COMB8_ENTRIES + j - 1, ced::to_affine(ced::small_multiple(base, j)));
      for (int d = 0; d < 8; ++d) b = ced::dbl<true>(b);
    }
  }
  return tab;
}
extern "C" int he_ced_mul_base8(const uint64_t* scalar, uint64_t* xy) {
  static thread_local u32 kw[8 * KSTRIDE];
  canon_kw(kw, scalar);
  canon::aff a = ced::to_affine(ced::mul_base_comb8(ed_comb8_table(), kw));
  st(xy, a.x); st(xy + 4, a.y);
  return 0;
}
// ---- BIP-340 / EdDSA verification halves ----
extern "C" int he_bip340_prepare(const uint64_t* pkx, const uint64_t* r, const uint64_t* s, const uint64_t* e,
                                 uint64_t* pxy, uint64_t* u2) {
  canon::aff P; fe v;
  lmask ok = canon::bip340_prepare(ld(pkx), ld(r), ld(s), ld(e), P, v);
  st(pxy, P.x); st(pxy + 4, P.y); st(u2, v);
  return ok ? 1 : 0;
}
extern "C" int he_ed_decode(const uint64_t* enc, uint64_t* xy) {
  canon::aff a;
  lmask ok = canon::ed_decode(ld(enc), a);
  st(xy, a.x); st(xy + 4, a.y);
  return ok ? 1 : 0;
}
extern "C" int he_eddsa_prepare(const uint64_t* a, const uint64_t* r, const uint64_t* s, const uint64_t* h,
                                uint64_t* axy, uint64_t* rxy, uint64_t* u2) {
  canon::aff A, R; fe v;
  lmask ok = canon::eddsa_prepare(ld(a), ld(r), ld(s), ld(h), A, R, v);
  st(axy, A.x); st(axy + 4, A.y); st(rxy, R.x); st(rxy + 4, R.y); st(u2, v);
  return ok ? 1 : 0;
}
// general scalar-field op: curve 0 secp256k1 (mod n), 1 P-256 (mod n), 2 Ed25519 (mod l); op 0 = a*b+c, 1 = a^-1
extern "C" int he_canon_scalar_general(int curve, int op, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out) {
  fe x = ld(a), y = b ? ld(b) : fe_zero(), z = c ? ld(c) : fe_zero(), r;
  if (curve == 0) r = canon::scalar_op<canon::NSecp>(op, x, y, z);
  else if (curve == 1) r = canon::scalar_op<canon::NP256>(op, x, y, z);
  else r = canon::scalar_op<canon::NEd>(op, x, y, z);
  st(out, r);
  return 0;
}
