// kernels_ecdsa.hip -- Ecdsa::<C, D>::verify (forge-ec-signature/src/ecdsa.rs:213-281) with the digest given,
// as a pipeline around the single-multiplication kernels:
//   k_ecdsa_pre<E>     215-251: zero / range checks, h = Scalar::from_bytes(digest), s^-1, u1 = h * s^-1,
//                      u2 = r * s^-1 in the curve's scalar field AS THE REFERENCE IMPLEMENTS IT, and
//                      Q = from_affine(pk); writes u1, u2, Q and one flag byte per signature
//   <curve>_launch_mul 254-255: multiply(G, u1) (fixed base) and multiply(Q, u2) with the curve's own kernel
//   k_ecdsa_finish<E>  256-274: R = r1 + r2, identity check, to_affine, field_to_bytes -> Scalar::from_bytes,
//                      comparison with r; writes status 1 valid / 0 invalid / 2 where the reference panics
//                      (CtOption::unwrap on None at 239 or 271)
// One signature per lane in the two small kernels; the multiplications are the measured hot-path kernels.
#include <hip/hip_runtime.h>

#include "../../include/fecgpu.h"
#include "ed25519.hpp"
#include "kernels.hpp"
#include "cu_split.hpp"
#include "p256.hpp"
#include "secp256k1.hpp"
#include "staging.hpp"

namespace fecgpu {

namespace {

enum : unsigned char { F_GO = 0, F_FALSE = 1, F_PANIC = 2 };

FEC_DEV fe load8(const u32* g) {
  const uint4 a = *reinterpret_cast<const uint4*>(g), b = *reinterpret_cast<const uint4*>(g + 4);
  fe r;
  r.w[0] = a.x; r.w[1] = a.y; r.w[2] = a.z; r.w[3] = a.w;
  r.w[4] = b.x; r.w[5] = b.y; r.w[6] = b.z; r.w[7] = b.w;
  return r;
}
FEC_DEV void store8(u32* g, const fe& v) {
  *reinterpret_cast<uint4*>(g) = make_uint4(v.w[0], v.w[1], v.w[2], v.w[3]);
  *reinterpret_cast<uint4*>(g + 4) = make_uint4(v.w[4], v.w[5], v.w[6], v.w[7]);
}

// secp256k1: scalar field secp256k1.rs:1953-1969, 2162-2195, 2270-2297, 2410-2456 (its N has the two top
// limbs swapped and its Mul keeps only the low 256 bits of the product); ct_lt overridden with a true
// comparison against that N
struct ESecp {
  typedef secp::pt pt;
  FEC_DEV static unsigned char scalars(const fe& h, const fe& r, const fe& s, fe& u1, fe& u2) {
    const bool bad = lane_of(fe_is_zero(r) | fe_is_zero(s) | secp::sc_ge_n(r) | secp::sc_ge_n(s));  // 215-228
    const bool panic = lane_of(secp::sc_ge_n(h));                                                   // 239
    const fe s_inv = secp::sc_inv(s);
    u1 = secp::sc_mul(h, s_inv);                                                                    // 250-251
    u2 = secp::sc_mul(r, s_inv);
    return bad ? F_FALSE : (panic ? F_PANIC : F_GO);
  }
  // identity -> false (259-262); x of to_affine (264) through FieldElement::to_bytes (138-178, a Montgomery
  // reduction) read as a scalar: >= n panics (271 unwrap), else compared with `target` (274)
  FEC_DEV static unsigned char compare_x(const pt& rp, const fe& target) {
    const bool ident = lane_of(secp::is_identity(rp));
    fe x, y;
    secp::to_affine(rp, x, y);
    const fe xr = secp::mul(x, fe_small(1));
    if (ident) return 0;
    if (lane_of(secp::sc_ge_n(xr))) return 2;
    return lane_of(fe_eq(xr, target)) ? 1 : 0;
  }
  FEC_DEV static unsigned char finish(const pt& a, const pt& b, const fe& r) { return compare_x(secp::padd(a, b), r); }  // 256
  // KeyExchange for Secp256k1 (secp256k1.rs:1884-1904) does not validate the public key
  FEC_DEV static bool pk_valid(const fe&, const fe&, bool) { return true; }
  FEC_DEV static fe x_value(const fe& x) { return secp::mul(x, fe_small(1)); }      // FieldElement::to_bytes (138-178)
  FEC_DEV static lmask to_affine(const pt& p, fe& x, fe& y) { return secp::to_affine(p, x, y); }
  FEC_DEV static fe wmul(const fe& a, const fe& b) { return secp::sc_mul(a, b); }   // impl Mul for Scalar
  FEC_DEV static fe wadd(const fe& a, const fe& b) { return secp::sc_add(a, b); }   // impl Add for Scalar
  static void launch_mul(const SchedEnv& env, bool fixed, const u32* k, const u32* p, u32* o, size_t n, hipStream_t s, unsigned = 1) {
    secp_launch_mul(env, fixed, k, p, o, n, s);
  }
};

// P-256: scalar field p256.rs:875-1100, 1409-1432 (reduce_wide drops the high half of its second fold);
// ct_lt is the trait default (forge-ec-core/src/lib.rs:497-531), a top-byte <= comparison
struct EP256 {
  typedef p256::pt pt;
  FEC_DEV static unsigned char scalars(const fe& h, const fe& r, const fe& s, fe& u1, fe& u2) {
    const p256::sc hs = p256::sc_of(h), rs = p256::sc_of(r), ss = p256::sc_of(s);
    const p256::sc order = {{0xF3B9CAC2FC632551ULL, 0xBCE6FAADA7179E84ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFF00000000ULL}};
    const bool bad = p256::sc_is_zero(rs) || p256::sc_is_zero(ss) ||
                     !(p256::sc_ct_lt_default(rs, order) && p256::sc_ct_lt_default(ss, order));     // 215-228
    const bool panic = p256::sc_ge_n(hs);                                                           // 239
    const fe s_inv = p256::sc_inv32(s);        // s != 0 on every lane that is not `bad`
    u1 = p256::sc_mul32(h, s_inv);                                                                  // 250-251
    u2 = p256::sc_mul32(r, s_inv);
    return bad ? F_FALSE : (panic ? F_PANIC : F_GO);
  }
  // identity -> false (259-262); x of to_affine (264); field_to_bytes = FieldElement::to_bytes (288-300): the raw
  // limbs; Scalar::from_bytes: valid iff < n, else the unwrap at 271 panics; compared with `target` (274)
  FEC_DEV static unsigned char compare_x(const pt& rp, const fe& target) {
    const bool ident = lane_of(p256::is_identity(rp));
    fe x, y;
    p256::to_affine(rp, x, y);
    if (ident) return 0;
    if (p256::sc_ge_n(p256::sc_of(x))) return 2;
    return lane_of(fe_eq(x, target)) ? 1 : 0;
  }
  FEC_DEV static unsigned char finish(const pt& a, const pt& b, const fe& r) { return compare_x(p256::padd(a, b), r); }  // 256
  // validate_public_key (p256.rs:2304-2312): !is_identity & validate_point (2187-2191) = is_on_curve (1636-1656)
  FEC_DEV static bool pk_valid(const fe& x, const fe& y, bool inf) {
    return !inf && lane_of(fe_eq(p256::sqr(y), p256::curve_rhs(x)));
  }
  FEC_DEV static fe x_value(const fe& x) { return x; }                               // FieldElement::to_bytes (288-300)
  FEC_DEV static lmask to_affine(const pt& p, fe& x, fe& y) { return p256::to_affine(p, x, y); }
  FEC_DEV static fe wmul(const fe& a, const fe& b) { return p256::sc_mul32(a, b); }
  FEC_DEV static fe wadd(const fe& a, const fe& b) { return p256::sc_fe(p256::sc_add(p256::sc_of(a), p256::sc_of(b))); }
  // cu_divisor == 2: this launch runs beside its fixed- / variable-base twin on another stream (cu_split.hpp)
  static void launch_mul(const SchedEnv& env, bool fixed, const u32* k, const u32* p, u32* o, size_t n, hipStream_t s,
                         unsigned cu_divisor = 1) {
    SchedEnv ef = env, ev = env;
    if (cu_divisor == 2) p256_cu_split(env, n, kP256VarAffineMs, ef, ev);   // (u2 * from_affine(public key): affine addend)
    p256_launch_mul(fixed ? ef : ev, fixed, k, p, o, n, s);
  }
};

template <class E>
__global__ __launch_bounds__(TPB) void k_ecdsa_pre(const unsigned char* __restrict__ digests, const u32* __restrict__ rs,
                                                   const u32* __restrict__ ss, const u32* __restrict__ pk,
                                                   const unsigned char* __restrict__ pk_inf,
                                                   const u32* __restrict__ weights, u32* __restrict__ u1,
                                                   u32* __restrict__ u2, u32* __restrict__ q,
                                                   unsigned char* __restrict__ flags, u32* __restrict__ ar, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  // Scalar::from_bytes: big-endian bytes -> little-endian limbs
  const fe d = load8(reinterpret_cast<const u32*>(digests + i * 32));
  fe h;
  FEC_UNROLL for (int w = 0; w < 8; ++w) h.w[w] = __builtin_bswap32(d.w[7 - w]);
  const fe r = load8(rs + i * 8), s = load8(ss + i * 8);
  fe a, b;
  flags[i] = E::scalars(h, r, s, a, b);
  if (weights != nullptr) {  // batch_verify (ecdsa.rs:349-350, 370): a_i * u1, a_i * u2, a_i * r_i
    const fe w = load8(weights + i * 8);
    a = E::wmul(w, a);
    b = E::wmul(w, b);
    store8(ar + i * 8, E::wmul(w, r));
  }
  store8(u1 + i * 8, a);
  store8(u2 + i * 8, b);
  // from_affine (secp256k1.rs:1365-1373, p256.rs:1859-1867): (x, y, 1), or the identity (0, 1, 0)
  const bool inf = pk_inf != nullptr && pk_inf[i] != 0;
  fe x = load8(pk + i * 16), y = load8(pk + i * 16 + 8), z = fe_small(1);
  if (inf) { x = fe_zero(); y = fe_small(1); z = fe_zero(); }
  store8(q + i * 24, x);
  store8(q + i * 24 + 8, y);
  store8(q + i * 24 + 16, z);
}

template <class E>
__global__ __launch_bounds__(TPB) void k_ecdsa_finish(const u32* __restrict__ ta, const u32* __restrict__ tb,
                                                      const u32* __restrict__ rs, const unsigned char* __restrict__ flags,
                                                      unsigned char* __restrict__ status, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  typename E::pt a, b;
  a.x = load8(ta + i * 24); a.y = load8(ta + i * 24 + 8); a.z = load8(ta + i * 24 + 16);
  b.x = load8(tb + i * 24); b.y = load8(tb + i * 24 + 8); b.z = load8(tb + i * 24 + 16);
  const unsigned char st = E::finish(a, b, load8(rs + i * 8));
  const unsigned char f = flags[i];
  status[i] = f == F_FALSE ? 0 : (f == F_PANIC ? 2 : st);
}

// The fixed-base multiplication of a pair (multiply(G, u1), multiply(Q, u2)) forked onto a second stream and joined back
// with events (no host blocking): the two launches overlap, each persistent kernel on half of the CUs.  Wins at every
// batch size (fecgpu.hip: SideStream).  Inactive -- `s` is the main stream -- without a distinct second stream.
struct Fork {
  hipStream_t main, s;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  bool active = false;
  Fork(hipStream_t main_, hipStream_t side) : main(main_), s(main_) {
    if (side == nullptr || side == main_) return;
    if (hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ev_out, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    (void)hipEventRecord(ev_in, main);
    (void)hipStreamWaitEvent(side, ev_in, 0);
    s = side;
    active = true;
  }
  void join() {
    if (!active) return;
    (void)hipEventRecord(ev_out, s);
    (void)hipStreamWaitEvent(main, ev_out, 0);
  }
  ~Fork() {
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_out) (void)hipEventDestroy(ev_out);
  }
};

template <class E>
void run(const SchedEnv& env, const unsigned char* dd, const u32* dr, const u32* ds, const u32* dpk, const unsigned char* dinf, const u32* gen,
         unsigned char* dstatus, void* work, size_t n, hipStream_t s, hipStream_t side) {
  char* w = static_cast<char*>(work);
  u32* u1 = reinterpret_cast<u32*>(w);
  u32* u2 = reinterpret_cast<u32*>(w + n * 32);
  u32* q = reinterpret_cast<u32*>(w + n * 64);
  u32* ta = reinterpret_cast<u32*>(w + n * 160);
  u32* tb = reinterpret_cast<u32*>(w + n * 256);
  unsigned char* flags = reinterpret_cast<unsigned char*>(w + n * 352);
  const dim3 g((unsigned)((n + TPB - 1) / TPB)), b(TPB);
  hipLaunchKernelGGL((k_ecdsa_pre<E>), g, b, 0, s, dd, dr, ds, dpk, dinf, (const u32*)nullptr, u1, u2, q, flags,
                     (u32*)nullptr, n);
  {
    Fork fork(s, side);
    E::launch_mul(env, true, u1, gen, ta, n, fork.s, fork.active ? 2 : 1);
    E::launch_mul(env, false, u2, q, tb, n, s, fork.active ? 2 : 1);
    fork.join();
  }
  hipLaunchKernelGGL((k_ecdsa_finish<E>), g, b, 0, s, (const u32*)ta, (const u32*)tb, dr, (const unsigned char*)flags, dstatus, n);
}

// ---- Ecdsa::<C, D>::batch_verify (ecdsa.rs:287-391), the part after the ordered point fold ----
// One wavefront: r_scalar_sum = sum of a_i * r_i in index order (368-372; the reference's scalar Add is not a
// group law, so the order is part of the result), then 361-384 on r_sum.  result: 1 true, 0 false, 2 panics.
// detail: 24 words r_sum, 8 words r_scalar_sum.
template <class E>
__global__ __launch_bounds__(64) void k_ecdsa_batch_finish(const u32* __restrict__ r_sum, const u32* __restrict__ ar,
                                                          size_t n, unsigned char* __restrict__ result,
                                                          u32* __restrict__ detail) {
  __shared__ u32 sh[64 * 8];
  const int lane = threadIdx.x;
  fe total = fe_zero();
  for (size_t base = 0; base < n; base += 64) {
    const int cnt = (n - base) < 64 ? (int)(n - base) : 64;
    if (lane < cnt) {
      const fe t = load8(ar + (base + lane) * 8);
      FEC_UNROLL for (int w = 0; w < 8; ++w) sh[lane * 8 + w] = t.w[w];
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll 1
      for (int j = 0; j < cnt; ++j) {
        fe t;
        FEC_UNROLL for (int w = 0; w < 8; ++w) t.w[w] = sh[j * 8 + w];
        total = E::wadd(total, t);
      }
    }
    __syncthreads();
  }
  if (lane != 0) return;
  typename E::pt p;
  p.x = load8(r_sum); p.y = load8(r_sum + 8); p.z = load8(r_sum + 16);
  result[0] = E::compare_x(p, total);
  store8(detail, p.x); store8(detail + 8, p.y); store8(detail + 16, p.z);
  store8(detail + 24, total);
}

// ---- KeyExchange::derive_shared_secret (secp256k1.rs:1884-1904, p256.rs:2281-2302) ----
// pre: public-key validation (P-256 only) and from_affine; the multiplication is the curve's own kernel;
// finish: to_affine, identity -> Err, else x.to_bytes().  status: 0 Ok, 1 Err(InvalidPublicKey), 2 Err (identity).
template <class E>
__global__ __launch_bounds__(TPB) void k_ecdh_pre(const u32* __restrict__ pk, const unsigned char* __restrict__ pk_inf,
                                                  u32* __restrict__ q, unsigned char* __restrict__ flags, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const bool inf = pk_inf != nullptr && pk_inf[i] != 0;
  fe x = load8(pk + i * 16), y = load8(pk + i * 16 + 8), z = fe_small(1);
  flags[i] = E::pk_valid(x, y, inf) ? 0 : 1;
  if (inf) { x = fe_zero(); y = fe_small(1); z = fe_zero(); }   // from_affine of the identity: (0, 1, 0)
  store8(q + i * 24, x);
  store8(q + i * 24 + 8, y);
  store8(q + i * 24 + 16, z);
}
template <class E>
__global__ __launch_bounds__(TPB) void k_ecdh_finish(const u32* __restrict__ t, const unsigned char* __restrict__ flags,
                                                     u32* __restrict__ out, unsigned char* __restrict__ status, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  typename E::pt p;
  p.x = load8(t + i * 24); p.y = load8(t + i * 24 + 8); p.z = load8(t + i * 24 + 16);
  fe x, y;
  const bool ident = lane_of(E::to_affine(p, x, y));
  const fe v = E::x_value(x);
  const unsigned char st = flags[i] != 0 ? 1 : (ident ? 2 : 0);
  fe o;  // big-endian bytes of the value, as eight words in memory order; zero unless Ok
  FEC_UNROLL for (int w = 0; w < 8; ++w) o.w[w] = st == 0 ? __builtin_bswap32(v.w[7 - w]) : 0u;
  store8(out + i * 8, o);
  status[i] = st;
}
template <class E>
void run_ecdh(const SchedEnv& env, const u32* sk, const u32* pk, const unsigned char* pk_inf, u32* out, unsigned char* status, void* work, size_t n,
              hipStream_t s) {
  char* w = static_cast<char*>(work);
  u32* q = reinterpret_cast<u32*>(w);
  u32* t = reinterpret_cast<u32*>(w + n * 96);
  unsigned char* flags = reinterpret_cast<unsigned char*>(w + n * 192);
  const dim3 g((unsigned)((n + TPB - 1) / TPB)), b(TPB);
  hipLaunchKernelGGL((k_ecdh_pre<E>), g, b, 0, s, pk, pk_inf, q, flags, n);
  E::launch_mul(env, false, sk, q, t, n, s);
  hipLaunchKernelGGL((k_ecdh_finish<E>), g, b, 0, s, (const u32*)t, (const unsigned char*)flags, out, status, n);
}

// ---- Curve::validate_point ----
// secp256k1 (secp256k1.rs:2722-2726) and P-256 (p256.rs:2187-2191): PointAffine::is_on_curve, the infinity flag
// counting as on the curve -- one pass.
template <int CURVE>
__global__ __launch_bounds__(TPB) void k_validate_weierstrass(const u32* __restrict__ xy, const unsigned char* __restrict__ inf,
                                                              unsigned char* __restrict__ ok, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const fe x = load8(xy + i * 16), y = load8(xy + i * 16 + 8);
  const bool on = CURVE == FEC_SECP256K1 ? lane_of(secp::affine_on_curve(x, y)) : lane_of(fe_eq(p256::sqr(y), p256::curve_rhs(x)));
  ok[i] = ((inf != nullptr && inf[i] != 0) || on) ? 1 : 0;
}
// Ed25519 keeps the trait default (forge-ec-core/src/lib.rs:905-925): is_on_curve (ed25519.rs:1719-1744) AND
// multiply(multiply(from_affine(p), 8), L).is_identity() -- two runs of the variable-base kernel with constant
// scalars between a pre pass (from_affine, on-curve flag) and a finishing pass.
__global__ __launch_bounds__(TPB) void k_fill_scalar(u32* __restrict__ dst, uint4 lo, uint4 hi, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  *reinterpret_cast<uint4*>(dst + i * 8) = lo;
  *reinterpret_cast<uint4*>(dst + i * 8 + 4) = hi;
}
__global__ __launch_bounds__(TPB) void k_ed_validate_pre(const u32* __restrict__ xy, const unsigned char* __restrict__ inf,
                                                         u32* __restrict__ a, unsigned char* __restrict__ flags, size_t n);
__global__ __launch_bounds__(TPB) void k_ed_validate_finish(const u32* __restrict__ t, const unsigned char* __restrict__ flags,
                                                            unsigned char* __restrict__ ok, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  ed::pt p;
  p.x = load8(t + i * 32); p.y = load8(t + i * 32 + 8); p.z = load8(t + i * 32 + 16); p.t = load8(t + i * 32 + 24);
  ok[i] = (flags[i] != 0 && lane_of(ed::is_identity(p))) ? 1 : 0;
}

// ---- Eddsa::<Ed25519, D>::verify / Ed25519::verify from the point computation on (eddsa.rs:174-211, 430-447) ----
FEC_DEV ed::pt ed_from_affine(const fe& x, const fe& y, bool inf) {  // ed25519.rs:1813-1826
  ed::pt p;
  p.x = x; p.y = y; p.z = fe_small(1); p.t = ed::mul(x, y);
  return ed::pt_select(p, ed::identity(), lanes_where(inf));
}
FEC_DEV ed::pt ed_load32(const u32* g) {
  ed::pt p;
  p.x = load8(g); p.y = load8(g + 8); p.z = load8(g + 16); p.t = load8(g + 24);
  return p;
}

// A = from_affine(pk) as the base of multiply(A, k)
__global__ __launch_bounds__(TPB) void k_eddsa_pre(const u32* __restrict__ pk, const unsigned char* __restrict__ pk_inf,
                                                   u32* __restrict__ a, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const bool inf = pk_inf != nullptr && pk_inf[i] != 0;
  const ed::pt p = ed_from_affine(load8(pk + i * 16), load8(pk + i * 16 + 8), inf);
  store8(a + i * 32, p.x); store8(a + i * 32 + 8, p.y); store8(a + i * 32 + 16, p.z); store8(a + i * 32 + 24, p.t);
}

// PointAffine::is_on_curve (ed25519.rs:1719-1744) / PointAffine::new (1476-1498): -x^2 + y^2 == 1 + d x^2 y^2
FEC_DEV lmask ed_affine_on_curve(const fe& x, const fe& y) {
  const fe x2 = ed::mul(x, x), y2 = ed::mul(y, y);
  const fe lhs = ed::add(ed::neg(x2), y2);
  const fe rhs = ed::add(fe_small(1), ed::mul(ed::D_(), ed::mul(x2, y2)));
  return fe_eq(lhs, rhs);
}
__global__ __launch_bounds__(TPB) void k_ed_validate_pre(const u32* __restrict__ xy, const unsigned char* __restrict__ inf,
                                                         u32* __restrict__ a, unsigned char* __restrict__ flags, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const bool is_inf = inf != nullptr && inf[i] != 0;
  const fe x = load8(xy + i * 16), y = load8(xy + i * 16 + 8);
  flags[i] = (is_inf || lane_of(ed_affine_on_curve(x, y))) ? 1 : 0;
  const ed::pt p = ed_from_affine(x, y, is_inf);
  store8(a + i * 32, p.x); store8(a + i * 32 + 8, p.y); store8(a + i * 32 + 16, p.z); store8(a + i * 32 + 24, p.t);
}

// sg = multiply(G, s), ka = multiply(A, k): R + ka, both to_affine, from_affine(..) - from_affine(..), is_identity
__global__ __launch_bounds__(TPB) void k_eddsa_finish(const u32* __restrict__ sg, const u32* __restrict__ ka,
                                                      const u32* __restrict__ r_xy, const unsigned char* __restrict__ r_inf,
                                                      unsigned char* __restrict__ status, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const ed::pt s_g = ed_load32(sg + i * 32), k_a = ed_load32(ka + i * 32);
  const ed::pt r = ed_from_affine(load8(r_xy + i * 16), load8(r_xy + i * 16 + 8), false);
  const ed::pt rk = ed::padd(r, k_a);                                                     // 200 / 435
  // to_affine (1793-1811) unwraps z.invert(): a zero z of a point that is not the identity panics
  const bool panic = lane_of((~ed::is_identity(s_g) & fe_is_zero(s_g.z)) | (~ed::is_identity(rk) & fe_is_zero(rk.z)));
  fe x1, y1, x2, y2;
  const lmask i1 = ed::to_affine(s_g, x1, y1), i2 = ed::to_affine(rk, x2, y2);           // 204-205 / 439-440
  const ed::pt p1 = ed_from_affine(x1, y1, lane_of(i1));
  ed::pt p2 = ed_from_affine(x2, y2, lane_of(i2));
  p2.x = ed::neg(p2.x);                                                                    // negate 1834-1841
  p2.t = ed::neg(p2.t);
  const bool same = lane_of(ed::is_identity(ed::padd(p1, p2)));                           // Sub 1936-1947; 210 / 446
  const bool rinf = r_inf != nullptr && r_inf[i] != 0;                                     // 174-177
  status[i] = rinf ? 0 : (panic ? 2 : (same ? 1 : 0));
}

// ---- Schnorr::<C, D>::verify per signature (forge-ec-signature/src/schnorr.rs:90-140), from the point computation on ----
//   103-105  an infinite signature point is rejected            125-126  s_g = multiply(G, s), e_p = multiply(from_affine(pk), e)
//   129-134  to_affine(e_p), PointAffine::new(x, -y): the curve equation re-validated under the reference's own arithmetic
//            (secp256k1.rs:856-869, p256.rs:1535-1552, ed25519.rs:1477-1498); None is `return false`
//   136-139  r' = s_g + from_affine(neg), to_affine             142  AffinePoint::ct_eq: (x == x & y == y) | (inf & inf)
struct VSecp {
  typedef secp::pt pt;
  static constexpr int PW = 24;
  FEC_DEV static pt from_affine(const fe& x, const fe& y, bool inf) {
    pt p; p.x = x; p.y = y; p.z = fe_small(1);
    return inf ? secp::identity() : p;
  }
  FEC_DEV static pt load(const u32* g) { pt p; p.x = load8(g); p.y = load8(g + 8); p.z = load8(g + 16); return p; }
  FEC_DEV static void store(u32* g, const pt& p) { store8(g, p.x); store8(g + 8, p.y); store8(g + 16, p.z); }
  FEC_DEV static fe neg(const fe& a) { return secp::neg(a); }
  FEC_DEV static bool on_curve(const fe& x, const fe& y) { return lane_of(secp::affine_on_curve(x, y)); }
  FEC_DEV static pt padd(const pt& a, const pt& b) { return secp::padd(a, b); }
  FEC_DEV static lmask to_affine(const pt& p, fe& x, fe& y) { return secp::to_affine(p, x, y); }
  FEC_DEV static bool panics(const pt&) { return false; }      // z == 0 is the identity (1331-1336): to_affine never unwraps None
};
struct VP256 {
  typedef p256::pt pt;
  static constexpr int PW = 24;
  FEC_DEV static pt from_affine(const fe& x, const fe& y, bool inf) {
    pt p; p.x = x; p.y = y; p.z = fe_small(1);
    return inf ? p256::identity() : p;
  }
  FEC_DEV static pt load(const u32* g) { pt p; p.x = load8(g); p.y = load8(g + 8); p.z = load8(g + 16); return p; }
  FEC_DEV static void store(u32* g, const pt& p) { store8(g, p.x); store8(g + 8, p.y); store8(g + 16, p.z); }
  FEC_DEV static fe neg(const fe& a) { return p256::neg(a); }
  FEC_DEV static bool on_curve(const fe& x, const fe& y) { return lane_of(fe_eq(p256::sqr(y), p256::curve_rhs(x))); }
  FEC_DEV static pt padd(const pt& a, const pt& b) { return p256::padd(a, b); }
  FEC_DEV static lmask to_affine(const pt& p, fe& x, fe& y) { return p256::to_affine(p, x, y); }
  FEC_DEV static bool panics(const pt&) { return false; }
};
struct VEd {
  typedef ed::pt pt;
  static constexpr int PW = 32;
  FEC_DEV static pt from_affine(const fe& x, const fe& y, bool inf) { return ed_from_affine(x, y, inf); }
  FEC_DEV static pt load(const u32* g) { return ed_load32(g); }
  FEC_DEV static void store(u32* g, const pt& p) { store8(g, p.x); store8(g + 8, p.y); store8(g + 16, p.z); store8(g + 24, p.t); }
  FEC_DEV static fe neg(const fe& a) { return ed::neg(a); }
  FEC_DEV static bool on_curve(const fe& x, const fe& y) { return lane_of(ed_affine_on_curve(x, y)); }
  FEC_DEV static pt padd(const pt& a, const pt& b) { return ed::padd(a, b); }
  FEC_DEV static lmask to_affine(const pt& p, fe& x, fe& y) { return ed::to_affine(p, x, y); }
  // to_affine (1793-1811) unwraps z.invert(): a zero z of a point that is not the identity panics
  FEC_DEV static bool panics(const pt& p) { return lane_of(~ed::is_identity(p) & fe_is_zero(p.z)); }
};

template <class V>
__global__ __launch_bounds__(TPB) void k_schnorr_verify_pre(const u32* __restrict__ pk, const unsigned char* __restrict__ pk_inf,
                                                            u32* __restrict__ a, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const bool inf = pk_inf != nullptr && pk_inf[i] != 0;
  V::store(a + i * V::PW, V::from_affine(load8(pk + i * 16), load8(pk + i * 16 + 8), inf));
}
template <class V>
__global__ __launch_bounds__(TPB) void k_schnorr_verify_finish(const u32* __restrict__ sg, const u32* __restrict__ ep,
                                                               const u32* __restrict__ r_xy, const unsigned char* __restrict__ r_inf,
                                                               unsigned char* __restrict__ status, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const bool rinf = r_inf != nullptr && r_inf[i] != 0;                                     // 103-105
  const typename V::pt s_g = V::load(sg + i * V::PW), e_p = V::load(ep + i * V::PW);
  bool panic = V::panics(e_p);
  fe x, y;
  (void)V::to_affine(e_p, x, y);                                                           // 129: (0, 0) for the identity
  const fe ny = V::neg(y);                                                                 // 130
  const bool some = V::on_curve(x, ny);                                                    // 130-134
  const typename V::pt rp = V::padd(s_g, V::from_affine(x, ny, false));                    // 136-138
  panic = panic || (some && V::panics(rp));
  fe rx, ry;
  const bool ri = lane_of(V::to_affine(rp, rx, ry));                                       // 139
  const bool same = lane_of(fe_eq(rx, load8(r_xy + i * 16)) & fe_eq(ry, load8(r_xy + i * 16 + 8)));
  (void)ri;                                                                                // (inf & inf): sig.r is finite here
  status[i] = rinf ? 0 : (V::panics(e_p) ? 2 : (!some ? 0 : (panic ? 2 : (same ? 1 : 0))));
}

}  // namespace

size_t schnorr_verify_work_bytes(int curve, size_t n) { return n * 3 * (curve == FEC_ED25519 ? 128 : 96); }
void schnorr_verify_pre_launch(int curve, const u32* pk, const unsigned char* pk_inf, u32* a, size_t n, hipStream_t s) {
  const dim3 g((unsigned)((n + TPB - 1) / TPB)), b(TPB);
  if (curve == FEC_SECP256K1) hipLaunchKernelGGL((k_schnorr_verify_pre<VSecp>), g, b, 0, s, pk, pk_inf, a, n);
  else if (curve == FEC_P256) hipLaunchKernelGGL((k_schnorr_verify_pre<VP256>), g, b, 0, s, pk, pk_inf, a, n);
  else hipLaunchKernelGGL((k_schnorr_verify_pre<VEd>), g, b, 0, s, pk, pk_inf, a, n);
}
void schnorr_verify_finish_launch(int curve, const u32* sg, const u32* ep, const u32* r_xy, const unsigned char* r_inf,
                                  unsigned char* status, size_t n, hipStream_t s) {
  const dim3 g((unsigned)((n + TPB - 1) / TPB)), b(TPB);
  if (curve == FEC_SECP256K1) hipLaunchKernelGGL((k_schnorr_verify_finish<VSecp>), g, b, 0, s, sg, ep, r_xy, r_inf, status, n);
  else if (curve == FEC_P256) hipLaunchKernelGGL((k_schnorr_verify_finish<VP256>), g, b, 0, s, sg, ep, r_xy, r_inf, status, n);
  else hipLaunchKernelGGL((k_schnorr_verify_finish<VEd>), g, b, 0, s, sg, ep, r_xy, r_inf, status, n);
}

void eddsa_pre_launch(const u32* pk, const unsigned char* pk_inf, u32* a, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_eddsa_pre, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, s, pk, pk_inf, a, n);
}
void eddsa_finish_launch(const u32* sg, const u32* ka, const u32* r_xy, const unsigned char* r_inf, unsigned char* status,
                         size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_eddsa_finish, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, s, sg, ka, r_xy, r_inf, status, n);
}

size_t ecdsa_work_bytes(size_t n) { return n * 353; }

// Curve::validate_point.  Work (Ed25519 only): point A (128 n), scalars (32 n), products T1, T2 (128 n each), flags (n).
size_t validate_work_bytes(int curve, size_t n) { return curve == FEC_ED25519 ? n * 417 : 0; }
void validate_launch(const SchedEnv& env, int curve, const u32* xy, const unsigned char* inf, unsigned char* ok, void* work, size_t n, hipStream_t s) {
  const dim3 g((unsigned)((n + TPB - 1) / TPB)), b(TPB);
  if (curve == FEC_SECP256K1) {
    hipLaunchKernelGGL((k_validate_weierstrass<FEC_SECP256K1>), g, b, 0, s, xy, inf, ok, n);
  } else if (curve == FEC_P256) {
    hipLaunchKernelGGL((k_validate_weierstrass<FEC_P256>), g, b, 0, s, xy, inf, ok, n);
  } else {
    char* w = static_cast<char*>(work);
    u32* a = reinterpret_cast<u32*>(w);
    u32* k = reinterpret_cast<u32*>(w + n * 128);
    u32* t1 = reinterpret_cast<u32*>(w + n * 160);
    u32* t2 = reinterpret_cast<u32*>(w + n * 288);
    unsigned char* flags = reinterpret_cast<unsigned char*>(w + n * 416);
    hipLaunchKernelGGL(k_ed_validate_pre, g, b, 0, s, xy, inf, a, flags, n);
    hipLaunchKernelGGL(k_fill_scalar, g, b, 0, s, k, make_uint4(8, 0, 0, 0), make_uint4(0, 0, 0, 0), n);          // Scalar::from(8)
    ed_launch_mul(env, k, a, t1, n, s);                                                                            // clear_cofactor
    hipLaunchKernelGGL(k_fill_scalar, g, b, 0, s, k, make_uint4(0x5CF5D3EDu, 0x5812631Au, 0xA2F79CD6u, 0x14DEF9DEu),
                       make_uint4(0, 0, 0, 0x10000000u), n);                                                  // order() = L
    ed_launch_mul(env, k, t1, t2, n, s);
    hipLaunchKernelGGL(k_ed_validate_finish, g, b, 0, s, (const u32*)t2, (const unsigned char*)flags, ok, n);
  }
}

size_t ecdh_work_bytes(size_t n) { return n * 193; }
void ecdh_launch(const SchedEnv& env, int curve, const u32* sk, const u32* pk, const unsigned char* pk_inf, u32* out,
                 unsigned char* status, void* work, size_t n, hipStream_t s) {
  if (curve == FEC_SECP256K1) run_ecdh<ESecp>(env, sk, pk, pk_inf, out, status, work, n, s);
  else run_ecdh<EP256>(env, sk, pk, pk_inf, out, status, work, n, s);
}

// batch_verify, first half: work area as ecdsa_launch plus ar at +n*353 rounded up to 16 (n * 32 bytes).
size_t ecdsa_batch_work_bytes(size_t n) { return ((n * 353 + 15) & ~(size_t)15) + n * 32; }
void ecdsa_batch_pre_launch(int curve, const unsigned char* digests, const u32* r, const u32* s_, const u32* pk,
                            const unsigned char* pk_inf, const u32* weights, void* work, size_t n, hipStream_t s) {
  char* w = static_cast<char*>(work);
  u32* u1 = reinterpret_cast<u32*>(w);
  u32* u2 = reinterpret_cast<u32*>(w + n * 32);
  u32* q = reinterpret_cast<u32*>(w + n * 64);
  unsigned char* flags = reinterpret_cast<unsigned char*>(w + n * 352);
  u32* ar = reinterpret_cast<u32*>(w + ((n * 353 + 15) & ~(size_t)15));
  const dim3 g((unsigned)((n + TPB - 1) / TPB)), b(TPB);
  if (curve == FEC_SECP256K1) hipLaunchKernelGGL((k_ecdsa_pre<ESecp>), g, b, 0, s, digests, r, s_, pk, pk_inf, weights, u1, u2, q, flags, ar, n);
  else hipLaunchKernelGGL((k_ecdsa_pre<EP256>), g, b, 0, s, digests, r, s_, pk, pk_inf, weights, u1, u2, q, flags, ar, n);
}
// second half: ta = multiply(G, a*u1), tb = multiply(Q, a*u2)
// `side` (may be null): a second stream for the fixed-base launch -- at the moderate n batch_verify is meant for, one
// launch fills a fraction of the chip and is bound by the latency of one multiplication, so the two overlap
void ecdsa_batch_mul_launch(const SchedEnv& env, int curve, const u32* gen, void* work, size_t n, hipStream_t s, hipStream_t side) {
  char* w = static_cast<char*>(work);
  const u32* u1 = reinterpret_cast<const u32*>(w);
  const u32* u2 = reinterpret_cast<const u32*>(w + n * 32);
  const u32* q = reinterpret_cast<const u32*>(w + n * 64);
  u32* ta = reinterpret_cast<u32*>(w + n * 160);
  u32* tb = reinterpret_cast<u32*>(w + n * 256);
  Fork fork(s, side);
  const bool two = fork.active;
  if (curve == FEC_SECP256K1) ESecp::launch_mul(env, true, u1, gen, ta, n, fork.s);
  else EP256::launch_mul(env, true, u1, gen, ta, n, fork.s, two ? 2 : 1);
  if (curve == FEC_SECP256K1) ESecp::launch_mul(env, false, u2, q, tb, n, s);
  else EP256::launch_mul(env, false, u2, q, tb, n, s, two ? 2 : 1);
  fork.join();
}
void ecdsa_batch_finish_launch(int curve, const u32* r_sum, const void* work, size_t n, unsigned char* result, u32* detail,
                               hipStream_t s) {
  const u32* ar = reinterpret_cast<const u32*>(static_cast<const char*>(work) + ((n * 353 + 15) & ~(size_t)15));
  if (curve == FEC_SECP256K1) hipLaunchKernelGGL((k_ecdsa_batch_finish<ESecp>), dim3(1), dim3(64), 0, s, r_sum, ar, n, result, detail);
  else hipLaunchKernelGGL((k_ecdsa_batch_finish<EP256>), dim3(1), dim3(64), 0, s, r_sum, ar, n, result, detail);
}

void ecdsa_launch(const SchedEnv& env, int curve, const unsigned char* digests, const u32* r, const u32* s_, const u32* pk,
                  const unsigned char* pk_inf, const u32* gen, unsigned char* status, void* work, size_t n,
                  hipStream_t s, hipStream_t side) {
  if (curve == FEC_SECP256K1) run<ESecp>(env, digests, r, s_, pk, pk_inf, gen, status, work, n, s, side);
  else run<EP256>(env, digests, r, s_, pk, pk_inf, gen, status, work, n, s, side);
}

}  // namespace fecgpu
