// fecgpu.hip -- HIP kernels + the extern "C" ABI of include/fecgpu.h.  gfx950 only.
//
// Mapping: one scalar-mul per lane, 64 per wavefront, 256-thread workgroups.
// The (scalar, point) batch is array-of-structs in HBM; each workgroup pulls its 256 elements
// with fully coalesced 16-byte loads into LDS (transposed to word-major so the per-lane reads
// are bank-conflict-free), every lane then runs the reference's op sequence on 8 x 32-bit
// words in VGPRs, and results go back through LDS as coalesced 16-byte stores.  The path is
// integer-VALU bound (~7*10^5 32-bit multiply-adds per secp256k1 scalar-mul against 224 bytes
// of HBM traffic), so there is no MFMA and no inter-workgroup communication.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/fecgpu.h"
#include "ed25519.hpp"
#include "p256.hpp"
#include "secp256k1.hpp"
#include "host_ctx.hpp"
#include "kernels.hpp"
#include "cu_split.hpp"

namespace fecgpu {


// ------------------------------------------------------------------------------------------
// curve adaptors: a uniform static interface over the three curve headers
// ------------------------------------------------------------------------------------------
struct Secp {
  static constexpr int PW = 24;  // 32-bit words per point
  using pt = secp::pt;
  FEC_DEV static pt load(const u32* l, int stride) {
    pt p;
    FEC_UNROLL for (int i = 0; i < 8; ++i) {
      p.x.w[i] = l[i * stride];
      p.y.w[i] = l[(8 + i) * stride];
      p.z.w[i] = l[(16 + i) * stride];
    }
    return p;
  }
  FEC_DEV static void store(u32* l, int stride, const pt& p) {
    FEC_UNROLL for (int i = 0; i < 8; ++i) {
      l[i * stride] = p.x.w[i];
      l[(8 + i) * stride] = p.y.w[i];
      l[(16 + i) * stride] = p.z.w[i];
    }
  }
  FEC_DEV static pt multiply(const pt& p, const u32* kw) { return secp::multiply(p, kw); }
  FEC_DEV static lmask to_affine(const pt& p, fe& x, fe& y) { return secp::to_affine(p, x, y); }
  FEC_DEV static pt identity() { return secp::identity(); }
  FEC_DEV static pt padd(const pt& a, const pt& b) { return secp::padd(a, b); }
  FEC_DEV static pt pdouble(const pt& a) { return secp::pdouble(a); }
  FEC_DEV static pt pdouble_trait(const pt& a) { return secp::pdouble_trait(a); }
  FEC_DEV static pt pnegate(const pt& a) {
    pt r = a;
    r.y = secp::neg(a.y);
    return r;
  }
  FEC_DEV static fe f_add(const fe& a, const fe& b) { return secp::add(a, b); }
  FEC_DEV static fe f_sub(const fe& a, const fe& b) { return secp::sub(a, b); }
  FEC_DEV static fe f_mul(const fe& a, const fe& b) { return secp::mul(a, b); }
  FEC_DEV static fe f_sqr(const fe& a) { return secp::sqr(a); }
  FEC_DEV static fe f_neg(const fe& a) { return secp::neg(a); }
  FEC_DEV static fe sc_mul(const fe& a, const fe& b) { return secp::sc_mul(a, b); }   // impl Mul for Scalar (2410-2456)
  FEC_DEV static fe sc_mul_flag(const fe& a, const fe& b, bool& overflowed) { overflowed = false; return sc_mul(a, b); }
  FEC_DEV static pt from_affine(const fe& x, const fe& y) { pt p; p.x = x; p.y = y; p.z = fe_small(1); return p; }   // 1365-1373
  // FieldElement::to_bytes (138-178): mont_reduce, i.e. Mul by raw 1; big-endian bytes
  FEC_DEV static fe bytes_value(const fe& a) { return secp::mul(a, fe_small(1)); }
  static constexpr bool BYTES_BIG_ENDIAN = true;
};

struct P256 {
  static constexpr int PW = 24;
  using pt = p256::pt;
  FEC_DEV static pt load(const u32* l, int stride) {
    pt p;
    FEC_UNROLL for (int i = 0; i < 8; ++i) {
      p.x.w[i] = l[i * stride];
      p.y.w[i] = l[(8 + i) * stride];
      p.z.w[i] = l[(16 + i) * stride];
    }
    return p;
  }
  FEC_DEV static void store(u32* l, int stride, const pt& p) {
    FEC_UNROLL for (int i = 0; i < 8; ++i) {
      l[i * stride] = p.x.w[i];
      l[(8 + i) * stride] = p.y.w[i];
      l[(16 + i) * stride] = p.z.w[i];
    }
  }
  FEC_DEV static pt multiply(const pt& p, const u32* kw) { return p256::multiply(p, kw); }
  FEC_DEV static lmask to_affine(const pt& p, fe& x, fe& y) { return p256::to_affine(p, x, y); }
  FEC_DEV static pt identity() { return p256::identity(); }
  FEC_DEV static pt padd(const pt& a, const pt& b) { return p256::padd(a, b); }
  FEC_DEV static pt pdouble(const pt& a) { return p256::pdouble(a); }
  FEC_DEV static pt pdouble_trait(const pt& a) { return p256::pdouble(a); }
  FEC_DEV static pt pnegate(const pt& a) {
    pt r = a;
    r.y = p256::neg(a.y);
    return r;
  }
  FEC_DEV static fe f_add(const fe& a, const fe& b) { return p256::add(a, b); }
  FEC_DEV static fe f_sub(const fe& a, const fe& b) { return p256::sub(a, b); }
  FEC_DEV static fe f_mul(const fe& a, const fe& b) { return p256::mul(a, b); }
  FEC_DEV static fe f_sqr(const fe& a) { return p256::sqr(a); }
  FEC_DEV static fe f_neg(const fe& a) { return p256::neg(a); }
  FEC_DEV static fe sc_mul(const fe& a, const fe& b) { return p256::sc_mul32(a, b); }  // impl Mul for Scalar (p256.rs:1409-1432)
  FEC_DEV static fe sc_mul_flag(const fe& a, const fe& b, bool& overflowed) { overflowed = false; return sc_mul(a, b); }
  FEC_DEV static pt from_affine(const fe& x, const fe& y) { pt p; p.x = x; p.y = y; p.z = fe_small(1); return p; }
  // FieldElement::to_bytes (p256.rs:288-300): the raw limbs; big-endian bytes
  FEC_DEV static fe bytes_value(const fe& a) { return a; }
  static constexpr bool BYTES_BIG_ENDIAN = true;
};

struct Ed {
  static constexpr int PW = 32;
  using pt = ed::pt;
  FEC_DEV static pt load(const u32* l, int stride) {
    pt p;
    FEC_UNROLL for (int i = 0; i < 8; ++i) {
      p.x.w[i] = l[i * stride];
      p.y.w[i] = l[(8 + i) * stride];
      p.z.w[i] = l[(16 + i) * stride];
      p.t.w[i] = l[(24 + i) * stride];
    }
    return p;
  }
  FEC_DEV static void store(u32* l, int stride, const pt& p) {
    FEC_UNROLL for (int i = 0; i < 8; ++i) {
      l[i * stride] = p.x.w[i];
      l[(8 + i) * stride] = p.y.w[i];
      l[(16 + i) * stride] = p.z.w[i];
      l[(24 + i) * stride] = p.t.w[i];
    }
  }
  FEC_DEV static pt multiply(const pt& p, const u32* kw) { return ed::multiply(p, kw); }
  FEC_DEV static lmask to_affine(const pt& p, fe& x, fe& y) { return ed::to_affine(p, x, y); }
  FEC_DEV static pt identity() { return ed::identity(); }
  FEC_DEV static pt padd(const pt& a, const pt& b) { return ed::padd(a, b); }
  FEC_DEV static pt pdouble(const pt& a) { return ed::padd(a, a); }
  FEC_DEV static pt pdouble_trait(const pt& a) { return ed::padd(a, a); }
  FEC_DEV static pt pnegate(const pt& a) {
    pt r = a;
    r.x = ed::neg(a.x);
    r.t = ed::neg(a.t);
    return r;
  }
  FEC_DEV static fe f_add(const fe& a, const fe& b) { return ed::add(a, b); }
  FEC_DEV static fe f_sub(const fe& a, const fe& b) { return ed::sub(a, b); }
  FEC_DEV static fe f_mul(const fe& a, const fe& b) { return ed::mul(a, b); }
  FEC_DEV static fe f_sqr(const fe& a) { return ed::mul(a, a); }
  FEC_DEV static fe f_neg(const fe& a) { return ed::neg(a); }
  // impl Mul for Scalar (ed25519.rs:1256-1376) under the release profile: u128 sums wrap, `overflowed` says that one did
  FEC_DEV static fe sc_mul_flag(const fe& a, const fe& b, bool& overflowed) {
    ed::sc4 x, y;
    FEC_UNROLL for (int i = 0; i < 4; ++i) {
      x.l[i] = (u64)a.w[2 * i] | ((u64)a.w[2 * i + 1] << 32);
      y.l[i] = (u64)b.w[2 * i] | ((u64)b.w[2 * i + 1] << 32);
    }
    const ed::sc4 r = ed::sc_mul_release(x, y, overflowed);
    fe o;
    FEC_UNROLL for (int i = 0; i < 4; ++i) {
      o.w[2 * i] = (u32)r.l[i];
      o.w[2 * i + 1] = (u32)(r.l[i] >> 32);
    }
    return o;
  }
  FEC_DEV static pt from_affine(const fe& x, const fe& y) {   // ed25519.rs:1813-1826: z = one(), t = x * y
    pt p;
    p.x = x;
    p.y = y;
    p.z = fe_small(1);
    p.t = ed::mul(x, y);
    return p;
  }
  // FieldElement::to_bytes (ed25519.rs:295-310): reduce(); LITTLE-endian bytes
  FEC_DEV static fe bytes_value(const fe& a) { return ed::reduce(a); }
  static constexpr bool BYTES_BIG_ENDIAN = false;
};


// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
// Curve::multiply lives in kernels_secp.hip (three-waves-per-SIMD ladder), kernels_p256.hip and kernels_ed.hip
// (persistent task schedulers, LDS addend-table kernel); u1*G + u2*Q is composed from them (launch_double_mul).

template <class C>
__global__ __launch_bounds__(TPB) void k_field_op(int op, const u32* __restrict__ a,
                                                  const u32* __restrict__ b, u32* __restrict__ out,
                                                  size_t n) {
  __shared__ u32 lds_a[8 * TPB];
  __shared__ u32 lds_b[8 * TPB];
  const int valid = block_valid(n);
  const size_t first = (size_t)blockIdx.x * TPB;
  stage_in<8>(lds_a, a + first * 8, valid);
  if (b) stage_in<8>(lds_b, b + first * 8, valid);
  __syncthreads();
  const int e = threadIdx.x;
  if (e < valid) {
    fe x = load_fe(lds_a + e, TPB);
    fe y = b ? load_fe(lds_b + e, TPB) : fe_zero();
    fe r;
    switch (op) {
      case FEC_F_ADD: r = C::f_add(x, y); break;
      case FEC_F_SUB: r = C::f_sub(x, y); break;
      case FEC_F_MUL: r = C::f_mul(x, y); break;
      case FEC_F_SQR: r = C::f_sqr(x); break;
      default: r = C::f_neg(x); break;
    }
    store_fe(lds_a + e, TPB, r);
  }
  __syncthreads();
  stage_out<8>(out + first * 8, lds_a, valid);
}

template <class C>
__global__ __launch_bounds__(TPB) void k_point_op(int op, const u32* __restrict__ p,
                                                  const u32* __restrict__ q, u32* __restrict__ out,
                                                  size_t n) {
  __shared__ u32 lds_p[C::PW * TPB];
  __shared__ u32 lds_q[C::PW * TPB];
  const int valid = block_valid(n);
  const size_t first = (size_t)blockIdx.x * TPB;
  stage_in<C::PW>(lds_p, p + first * C::PW, valid);
  if (q) stage_in<C::PW>(lds_q, q + first * C::PW, valid);
  __syncthreads();
  const int e = threadIdx.x;
  if (e < valid) {
    typename C::pt a = C::load(lds_p + e, TPB);
    typename C::pt r;
    switch (op) {
      case FEC_P_ADD: {
        typename C::pt b = C::load(lds_q + e, TPB);
        r = C::padd(a, b);
        break;
      }
      case FEC_P_DOUBLE: r = C::pdouble(a); break;
      case FEC_P_NEGATE: r = C::pnegate(a); break;
      default: r = C::pdouble_trait(a); break;
    }
    C::store(lds_p + e, TPB, r);
  }
  __syncthreads();
  stage_out<C::PW>(out + first * C::PW, lds_p, valid);
}

// Curve::multi_scalar_multiply's fold (forge-ec-core/src/lib.rs:944-948, p256.rs:2204-2208):
//   result = identity; for i in 0..n { result += product[i] }
// The reference's Add is neither associative nor commutative, so the order is part of the result:
// the n products (already computed by the batch kernel) are folded strictly left to right.
// Each addition is spread over a few lanes of the wavefront (secp::padd_coop: four lanes, 6 instead of 16
// dependent field operations; p256::padd_coop: five lanes, 5 instead of 16; ed::padd_coop: four lanes, 3
// instead of 9) -- the same products on the same operands, so the sums are bit-identical.
// One wavefront folds `terms` left to right from the identity with the cooperative addition.
FEC_DEV secp::pt fold_coop_secp(const u32* __restrict__ terms, size_t n, u32* sh) {
  using namespace secp::coop;
  const int lane = threadIdx.x & 63;
  secp::pt acc = secp::identity();
  if (lane == 0) st(sh, ONE, fe_small(1));
  // term i + 1 is fetched (one word per lane) while addition i runs, so its HBM/L2 latency is hidden
  u32 next_word = (n != 0 && lane < 24) ? terms[lane] : 0u;
#pragma unroll 1
  for (size_t i = 0; i < n; ++i) {
    if (lane == 0) {
      st(sh, PX, acc.x);
      st(sh, PY, acc.y);
      st(sh, PZ, acc.z);
    }
    if (lane < 24) sh[QX * 8 + lane] = next_word;  // q = term i (X, Y, Z: slots QX..QZ are contiguous)
    if (i + 1 < n && lane < 24) next_word = terms[(i + 1) * Secp::PW + lane];
    sync();
    acc = secp::padd_coop(sh);
  }
  return acc;
}

// P-256: the same fold with p256::padd_coop (five lanes, 5 instead of 16 dependent field operations)
FEC_DEV p256::pt fold_coop_p256(const u32* __restrict__ terms, size_t n, u32* sh) {
  using namespace p256::coop;
  const int lane = threadIdx.x & 63;
  p256::pt acc = p256::identity();
  if (lane == 0) coopx::st(sh, ONE, fe_small(1));
  u32 next_word = (n != 0 && lane < 24) ? terms[lane] : 0u;
#pragma unroll 1
  for (size_t i = 0; i < n; ++i) {
    if (lane == 0) {
      coopx::st(sh, PX, acc.x);
      coopx::st(sh, PY, acc.y);
      coopx::st(sh, PZ, acc.z);
    }
    if (lane < 24) sh[QX * 8 + lane] = next_word;  // q = term i (X, Y, Z: slots QX..QZ are contiguous)
    if (i + 1 < n && lane < 24) next_word = terms[(i + 1) * P256::PW + lane];
    coopx::sync();
    acc = p256::padd_coop(sh);
  }
  return acc;
}
// Ed25519: ed::padd_coop (four lanes, 3 instead of 9 dependent field operations)
FEC_DEV ed::pt fold_coop_ed(const u32* __restrict__ terms, size_t n, u32* sh) {
  using namespace ed::coop;
  const int lane = threadIdx.x & 63;
  ed::pt acc = ed::identity();
  if (lane == 0) {
    coopx::st(sh, ONE, fe_small(1));
    coopx::st(sh, DCONST, ed::D_());
  }
  u32 next_word = (n != 0 && lane < 32) ? terms[lane] : 0u;
#pragma unroll 1
  for (size_t i = 0; i < n; ++i) {
    if (lane == 0) {
      coopx::st(sh, PX, acc.x);
      coopx::st(sh, PY, acc.y);
      coopx::st(sh, PZ, acc.z);
      coopx::st(sh, PT, acc.t);
    }
    if (lane < 32) sh[QX * 8 + lane] = next_word;  // q = term i (X, Y, Z, T: slots QX..QT are contiguous)
    if (i + 1 < n && lane < 32) next_word = terms[(i + 1) * Ed::PW + lane];
    coopx::sync();
    acc = ed::padd_coop(sh);
  }
  return acc;
}

template <class C>
__global__ __launch_bounds__(64) void k_fold_sum(const u32* __restrict__ products, u32* __restrict__ out, size_t n) {
  if (blockIdx.x != 0) return;
  if constexpr (__is_same(typename C::pt, secp::pt)) {
    __shared__ __attribute__((aligned(16))) u32 sh[secp::coop::WORDS];
    secp::pt acc = fold_coop_secp(products, n, sh);
    if (threadIdx.x == 0) C::store(out, 1, acc);
  } else if constexpr (__is_same(typename C::pt, p256::pt)) {
    __shared__ __attribute__((aligned(16))) u32 sh[p256::coop::WORDS];
    p256::pt acc = fold_coop_p256(products, n, sh);
    if (threadIdx.x == 0) C::store(out, 1, acc);
  } else {
    __shared__ __attribute__((aligned(16))) u32 sh[ed::coop::WORDS];
    ed::pt acc = fold_coop_ed(products, n, sh);
    if (threadIdx.x == 0) C::store(out, 1, acc);
  }
}

// xy[i] = to_affine(points[i]) as (x, y); inf[i] = 1 where the point is the identity
template <class C>
__global__ __launch_bounds__(TPB) void k_to_affine(const u32* __restrict__ points, u32* __restrict__ xy,
                                                   unsigned char* __restrict__ inf, size_t n) {
  __shared__ u32 lds_p[C::PW * TPB];
  const int valid = block_valid(n);
  const size_t first = (size_t)blockIdx.x * TPB;
  stage_in<C::PW>(lds_p, points + first * C::PW, valid);
  __syncthreads();
  const int e = threadIdx.x;
  if (e < valid) {
    typename C::pt p = C::load(lds_p + e, TPB);
    fe x, y;
    lmask m = C::to_affine(p, x, y);
    bool mine = lane_of(m);
    store_fe(lds_p + e, TPB, x);  // a lane reads and writes only its own LDS column
    store_fe(lds_p + 8 * TPB + e, TPB, y);
    inf[first + e] = mine ? 1 : 0;
  }
  __syncthreads();
  stage_out<16>(xy + first * 16, lds_p, valid);
}

// schnorr::batch_verify::<Secp256k1, D> (forge-ec-signature/src/schnorr.rs:194-290), the per-signature
// terms of the two folds at 262-281, challenges e_i and weights a_i supplied by the caller:
//   A_i = multiply(G, s_i * a_i)                                           (266-268)
//   B_i = multiply(from_affine(R_i) + multiply(from_affine(P_i), e_i), a_i) (273-280)
// as a pipeline over the ladder kernel (kernels_secp.hip): k_schnorr_pre (s_i * a_i, from_affine(P_i)),
// the fixed-base and one variable-base launch side by side in time, k_schnorr_mid (R_i + e_i P_i), the
// second variable-base launch.  (Round 1 ran the three ladders one after the other in one lane of one
// kernel: 58 spilled VGPRs, 8.5 ms at n = 4096 where two ladder latencies are 4.6 ms.)
template <class C>
__global__ __launch_bounds__(TPB) void k_schnorr_pre(const u32* __restrict__ pk_xy, const u32* __restrict__ ss,
                                                     const u32* __restrict__ as, u32* __restrict__ sa,
                                                     u32* __restrict__ p_out, unsigned char* __restrict__ wrapped, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  fe s, a;
  FEC_UNROLL for (int w = 0; w < 8; ++w) { s.w[w] = ss[i * 8 + w]; a.w[w] = as[i * 8 + w]; }
  bool ovf = false;
  const fe prod = C::sc_mul_flag(s, a, ovf);                  // impl Mul for Scalar (Ed25519: release profile, see Ed)
  if (ovf && wrapped) *wrapped = 1;                           // (every writer stores the same value)
  FEC_UNROLL for (int w = 0; w < 8; ++w) sa[i * 8 + w] = prod.w[w];
  // from_affine: the caller has rejected identities
  fe x, y;
  FEC_UNROLL for (int w = 0; w < 8; ++w) { x.w[w] = pk_xy[i * 16 + w]; y.w[w] = pk_xy[i * 16 + 8 + w]; }
  C::store(p_out + i * C::PW, 1, C::from_affine(x, y));
}
// q_i = from_affine(R_i) + ep_i   (277-279)
template <class C>
__global__ __launch_bounds__(TPB) void k_schnorr_mid(const u32* __restrict__ r_xy, const u32* __restrict__ ep,
                                                     u32* __restrict__ q_out, size_t n) {
  __shared__ u32 lds_p[C::PW * TPB];
  __shared__ u32 lds_r[16 * TPB];
  const int valid = block_valid(n);
  const size_t first = (size_t)blockIdx.x * TPB;
  stage_in<C::PW>(lds_p, ep + first * C::PW, valid);
  stage_in<16>(lds_r, r_xy + first * 16, valid);
  __syncthreads();
  const int e = threadIdx.x;
  if (e < valid) {
    const typename C::pt r = C::from_affine(load_fe(lds_r + e, TPB), load_fe(lds_r + 8 * TPB + e, TPB));
    C::store(lds_p + e, TPB, C::padd(r, C::load(lds_p + e, TPB)));
  }
  __syncthreads();
  stage_out<C::PW>(q_out + first * C::PW, lds_p, valid);
}

// The two strictly sequential folds (s_g += ..., r_e_p += ...: 268, 281) and the comparison at 286:
// block 0 folds the A terms, block 1 the B terms, each with the four-lane cooperative addition; the last block to finish
// converts both sums with to_affine and applies AffinePoint::ct_eq (1292-1296).
// out: [0..7] = x, y of to_affine(s_g), [8..15] of to_affine(r_e_p) (64-bit limbs as u32 pairs);
// flags: [0] = result, [1], [2] = the two infinity flags.
template <class C>
__global__ __launch_bounds__(64) void k_schnorr_fold_compare(const u32* __restrict__ terms_a,
                                                             const u32* __restrict__ terms_b,
                                                             u32* __restrict__ sums, u32* __restrict__ out_xy,
                                                             unsigned char* __restrict__ flags,
                                                             unsigned int* __restrict__ done, size_t n) {
  constexpr bool kSecp = __is_same(typename C::pt, secp::pt);
  constexpr bool kEd = __is_same(typename C::pt, ed::pt);
  __shared__ __attribute__((aligned(16))) u32 sh[kSecp ? secp::coop::WORDS : (kEd ? ed::coop::WORDS : p256::coop::WORDS)];
  const u32* terms = blockIdx.x == 0 ? terms_a : terms_b;
  typename C::pt acc;  // the whole wavefront folds: each addition on four (secp256k1, Ed25519) / five (P-256) lanes
  if constexpr (kSecp) acc = fold_coop_secp(terms, n, sh);
  else if constexpr (kEd) acc = fold_coop_ed(terms, n, sh);
  else acc = fold_coop_p256(terms, n, sh);
  if (threadIdx.x != 0) return;
  C::store(sums + blockIdx.x * C::PW, 1, acc);
  __threadfence();
  if (atomicAdd(done, 1u) != 1u) return;  // the other fold is still running: it will finish the job
  __threadfence();
  fe x[2], y[2];
  bool inf[2];
  bool panics = false;
#pragma unroll 1
  for (int k = 0; k < 2; ++k) {
    typename C::pt p = C::load(sums + k * C::PW, 1);
    fe xx, yy;
    if constexpr (kEd) {   // to_affine (1793-1811) unwraps z.invert(): a zero z of a point that is not the identity panics (1805)
      if (!lane_of(ed::is_identity(p)) && lane_of(fe_is_zero(p.z))) panics = true;
    }
    inf[k] = lane_of(C::to_affine(p, xx, yy));
    x[k] = xx;
    y[k] = yy;
    store_fe(out_xy + k * 16, 1, xx);
    store_fe(out_xy + k * 16 + 8, 1, yy);
  }
  const bool same = lane_of(fe_eq(x[0], x[1]) & fe_eq(y[0], y[1]));
  flags[0] = panics ? 2 : ((same || (inf[0] && inf[1])) ? 1 : 0);
  flags[1] = inf[0] ? 1 : 0;
  flags[2] = inf[1] ? 1 : 0;
}

// PointAffine::to_bytes -> [u8; 33] (secp256k1.rs:875-896, p256.rs:1558-1578, ed25519.rs:1505-1525;
// the same bytes as forge-ec-encoding CompressedPoint::from_affine, point.rs:38-67): 0x00 + zeros
// for the identity, else 0x02 | (y.to_bytes()[31] & 1), then x.to_bytes().  For Ed25519 to_bytes is
// little-endian, so byte 31 is the top byte and the "parity" is bit 248 of y -- reproduced as is.
// The workgroup's 256 x 33 bytes are assembled in LDS and written as coalesced dwords.
template <class C>
__global__ __launch_bounds__(TPB) void k_compress(const u32* __restrict__ xy, const unsigned char* __restrict__ inf,
                                                  unsigned char* __restrict__ out, size_t n) {
  __shared__ u32 lds_p[16 * TPB];
  __shared__ u32 lds_o[TPB * 33 / 4];
  const int valid = block_valid(n);
  const size_t first = (size_t)blockIdx.x * TPB;
  stage_in<16>(lds_p, xy + first * 16, valid);
  __syncthreads();
  const int e = threadIdx.x;
  if (e < valid) {
    unsigned char* o = reinterpret_cast<unsigned char*>(lds_o) + e * 33;
    const bool is_inf = inf != nullptr && inf[first + e] != 0;
    fe x = C::bytes_value(load_fe(lds_p + e, TPB));
    fe y = C::bytes_value(load_fe(lds_p + 8 * TPB + e, TPB));
    // y.to_bytes()[31]: the least significant byte when big-endian, the most significant otherwise
    const u32 odd = C::BYTES_BIG_ENDIAN ? (y.w[0] & 1u) : ((y.w[7] >> 24) & 1u);
    o[0] = is_inf ? 0 : (unsigned char)(2u + odd);
    FEC_UNROLL for (int k = 0; k < 32; ++k) {
      const u32 byte = (x.w[k >> 2] >> (8 * (k & 3))) & 0xFFu;  // byte k of the value, little-endian
      o[1 + (C::BYTES_BIG_ENDIAN ? 31 - k : k)] = is_inf ? 0 : (unsigned char)byte;
    }
  }
  __syncthreads();
  // first * 33 is a multiple of 4 (TPB * 33 = 8448)
  const int bytes = valid * 33, words = bytes >> 2;
  u32* g = reinterpret_cast<u32*>(out + first * 33);
  for (int v = threadIdx.x; v < words; v += TPB) g[v] = lds_o[v];
  if (threadIdx.x < (bytes & 3))
    out[first * 33 + (size_t)(words * 4 + threadIdx.x)] =
        reinterpret_cast<const unsigned char*>(lds_o)[words * 4 + threadIdx.x];
}

// Peak 32x32+64 multiply-add rate: 8 independent v_mad_u64_u32 chains per lane, no memory.
constexpr int PEAK_ITERS = 4096;
__global__ __launch_bounds__(TPB) void k_peak_mad32(u32* out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  u64 c0 = a, c1 = b, c2 = a + 1, c3 = b + 2, c4 = a + 3, c5 = b + 4, c6 = a + 5, c7 = b + 6;
  for (int it = 0; it < PEAK_ITERS; ++it) {
#define FEC_MAD8                                                                            \
  "v_mad_u64_u32 %0, s[10:11], %8, %9, %0\n v_mad_u64_u32 %1, s[12:13], %8, %9, %1\n"       \
  "v_mad_u64_u32 %2, s[14:15], %8, %9, %2\n v_mad_u64_u32 %3, s[16:17], %8, %9, %3\n"       \
  "v_mad_u64_u32 %4, s[18:19], %8, %9, %4\n v_mad_u64_u32 %5, s[20:21], %8, %9, %5\n"       \
  "v_mad_u64_u32 %6, s[22:23], %8, %9, %6\n v_mad_u64_u32 %7, s[24:25], %8, %9, %7\n"
    asm volatile(FEC_MAD8 FEC_MAD8 FEC_MAD8 FEC_MAD8 FEC_MAD8 FEC_MAD8 FEC_MAD8 FEC_MAD8
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
                 : "v"(a), "v"(b)
                 : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "s20",
                   "s21", "s22", "s23", "s24", "s25");
#undef FEC_MAD8
  }
  u64 cs = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (u32)cs ^ (u32)(cs >> 32);
}

// ---- fixed-base prefix tables, one level per launch (ensure_gen_prefix) -------------------------------------------
// P-256 (p256.rs:2126-2134, one step of the loop): child[g] = double(parent[g >> 1]), + base if g & 1
__global__ __launch_bounds__(TPB) void k_p256_prefix_level(const u32* __restrict__ parent, u32* __restrict__ child,
                                                           const u32* __restrict__ base, size_t n) {
  const size_t g = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (g >= n) return;
  p256::pt r = p256::pdouble(P256::load(parent + (g >> 1) * 24, 1));
  if (g & 1) r = p256::padd(r, P256::load(base, 1));
  P256::store(child + g * 24, 1, r);
}
// level 0 / entry 0: the multiplication's initial state -- secp256k1 (identity, base) with identity = (0, one(), 0), one()
// = raw 1 (secp256k1.rs:1322, 585); P-256 the identity (0, 1, 0) (p256.rs:1827); Ed25519 the identity (0, 1, 1, 0)
// (ed25519.rs:1776)
__global__ __launch_bounds__(64) void k_prefix_level0(int curve, const u32* __restrict__ base, u32* __restrict__ dst) {
  const int t = threadIdx.x;
  const int words = curve == FEC_SECP256K1 ? 48 : (curve == FEC_P256 ? 24 : 32);
  if (t >= words) return;
  u32 v = 0;
  if (t == 8) v = 1u;
  if (curve == FEC_ED25519 && t == 16) v = 1u;
  if (curve == FEC_SECP256K1 && t >= 24) v = base[t - 24];
  dst[t] = v;
}
// Ed25519 (ed25519.rs:2073-2094): the result after the low j + 1 bits whose bit j is set = the result after the low j
// bits + addend_j (`addend` = entry j of the doubling-chain table): upper[g] = lower[g] + addend, in place
__global__ __launch_bounds__(TPB) void k_ed_prefix_level(const u32* __restrict__ lower, u32* __restrict__ upper,
                                                         const u32* __restrict__ addend, size_t n) {
  const size_t g = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (g >= n) return;
  Ed::store(upper + g * 32, 1, ed::padd(Ed::load(lower + g * 32, 1), Ed::load(addend, 1)));
}

}  // namespace fecgpu


// ==========================================================================================
// host side: context + extern "C" ABI
// ==========================================================================================
using namespace fecgpu;
using namespace fecgpu::host;

namespace {

// Inputs of generator() as the reference writes them.  secp256k1 (secp256k1.rs:2608-2625) pushes
// the affine constants through its own to_montgomery(), i.e. Mul by the reference's R_SQUARED
// (219-235); Ed25519 (ed25519.rs:2015-2052) hard-codes x, y and from_affine() sets t = x*y.  Both
// products are evaluated on the device with the same Mul kernels at ctx creation, so the table
// below holds only the reference's literals.  P-256 (p256.rs:2092-2110) is plain affine, Z = 1.
const u64 SECP_GXY[8] = {0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL,
                         0x79BE667EF9DCBBACULL, 0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL,
                         0x5DA4FBFC0E1108A8ULL, 0x483ADA7726A3C465ULL};
const u64 SECP_R2X2[8] = {0x000E9F61ULL, 0x07A20000ULL, 0x00000100ULL, 0, 0x000E9F61ULL, 0x07A20000ULL,
                          0x00000100ULL, 0};
const u64 GEN_P256[12] = {0xF4A13945D898C296ULL, 0x77037D812DEB33A0ULL, 0xF8BCE6E563A440F2ULL,
                          0x6B17D1F2E12C4247ULL, 0xCBB6406837BF51F5ULL, 0x2BCE33576B315ECEULL,
                          0x8EE7EB4A7C0F9E16ULL, 0x4FE342E2FE1A7F9BULL, 1,
                          0,                     0,                     0};
const u64 GEN_ED[16] = {0x1A1462FAFB9683F2ULL, 0xD2E8A68B8B30C404ULL, 0xA0C0F3A1E9E71B63ULL,
                        0x216936D3CD6E53FEULL, 0x2DFC9311D90045F9ULL, 0x0A71C760BF38C6A7ULL,
                        0xA6FB8EEBCEAA2C8DULL, 0x5FD9C9E6CC3CCCCCULL, 1, 0, 0, 0,
                        0, 0, 0, 0};  // T is filled in on the device
const u64 FE_ONE[4] = {1, 0, 0, 0};
// Fixed-base prefix tables: 2^24 entries by default (secp256k1 3.0 GiB, built level by level in a few ms, see ensure_gen_prefix;
// 24 of the 256 ladder steps are then a table fetch); at most 2^28 (48 GiB for secp256k1: sized for 288 GB of HBM).
constexpr unsigned kDefaultPrefixBits = 24, kMaxPrefixBits = 28;
// ... and only for a ctx that multiplies by the generator in earnest: the table of a curve is built by the launch that
// takes the ctx past this many such multiplications (FEC_FIXED_PREFIX_AFTER; 0 after an explicit
// fec_ctx_set_fixed_prefix_bits).  2^21: the allocation and the build (tens of ms) are then below two batches' saving.
constexpr size_t kPrefixAfter = (size_t)1 << 21;


// The fixed-base prefix table of `curve`'s generator() (kernels_secp.hip: k_secp_mul MODE 2 / 3; kernels_p256.hip: claim();
// kernels_ed.hip: multiply_fixed_in_place), built on the stream of the fixed-base launch that takes the ctx past
// `prefix_after` multiplications by the generator (a table costs GiBs of device memory and milliseconds: a ctx that
// multiplies a few thousand scalars never pays for one; an explicit fec_ctx_set_fixed_prefix_bits builds at the next
// launch).  `n`: the elements of this launch.  Refused memory is not an error: the launches then run the whole ladder
// (SchedEnv carries a null table).
//
// The table grows level by level, every level one launch that performs ONE step of the reference's loop per entry:
//   secp256k1  level j entry g = one ladder step from level j-1 entry g >> 1 with the bit g & 1 (k_secp_mul<3>);
//              level 0 = (identity, G); levels alternate between the table and a scratch of half its size
//   P-256      level j entry g = double(level j-1 entry g >> 1), + G if g & 1; level 0 = identity; same two buffers
//   Ed25519    entries [2^j, 2^(j+1)) = entries [0, 2^j) + addend_j, in place; entry 0 = identity
// 2^(w+1) steps in all -- 2 to 4 ms at w = 24 -- instead of w steps per entry.
constexpr size_t prefix_entry_words(int curve) { return curve == FEC_SECP256K1 ? 48 : (curve == FEC_P256 ? 24 : 32); }
// bytes of the two buffers a w-bit table of `curve` is built in: the table, and (not Ed25519) the scratch of half its size
inline size_t prefix_table_bytes(int curve, unsigned w) { return ((size_t)1 << w) * prefix_entry_words(curve) * sizeof(u32); }
inline size_t prefix_half_bytes(int curve, unsigned w) { return curve == FEC_ED25519 ? 0 : (w ? prefix_table_bytes(curve, w - 1) : prefix_entry_words(curve) * sizeof(u32)); }

// Queues the w level launches that fill `tab` (2^w entries) for the base at device address `base` on stream s.
// `ed_addends`: the Ed25519 doubling-chain table of that base (ensure_ed_table).
void queue_prefix_levels(int curve, const u32* base, const u32* ed_addends, unsigned w, u32* tab, u32* half, hipStream_t s) {
  u32* buf[2] = {tab, half};                                       // level j lives in buf[(w - j) & 1]
  hipLaunchKernelGGL(k_prefix_level0, dim3(1), dim3(64), 0, s, curve, base, curve == FEC_ED25519 ? tab : buf[w & 1]);
  for (unsigned j = 1; j <= w; ++j) {
    const size_t cnt = (size_t)1 << j;                             // entries of level j
    if (curve == FEC_SECP256K1) {
      secp_prefix_level_launch(buf[(w - j + 1) & 1], buf[(w - j) & 1], cnt, s);
    } else if (curve == FEC_P256) {
      hipLaunchKernelGGL(k_p256_prefix_level, dim3(grid_for(cnt)), dim3(TPB), 0, s, (const u32*)buf[(w - j + 1) & 1],
                         buf[(w - j) & 1], base, cnt);
    } else {                                                       // addend_(j-1) = entry j - 1 of the doubling-chain table
      hipLaunchKernelGGL(k_ed_prefix_level, dim3(grid_for(cnt / 2)), dim3(TPB), 0, s, (const u32*)tab, tab + (cnt / 2) * 32,
                         ed_addends + (size_t)(j - 1) * 32, cnt / 2);
    }
  }
}

int ensure_ed_table(fec_ctx* ctx, const u64* d_base, const u64* host_base, hipStream_t s);

// ONE table per device, curve and size for the whole process: every ctx on that device (the [0, 0, 0] shard workers of a
// multi-device ctx, the ctxs of several host threads) holds a reference to the same allocation.  A table is complete
// before it is published (its builder waits for its stream), so a ctx that finds one needs no ordering with the builder.
struct SharedPrefix {
  int device, curve;
  unsigned bits;
  u32* table;
  int refs;
};
std::mutex g_prefix_mu;                 // guards g_prefix AND serialises builds (a second ctx waits, then shares)
std::vector<SharedPrefix> g_prefix;

// the largest w <= wanted (>= 16, else 0) whose table + build scratch fit `budget_pct` of the device's free memory
unsigned prefix_bits_within_budget(int curve, unsigned wanted, unsigned budget_pct) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  const double allowed = (double)free_b * (double)budget_pct / 100.0;
  const unsigned floor_w = wanted < 16 ? wanted : 16;   // (a table is never shrunk below 2^16 entries: then rather none)
  for (unsigned w = wanted; w >= floor_w && w > 0; --w)
    if ((double)prefix_table_bytes(curve, w) + (double)prefix_half_bytes(curve, w) <= allowed) return w;
  return 0;
}

// This ctx's reference to the shared table of `curve`, given up (the last reference frees the table; hipFree waits for
// the device, so no launch is still reading it).
void release_gen_prefix(fec_ctx* ctx, int curve) {
  u32* t = ctx->d_gen_prefix[curve];
  ctx->d_gen_prefix[curve] = nullptr;
  ctx->gen_prefix_bits[curve] = 0;
  if (!t) return;
  std::lock_guard<std::mutex> lock(g_prefix_mu);
  for (size_t i = 0; i < g_prefix.size(); ++i) {
    if (g_prefix[i].table != t) continue;
    if (--g_prefix[i].refs <= 0) {
      (void)hipSetDevice(g_prefix[i].device);
      (void)hipFree(g_prefix[i].table);
      g_prefix.erase(g_prefix.begin() + (long)i);
      if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
    }
    return;
  }
}

// Attaches the ctx to the process's table of `curve` -- an existing one of the wanted size, else of the size the memory
// budget allows -- building it (`may_build`) on stream s when there is none.  Returns 1 when the ctx now has a table, 0
// when there is none to be had (no memory within the budget, or building is not allowed here), -1 when the build itself
// failed (a launch or stream error: not a question of memory, the next launch may try again).
// No table is not an error: the launches then run the whole ladder (SchedEnv carries a null table).
int attach_gen_prefix(fec_ctx* ctx, int curve, hipStream_t s, bool may_build) {
  if (ctx->d_gen_prefix[curve]) return 1;
  if (ctx->prefix_bits == 0) return 0;
  std::lock_guard<std::mutex> lock(g_prefix_mu);
  auto find = [&](unsigned bits) -> SharedPrefix* {
    for (auto& e : g_prefix)
      if (e.device == ctx->device && e.curve == curve && e.bits == bits) return &e;
    return nullptr;
  };
  SharedPrefix* e = find(ctx->prefix_bits);
  unsigned w = ctx->prefix_bits;
  if (!e) {
    w = prefix_bits_within_budget(curve, ctx->prefix_bits, ctx->prefix_budget_pct);
    if (w == 0) return 0;
    e = find(w);
  }
  if (e) {
    ++e->refs;
    ctx->d_gen_prefix[curve] = e->table;
    ctx->gen_prefix_bits[curve] = e->bits;
    return 1;
  }
  if (!may_build) return 0;
  void* t = nullptr;
  void* half = nullptr;
  auto give_up = [&](int why) -> int {
    (void)hipGetLastError();
    if (t) (void)hipFree(t);
    if (half) (void)hipFree(half);
    return why;
  };
  if (hipMalloc(&t, prefix_table_bytes(curve, w)) != hipSuccess) return give_up(0);
  if (prefix_half_bytes(curve, w) != 0 && hipMalloc(&half, prefix_half_bytes(curve, w)) != hipSuccess) return give_up(0);
  if (curve == FEC_ED25519 && ensure_ed_table(ctx, ctx->d_gen[FEC_ED25519], ctx->h_gen_ed, s) != FEC_OK) return give_up(-1);
  order_after_previous(ctx, s);
  queue_prefix_levels(curve, reinterpret_cast<const u32*>(ctx->d_gen[curve]), ctx->d_ed_table, w, static_cast<u32*>(t),
                      static_cast<u32*>(half), s);
  // once per device and curve: wait here, so that a failed build never becomes a table and a finished one needs no event
  const bool launch_failed = hipGetLastError() != hipSuccess;
  const bool sync_failed = hipStreamSynchronize(s) != hipSuccess;
  if (launch_failed || sync_failed) return give_up(-1);
  if (half) (void)hipFree(half);
  half = nullptr;
  try {
    g_prefix.push_back(SharedPrefix{ctx->device, curve, w, static_cast<u32*>(t), 1});
  } catch (...) {
    return give_up(0);
  }
  ctx->d_gen_prefix[curve] = static_cast<u32*>(t);
  ctx->gen_prefix_bits[curve] = w;
  return 1;
}

// Called by every launch that multiplies `n` scalars by the generator.  Who may BUILD a table (gigabytes of device
// memory, a host synchronisation): a host-pointer entry point -- synchronous anyway -- of a ctx that has multiplied
// prefix_after scalars by the generator, and ANY launch of a ctx whose caller asked for tables
// (fec_ctx_set_fixed_prefix_bits / fec_ctx_build_fixed_prefix).  A *_dev entry point of a ctx left to its defaults only
// enqueues: it takes a table that already exists on its device (another ctx's, or an earlier host-pointer call's) and
// never allocates or waits.  A refusal is not permanent: the ctx asks again after another kPrefixAfter multiplications.
void ensure_gen_prefix(fec_ctx* ctx, int curve, hipStream_t s, size_t n) {
  if (ctx->prefix_bits == 0 || ctx->d_gen_prefix[curve]) return;
  ctx->fixed_elems[curve] += n;
  if (ctx->gen_prefix_tried[curve]) {
    if (ctx->fixed_elems[curve] < kPrefixAfter) return;
    ctx->gen_prefix_tried[curve] = false;          // (the device may have memory to spare by now)
  }
  if (ctx->fixed_elems[curve] < ctx->prefix_after) return;
  const bool may_build = ctx->prefix_explicit || ctx->in_host_call;
  if (attach_gen_prefix(ctx, curve, s, may_build) == 0 && may_build) {   // refused (budget, allocation): count afresh
    ctx->gen_prefix_tried[curve] = true;
    ctx->fixed_elems[curve] = 0;
  }
}

// Any OTHER fixed base: a table for this one launch, in the launch stream's scratch, sized to the batch -- 2^w entries
// with w = log2(n) - 2 cost n / 2 steps to build and save n * w: from 2^16 elements on.  `front`: bytes of the scratch
// the caller needs for itself, in front of the table.  On success `env` names the table for `base`; the scratch pointer
// is returned through `scratch` (null: no scratch could be had -- the caller's own `front` bytes included).
void per_call_prefix(fec_ctx* ctx, int curve, const u32* base, size_t n, hipStream_t s, size_t front, SchedEnv& env,
                     void** scratch) {
  *scratch = nullptr;
  unsigned w = 0;
  if (ctx->prefix_bits != 0 && n >= ((size_t)1 << 16)) {
    unsigned lg = 0;
    while (((size_t)2 << lg) <= n) ++lg;                           // floor(log2(n))
    w = lg - 2;
    if (w > ctx->prefix_bits) w = ctx->prefix_bits;
    if (w > 22) w = 22;
  }
  const size_t fr = (front + 255) & ~(size_t)255;
  const size_t bytes = fr + (w ? prefix_table_bytes(curve, w) + prefix_half_bytes(curve, w) : 0);
  if (bytes == 0) return;
  char* p = static_cast<char*>(scratch_for(ctx, s, bytes));
  if (!p && w) {                                                   // no room for a table: the caller's own bytes alone
    w = 0;
    p = front ? static_cast<char*>(scratch_for(ctx, s, fr)) : nullptr;
  }
  *scratch = p;
  if (!p || w == 0) return;
  u32* tab = reinterpret_cast<u32*>(p + fr);
  u32* half = prefix_half_bytes(curve, w) ? reinterpret_cast<u32*>(p + fr + prefix_table_bytes(curve, w)) : nullptr;
  queue_prefix_levels(curve, base, ctx->d_ed_table, w, tab, half, s);
  env.gen[curve] = base;
  env.gen_prefix[curve] = tab;
  env.gen_prefix_bits[curve] = w;
}
void drop_gen_prefix(fec_ctx* ctx) {
  for (int c = 0; c < 3; ++c) {
    release_gen_prefix(ctx, c);
    ctx->gen_prefix_tried[c] = false;
    ctx->fixed_elems[c] = 0;
  }
}

// Build (or reuse) the Ed25519 addend table for the base at device address d_base.  `host_base`
// (may be null) is the same point on the host and lets repeated calls with one base skip the build.
int ensure_ed_table(fec_ctx* ctx, const u64* d_base, const u64* host_base, hipStream_t s) {
  order_after_previous(ctx, s);  // the table is ctx-owned: a (re)build or a reuse on another stream waits for the last user
  if (!ctx->d_ed_table && hipMalloc(&ctx->d_ed_table, 256 * 32 * sizeof(u32)) != hipSuccess) {
    (void)hipGetLastError();
    return FEC_E_OOM;
  }
  if (host_base && ctx->ed_table_valid && std::memcmp(host_base, ctx->ed_table_base, 128) == 0) return FEC_OK;
  ed_build_table_launch(reinterpret_cast<const u32*>(d_base), ctx->d_ed_table, s);
  if (hipGetLastError() != hipSuccess) return FEC_E_LAUNCH;
  ctx->ed_table_valid = host_base != nullptr;
  if (host_base) std::memcpy(ctx->ed_table_base, host_base, 128);
  return FEC_OK;
}

int launch_ed_fixed(fec_ctx* ctx, const u64* ds, const u64* dbase, const u64* host_base, u64* dout, size_t n,
                    void* stream) {
  if (n == 0) return FEC_OK;
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  int rc = ensure_ed_table(ctx, dbase, host_base, s);
  if (rc != FEC_OK) return rc;
  const bool is_gen = dbase == ctx->d_gen[FEC_ED25519];
  if (is_gen) ensure_gen_prefix(ctx, FEC_ED25519, s, n);
  // per-stream scratch: the batch-wide popcount sort of large batches, and behind it the prefix table of a base that is
  // not the generator (per_call_prefix)
  void* work = nullptr;
  const size_t work_bytes = ed_fixed_work_bytes(n);
  Launch L(ctx, stream, work_bytes ? "k_ed_fixed_sorted (+ k_ed_pc_hist, k_ed_pc_scan, k_ed_pc_scatter)" : "k_ed_fixed_base");
  SchedEnv env = sched_env(ctx);
  if (is_gen) {
    if (work_bytes != 0 && !(work = scratch_for(ctx, s, work_bytes))) return FEC_E_OOM;
  } else {
    per_call_prefix(ctx, FEC_ED25519, reinterpret_cast<const u32*>(dbase), n, L.s, work_bytes, env, &work);
    if (work_bytes != 0 && !work) return FEC_E_OOM;
    if (work_bytes == 0) work = nullptr;   // (the scratch then holds the table alone: no sort area)
  }
  ed_fixed_launch(env, reinterpret_cast<const u32*>(ds), reinterpret_cast<const u32*>(dbase), ctx->d_ed_table,
                  reinterpret_cast<u32*>(dout), n, work, L.s);
  return L.done();
}

int launch_mul(fec_ctx* ctx, int curve, bool fixed, const u64* ds, const u64* dp, u64* dout, size_t n,
               void* stream) {
  if (n == 0) return FEC_OK;
  if (fixed && curve == FEC_ED25519) return launch_ed_fixed(ctx, ds, dp, nullptr, dout, n, stream);  // LDS addend table
  const u32* s = reinterpret_cast<const u32*>(ds);
  const u32* p = reinterpret_cast<const u32*>(dp);
  u32* o = reinterpret_cast<u32*>(dout);
  const char* name = curve == FEC_SECP256K1 ? (fixed ? "k_secp_mul<fixed>" : "k_secp_mul<var>")
                     : curve == FEC_P256    ? (fixed ? "k_p256_mul_sched<fixed>" : "k_p256_mul_sched<var>")
                                            : "k_ed_mul_pers";
  if (fixed && dp == ctx->d_gen[curve]) ensure_gen_prefix(ctx, curve, stream ? (hipStream_t)stream : ctx->stream, n);
  Launch L(ctx, stream, name);
  SchedEnv env = sched_env(ctx);
  if (fixed && dp != ctx->d_gen[curve]) {   // a base of the caller's own: a table for this launch (per_call_prefix)
    void* unused = nullptr;
    per_call_prefix(ctx, curve, p, n, L.s, 0, env, &unused);
  }
  switch (curve) {
    case FEC_SECP256K1: secp_launch_mul(env, fixed, s, p, o, n, L.s); break;
    case FEC_P256: p256_launch_mul(env, fixed, s, p, o, n, L.s); break;
    default: ed_launch_mul(env, s, p, o, n, L.s); break;
  }
  return L.done();
}

// A launch forked onto the ctx's second stream and joined back (events; no host blocking).  Inactive -- `s` is the
// main stream and fork_done / join do nothing -- for large batches, when the ctx has no second stream, or when the
// main stream IS the second stream.
struct SideStream {
  // Round 2 forked only up to 98304 elements (two launches side by side fill the chip: 256 CUs x 768 lanes / 2).  Measured
  // in round 3 (profiles/double_mul_side_stream_r03.jsonl): forking wins at EVERY size for the two Weierstrass curves --
  // the tail of one launch overlaps the head of the other, and below 2^18 elements a lone persistent kernel cannot fill
  // its 1024 slots per CU -- 2^17: secp256k1 8.68 -> 7.85 ms, P-256 8.80 -> 6.38 ms; 2^20: 57.56 -> 56.83, 48.65 -> 48.10.
  static constexpr size_t kSideStreamMax = (size_t)-1;
  hipStream_t main, s;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  bool active = false;
  SideStream(fec_ctx* ctx, hipStream_t main_, size_t n, size_t limit = kSideStreamMax) : main(main_), s(main_) {
    // (not inside a multi-chunk host pipeline: its second lane IS the side stream and is busy with the other chunk)
    if (n > limit || !ctx->stream2 || ctx->stream2 == main_ || ctx->in_multi_chunk_pipeline) return;
    if (hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ev_out, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    (void)hipEventRecord(ev_in, main);
    (void)hipStreamWaitEvent(ctx->stream2, ev_in, 0);
    s = ctx->stream2;
    active = true;
  }
  void fork_done() { if (active) (void)hipEventRecord(ev_out, s); }
  void join() { if (active) (void)hipStreamWaitEvent(main, ev_out, 0); }
  ~SideStream() {
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_out) (void)hipEventDestroy(ev_out);
  }
};

// out[i] = multiply(G, u1[i]) + multiply(q[i], u2[i])   (ecdsa.rs:254-256).
// Composed from the single-multiplication kernels: u1*G and u2*Q into per-stream scratch (secp256k1: the
// 3-waves-per-SIMD ladder, fixed then variable base; P-256: the task scheduler twice; Ed25519: the LDS
// addend-table kernel and the scheduler), then one point-addition pass in the reference's operand order.
// The fused masked-ladder forms of round 1 are gone (secp256k1: 64.9 ms fused against 60.7 ms composed).
int launch_double_mul(fec_ctx* ctx, int curve, const u64* d1, const u64* d2, const u64* dq, u64* dout,
                      size_t n, void* stream) {
  if (n == 0) return FEC_OK;
  const u32* a = reinterpret_cast<const u32*>(d1);
  const u32* b2 = reinterpret_cast<const u32*>(d2);
  const u32* q = reinterpret_cast<const u32*>(dq);
  const u32* gen = reinterpret_cast<const u32*>(ctx->d_gen[curve]);
  u32* o = reinterpret_cast<u32*>(dout);
  dim3 g(grid_for(n)), b(TPB);
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  const size_t pb = (size_t)plimbs(curve) * 8;
  const size_t ed_work = curve == FEC_ED25519 ? ed_fixed_work_bytes(n) : 0;
  char* scratch = static_cast<char*>(scratch_for(ctx, st, 2 * n * pb + ed_work));
  if (!scratch) return FEC_E_OOM;
  u32* ta = reinterpret_cast<u32*>(scratch);
  u32* tb = reinterpret_cast<u32*>(scratch + n * pb);
  if (curve == FEC_ED25519) {  // the addend table of G is built (once) before the timed sequence
    int rc = ensure_ed_table(ctx, ctx->d_gen[FEC_ED25519], ctx->h_gen_ed, st);
    if (rc != FEC_OK) return rc;
  }
  ensure_gen_prefix(ctx, curve, st, n);
  Launch L(ctx, stream, curve == FEC_SECP256K1 ? "k_secp_mul x2 + k_point_op"
                        : (curve == FEC_P256 ? "k_p256_mul_sched x2 + k_point_op" : "k_ed_fixed_base + k_ed_mul_pers + k_point_op"));
  // The fixed-base product runs on the ctx's second stream beside the variable-base one (the persistent kernels, one
  // workgroup per CU, each on half of the CUs): see SideStream.
  // (Ed25519: the table kernel is short and not capped to half of the CUs; side by side pays up to 2^15 elements)
  // (fec_ctx_set_side_stream_max -- the measurement knob of tools/double_mul_small_perf.py -- moves the limit)
  SideStream side(ctx, L.s, n, curve == FEC_ED25519 ? (size_t)1 << 15 : ctx->side_stream_max);
  if (curve == FEC_SECP256K1) {  // the 3-waves-per-SIMD ladder twice (fixed G, then Q) beats the fused 2-wave kernel
    secp_launch_mul(sched_env(ctx), true, a, gen, ta, n, side.s);
    side.fork_done();
    secp_launch_mul(sched_env(ctx), false, b2, q, tb, n, L.s);
    side.join();
    hipLaunchKernelGGL((k_point_op<Secp>), g, b, 0, L.s, (int)FEC_P_ADD, (const u32*)ta, (const u32*)tb, o, n);
  } else if (curve == FEC_P256) {
    SchedEnv ef = sched_env(ctx), ev = ef;
    if (side.active) p256_cu_split(sched_env(ctx), n, kP256VarMs, ef, ev);   // the CUs in proportion to the two launches' work
    p256_launch_mul(ef, true, a, gen, ta, n, side.s);
    side.fork_done();
    p256_launch_mul(ev, false, b2, q, tb, n, L.s);
    side.join();
    hipLaunchKernelGGL((k_point_op<P256>), g, b, 0, L.s, (int)FEC_P_ADD, (const u32*)ta, (const u32*)tb, o, n);
  } else {
    ed_fixed_launch(sched_env(ctx), a, gen, ctx->d_ed_table, ta, n, ed_work ? scratch + 2 * n * pb : nullptr, side.s);  // (ed_work != 0 only where the side stream is off)
    side.fork_done();
    ed_launch_mul(sched_env(ctx), b2, q, tb, n, L.s, side.active ? 2 : 1);
    side.join();
    hipLaunchKernelGGL((k_point_op<Ed>), g, b, 0, L.s, (int)FEC_P_ADD, (const u32*)ta, (const u32*)tb, o, n);
  }
  return L.done();
}

int launch_to_affine(fec_ctx* ctx, int curve, const u64* dp, u64* dxy, unsigned char* dinf, size_t n, void* stream) {
  if (n == 0) return FEC_OK;
  const u32* p = reinterpret_cast<const u32*>(dp);
  u32* o = reinterpret_cast<u32*>(dxy);
  dim3 g(grid_for(n)), b(TPB);
  Launch L(ctx, stream, "k_to_affine");
  switch (curve) {
    case FEC_SECP256K1: hipLaunchKernelGGL((k_to_affine<Secp>), g, b, 0, L.s, p, o, dinf, n); break;
    case FEC_P256: hipLaunchKernelGGL((k_to_affine<P256>), g, b, 0, L.s, p, o, dinf, n); break;
    default: hipLaunchKernelGGL((k_to_affine<Ed>), g, b, 0, L.s, p, o, dinf, n); break;
  }
  return L.done();
}

int launch_compress(fec_ctx* ctx, int curve, const u64* dxy, const unsigned char* dinf, unsigned char* dout, size_t n,
                    void* stream) {
  if (n == 0) return FEC_OK;
  const u32* p = reinterpret_cast<const u32*>(dxy);
  dim3 g(grid_for(n)), b(TPB);
  Launch L(ctx, stream, "k_compress");
  switch (curve) {
    case FEC_SECP256K1: hipLaunchKernelGGL((k_compress<Secp>), g, b, 0, L.s, p, dinf, dout, n); break;
    case FEC_P256: hipLaunchKernelGGL((k_compress<P256>), g, b, 0, L.s, p, dinf, dout, n); break;
    default: hipLaunchKernelGGL((k_compress<Ed>), g, b, 0, L.s, p, dinf, dout, n); break;
  }
  return L.done();
}

// Ecdsa::<C, D>::verify for secp256k1 / P-256: the pipeline of kernels_ecdsa.hip on per-stream scratch.
// (The single-kernel secp256k1 form of round 1 measured 70.7 ms per 2^20 against 65.7 ms for this pipeline.)
int launch_ecdsa_verify(fec_ctx* ctx, int curve, const unsigned char* dd, const u64* dr, const u64* ds, const u64* dpk,
                        const unsigned char* dinf, unsigned char* dstatus, size_t n, void* stream) {
  if (n == 0) return FEC_OK;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  void* work = scratch_for(ctx, st, ecdsa_work_bytes(n));
  if (!work) return FEC_E_OOM;
  ensure_gen_prefix(ctx, curve, st, n);
  Launch L(ctx, stream, curve == FEC_SECP256K1 ? "k_ecdsa_pre + k_secp_mul x2 + k_ecdsa_finish"
                                               : "k_ecdsa_pre + k_p256_mul_sched x2 + k_ecdsa_finish");
  ecdsa_launch(sched_env(ctx), curve, dd, reinterpret_cast<const u32*>(dr), reinterpret_cast<const u32*>(ds),
               reinterpret_cast<const u32*>(dpk), dinf, reinterpret_cast<const u32*>(ctx->d_gen[curve]), dstatus, work, n,
               L.s, ctx->stream2 != L.s ? ctx->stream2 : nullptr);
  return L.done();
}

// Eddsa::<Ed25519, D>::verify / Ed25519::verify from the point computation on (eddsa.rs:174-211, 430-447):
// A = from_affine(pk); s*G by the LDS addend-table kernel, k*A by the task scheduler; R + k*A, the two
// to_affine, the difference and is_identity in one finishing pass.  Work area: A, s*G, k*A (3 x 128 B).
int launch_eddsa_verify(fec_ctx* ctx, const u64* dr, const unsigned char* drinf, const u64* dpk, const unsigned char* dpinf,
                        const u64* ds, const u64* dk, unsigned char* dstatus, size_t n, void* stream) {
  if (n == 0) return FEC_OK;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  const size_t ed_work = ed_fixed_work_bytes(n);
  char* work = static_cast<char*>(scratch_for(ctx, st, n * 384 + ed_work));
  if (!work) return FEC_E_OOM;
  u32* a = reinterpret_cast<u32*>(work);
  u32* sg = reinterpret_cast<u32*>(work + n * 128);
  u32* ka = reinterpret_cast<u32*>(work + n * 256);
  int rc = ensure_ed_table(ctx, ctx->d_gen[FEC_ED25519], ctx->h_gen_ed, st);
  if (rc != FEC_OK) return rc;
  ensure_gen_prefix(ctx, FEC_ED25519, st, n);
  Launch L(ctx, stream, "k_eddsa_pre + k_ed_fixed_base + k_ed_mul_pers + k_eddsa_finish");
  eddsa_pre_launch(reinterpret_cast<const u32*>(dpk), dpinf, a, n, L.s);
  ed_fixed_launch(sched_env(ctx), reinterpret_cast<const u32*>(ds), reinterpret_cast<const u32*>(ctx->d_gen[FEC_ED25519]),
                  ctx->d_ed_table, sg, n, ed_work ? work + n * 384 : nullptr, L.s);
  ed_launch_mul(sched_env(ctx), reinterpret_cast<const u32*>(dk), a, ka, n, L.s);
  eddsa_finish_launch(sg, ka, reinterpret_cast<const u32*>(dr), drinf, dstatus, n, L.s);
  return L.done();
}

// Schnorr::<C, D>::verify per signature from the point computation on (schnorr.rs:90-140): A = from_affine(pk);
// s*G by the curve's fixed-base kernel (forked to the second stream for the Weierstrass curves), e*A by its
// variable-base kernel; the rest in one finishing pass.  Work area: A, s*G, e*A.
int launch_schnorr_verify(fec_ctx* ctx, int curve, const u64* dpk, const unsigned char* dpinf, const u64* dr,
                          const unsigned char* drinf, const u64* ds, const u64* de, unsigned char* dstatus, size_t n,
                          void* stream) {
  if (n == 0) return FEC_OK;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  const size_t pb = (size_t)plimbs(curve) * 8;
  const size_t ed_work = curve == FEC_ED25519 ? ed_fixed_work_bytes(n) : 0;
  char* work = static_cast<char*>(scratch_for(ctx, st, schnorr_verify_work_bytes(curve, n) + ed_work));
  if (!work) return FEC_E_OOM;
  u32* a = reinterpret_cast<u32*>(work);
  u32* sg = reinterpret_cast<u32*>(work + n * pb);
  u32* ep = reinterpret_cast<u32*>(work + 2 * n * pb);
  const u32* gen = reinterpret_cast<const u32*>(ctx->d_gen[curve]);
  const u32* sc = reinterpret_cast<const u32*>(ds);
  const u32* ec = reinterpret_cast<const u32*>(de);
  if (curve == FEC_ED25519) {
    int rc = ensure_ed_table(ctx, ctx->d_gen[FEC_ED25519], ctx->h_gen_ed, st);
    if (rc != FEC_OK) return rc;
  }
  ensure_gen_prefix(ctx, curve, st, n);
  Launch L(ctx, stream, curve == FEC_SECP256K1 ? "k_schnorr_verify_pre + k_secp_mul x2 + k_schnorr_verify_finish"
                        : (curve == FEC_P256 ? "k_schnorr_verify_pre + k_p256_mul_sched x2 + k_schnorr_verify_finish"
                                             : "k_schnorr_verify_pre + k_ed_fixed_base + k_ed_mul_pers + k_schnorr_verify_finish"));
  schnorr_verify_pre_launch(curve, reinterpret_cast<const u32*>(dpk), dpinf, a, n, L.s);
  if (curve == FEC_ED25519) {
    ed_fixed_launch(sched_env(ctx), sc, gen, ctx->d_ed_table, sg, n, ed_work ? work + 3 * n * pb : nullptr, L.s);
    ed_launch_mul(sched_env(ctx), ec, a, ep, n, L.s);
  } else {
    SideStream side(ctx, L.s, n);
    if (curve == FEC_SECP256K1) {
      secp_launch_mul(sched_env(ctx), true, sc, gen, sg, n, side.s);
      side.fork_done();
      secp_launch_mul(sched_env(ctx), false, ec, a, ep, n, L.s);
    } else {
      SchedEnv ef = sched_env(ctx), ev = ef;
      if (side.active) p256_cu_split(sched_env(ctx), n, kP256VarAffineMs, ef, ev);   // (e * from_affine(P): affine addend)
      p256_launch_mul(ef, true, sc, gen, sg, n, side.s);
      side.fork_done();
      p256_launch_mul(ev, false, ec, a, ep, n, L.s);
    }
    side.join();
  }
  schnorr_verify_finish_launch(curve, sg, ep, reinterpret_cast<const u32*>(dr), drinf, dstatus, n, L.s);
  return L.done();
}

// Curve::validate_point per affine point (kernels_ecdsa.hip)
int launch_validate(fec_ctx* ctx, int curve, const u64* dxy, const unsigned char* dinf, unsigned char* dok, size_t n, void* stream) {
  if (n == 0) return FEC_OK;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  void* work = nullptr;
  if (validate_work_bytes(curve, n)) {
    work = scratch_for(ctx, st, validate_work_bytes(curve, n));
    if (!work) return FEC_E_OOM;
  }
  Launch L(ctx, stream, curve == FEC_ED25519 ? "k_ed_validate_pre + k_ed_mul_pers x2 + k_ed_validate_finish" : "k_validate_weierstrass");
  validate_launch(sched_env(ctx), curve, reinterpret_cast<const u32*>(dxy), dinf, dok, work, n, L.s);
  return L.done();
}

// KeyExchange::derive_shared_secret for secp256k1 / P-256 on per-stream scratch (kernels_ecdsa.hip)
int launch_ecdh(fec_ctx* ctx, int curve, const u64* dsk, const u64* dpk, const unsigned char* dinf, unsigned char* dout,
                unsigned char* dstatus, size_t n, void* stream) {
  if (n == 0) return FEC_OK;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  void* work = scratch_for(ctx, st, ecdh_work_bytes(n));
  if (!work) return FEC_E_OOM;
  Launch L(ctx, stream, curve == FEC_SECP256K1 ? "k_ecdh_pre + k_secp_mul + k_ecdh_finish" : "k_ecdh_pre + k_p256_mul_sched + k_ecdh_finish");
  ecdh_launch(sched_env(ctx), curve, reinterpret_cast<const u32*>(dsk), reinterpret_cast<const u32*>(dpk), dinf, reinterpret_cast<u32*>(dout),
              dstatus, work, n, L.s);
  return L.done();
}

int launch_field(fec_ctx* ctx, int curve, int op, const u64* da, const u64* db, u64* dout, size_t n,
                 void* stream = nullptr) {
  if (n == 0) return FEC_OK;
  const u32* a = reinterpret_cast<const u32*>(da);
  const u32* bb = reinterpret_cast<const u32*>(db);
  u32* o = reinterpret_cast<u32*>(dout);
  dim3 g(grid_for(n)), b(TPB);
  Launch L(ctx, stream, "k_field_op");
  switch (curve) {
    case FEC_SECP256K1: hipLaunchKernelGGL((k_field_op<Secp>), g, b, 0, L.s, op, a, bb, o, n); break;
    case FEC_P256: hipLaunchKernelGGL((k_field_op<P256>), g, b, 0, L.s, op, a, bb, o, n); break;
    default: hipLaunchKernelGGL((k_field_op<Ed>), g, b, 0, L.s, op, a, bb, o, n); break;
  }
  return L.done();
}

int launch_point(fec_ctx* ctx, int curve, int op, const u64* dp, const u64* dq, u64* dout, size_t n,
                 void* stream = nullptr) {
  if (n == 0) return FEC_OK;
  const u32* p = reinterpret_cast<const u32*>(dp);
  const u32* q = reinterpret_cast<const u32*>(dq);
  u32* o = reinterpret_cast<u32*>(dout);
  dim3 g(grid_for(n)), b(TPB);
  Launch L(ctx, stream, "k_point_op");
  switch (curve) {
    case FEC_SECP256K1: hipLaunchKernelGGL((k_point_op<Secp>), g, b, 0, L.s, op, p, q, o, n); break;
    case FEC_P256: hipLaunchKernelGGL((k_point_op<P256>), g, b, 0, L.s, op, p, q, o, n); break;
    default: hipLaunchKernelGGL((k_point_op<Ed>), g, b, 0, L.s, op, p, q, o, n); break;
  }
  return L.done();
}

// Host-pointer batches run as a two-lane software pipeline over chunks of ctx->chunk elements:
//   H2D(c) K(c) on lane c%2, then D2H(c-1) on the other lane -- so while the host waits for
// chunk c-1's results the GPU is already running chunk c.  Copies from/to pageable caller memory
// overlap the kernels of the other lane; device staging is bounded by two chunks however large n
// is.  An input with stride 0 is shared by all elements (a fixed base) and copied once per lane.
struct HostIn {
  const void* ptr;
  size_t stride;  // bytes per element; 0 = one shared value of `bytes` bytes
  size_t bytes;   // only for stride == 0
};
template <class F>
int host_pipeline(fec_ctx* ctx, size_t n, const HostIn (&in)[3], void* hout, size_t out_stride, F body) {
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  const size_t chunk = pipeline_chunk(ctx);
  const size_t nchunks = (n + chunk - 1) / chunk;
  struct InPipeline {  // (SideStream: a multi-chunk pipeline keeps both streams busy by itself)
    fec_ctx* c;
    ~InPipeline() { c->in_multi_chunk_pipeline = false; }
  } in_pipeline{ctx};
  ctx->in_multi_chunk_pipeline = nchunks > 1;
  hipStream_t lanes[2] = {ctx->stream, ctx->stream2};
  auto copy_back = [&](size_t c) -> int {
    const int lane = (int)(c & 1);
    const size_t lo = c * chunk, cnt = (lo + chunk <= n ? chunk : n - lo);
    if (hipMemcpyAsync((char*)hout + lo * out_stride, ctx->d_buf[lane * 4 + 3], cnt * out_stride,
                       hipMemcpyDeviceToHost, lanes[lane]) != hipSuccess)
      return FEC_E_DEVICE;
    return FEC_OK;
  };
  return drained(ctx, [&]() -> int {  // (a failure half-way leaves nothing queued on the caller's arrays)
  for (size_t c = 0; c < nchunks; ++c) {
    const int lane = (int)(c & 1);
    const size_t lo = c * chunk, cnt = (lo + chunk <= n ? chunk : n - lo);
    void* d_in[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < 3; ++i) {
      if (!in[i].ptr) continue;
      const size_t bytes = in[i].stride ? cnt * in[i].stride : in[i].bytes;
      int rc = ensure(ctx, lane * 4 + i, in[i].stride ? (chunk < n ? chunk : n) * in[i].stride : in[i].bytes);
      if (rc != FEC_OK) return rc;
      d_in[i] = ctx->d_buf[lane * 4 + i];
      if (in[i].stride == 0 && c >= 2) continue;  // the shared value is already on this lane
      const char* src = (const char*)in[i].ptr + (in[i].stride ? lo * in[i].stride : 0);
      if (hipMemcpyAsync(d_in[i], src, bytes, hipMemcpyHostToDevice, lanes[lane]) != hipSuccess)
        return FEC_E_DEVICE;
    }
    int rc = ensure(ctx, lane * 4 + 3, (chunk < n ? chunk : n) * out_stride);
    if (rc != FEC_OK) return rc;
    rc = body(d_in[0], d_in[1], d_in[2], ctx->d_buf[lane * 4 + 3], cnt, (void*)lanes[lane]);
    if (rc != FEC_OK) return rc;
    if (c > 0) {
      rc = copy_back(c - 1);
      if (rc != FEC_OK) return rc;
    }
  }
  int rc = copy_back(nchunks - 1);
  if (rc != FEC_OK) return rc;
  return sync_and_check(ctx, ctx->stream, ctx->stream2);
  });
}

// Multi-device ctx: one host thread per shard worker, each running `call(g)` for its child ctx.  Returns the first
// failure in shard order.
constexpr size_t kMaxShards = 16;  // fec_ctx_create_multi's limit
template <class F>
int multi_each(fec_ctx* ctx, F call) {
  const size_t N = ctx->children.size();
  if (N == 0 || N > kMaxShards) return FEC_E_ARG;
  // fixed-size state: nothing here allocates, so the only thing that can throw is the creation of a thread, and
  // an exception inside a worker is caught inside the worker (it would otherwise terminate the process)
  int rc[kMaxShards];
  std::thread workers[kMaxShards];
  for (size_t g = 0; g < N; ++g) rc[g] = FEC_OK;
  for (size_t g = 0; g < N; ++g) {
    try {
      workers[g] = std::thread([&rc, &call, g] {
        try {
          rc[g] = call(g);
        } catch (const std::bad_alloc&) {
          rc[g] = FEC_E_OOM;
        } catch (...) {
          rc[g] = FEC_E_DEVICE;
        }
      });
    } catch (...) {  // std::system_error: the thread could not be started
      rc[g] = FEC_E_COMM;
    }
  }
  for (size_t g = 0; g < N; ++g)
    if (workers[g].joinable()) workers[g].join();
  for (size_t g = 0; g < N; ++g)
    if (rc[g] != FEC_OK) return rc[g];
  return FEC_OK;
}
// Host-pointer calls: contiguous shards [g*n/N, (g+1)*n/N), each worker calling the single-device entry point on its
// child ctx with offset pointers.
template <class F>
int multi_shard(fec_ctx* ctx, size_t n, F call) {
  const size_t N = ctx->children.size();
  if (N == 0 || N > kMaxShards) return FEC_E_ARG;
  return multi_each(ctx, [&](size_t g) -> int {
    const size_t lo = n / N * g + (n % N) * g / N, hi = n / N * (g + 1) + (n % N) * (g + 1) / N;
    return hi == lo ? (int)FEC_OK : call(ctx->children[g], lo, hi - lo);
  });
}

// Device-RESIDENT shards (fec_multi_batch_*_dev): shard g -- counts[g] elements -- already sits in the memory of the
// ctx's g-th device; `launch(child, g, lo, cnt, out, stream)` enqueues the kernels for elements [lo, lo + cnt) of that
// shard.  With `gathered` (an array on the consumer-th device of the ctx) every shard's results are also copied into
// it at the shard's offset -- peer copies over the devices' own xGMI link to the consumer (SURVEY.md section 8e's direct
// pattern: every device writes its block to the consumer, nothing is relayed), chunk by chunk on a stream of their own so
// that the copy of one chunk runs under the kernels of the next.  Synchronous: returns when every shard and copy has
// completed and every device's error word has been read.
template <class F>
int multi_dev_run(fec_ctx* ctx, int curve, const size_t* counts, uint64_t* const* out, uint64_t* gathered, int consumer,
                  void* const* streams, F launch) {
  const size_t N = ctx->children.size();
  if (N == 0 || N > kMaxShards) return FEC_E_ARG;
  if (gathered && (consumer < 0 || (size_t)consumer >= N)) return FEC_E_ARG;
  const size_t pl = (size_t)plimbs(curve);
  size_t offset[kMaxShards + 1];
  offset[0] = 0;
  for (size_t g = 0; g < N; ++g) offset[g + 1] = offset[g] + counts[g];
  const int dst_dev = gathered ? ctx->children[(size_t)consumer]->device : -1;
  return multi_each(ctx, [&](size_t g) -> int {
    fec_ctx* c = ctx->children[g];
    const size_t cnt = counts[g];
    if (cnt == 0) return FEC_OK;
    if (hipSetDevice(c->device) != hipSuccess) return FEC_E_DEVICE;
    hipStream_t ks = streams && streams[g] ? (hipStream_t)streams[g] : c->stream;
    if (gathered) {
      if (!c->stream_gather && hipStreamCreateWithFlags(&c->stream_gather, hipStreamNonBlocking) != hipSuccess) return FEC_E_COMM;
      if (!c->ev_gather && hipEventCreateWithFlags(&c->ev_gather, hipEventDisableTiming) != hipSuccess) return FEC_E_COMM;
      if (dst_dev != c->device) {   // direct access to the consumer's memory (already enabled is fine; refused: the copy is staged by the runtime)
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, c->device, dst_dev) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(dst_dev, 0);
        (void)hipGetLastError();
      }
    }
    int rc = drained(c, [&]() -> int {
      const size_t chunk = c->chunk < cnt ? c->chunk : cnt;
      for (size_t lo = 0; lo < cnt; lo += chunk) {
        const size_t m = lo + chunk <= cnt ? chunk : cnt - lo;
        uint64_t* o = out[g] + lo * pl;
        const int r = launch(c, g, lo, m, o, (void*)ks);
        if (r != FEC_OK) return r;
        if (gathered) {
          if (hipEventRecord(c->ev_gather, ks) != hipSuccess || hipStreamWaitEvent(c->stream_gather, c->ev_gather, 0) != hipSuccess ||
              hipMemcpyPeerAsync(gathered + (offset[g] + lo) * pl, dst_dev, o, c->device, m * pl * 8, c->stream_gather) != hipSuccess) {
            (void)hipGetLastError();
            return FEC_E_COMM;
          }
        }
      }
      int r = sync_and_check(c, ks);
      if (r == FEC_OK && gathered && hipStreamSynchronize(c->stream_gather) != hipSuccess) {
        (void)hipGetLastError();
        r = FEC_E_COMM;
      }
      return r;
    });
    if (rc != FEC_OK) {   // nothing stays queued on the caller's arrays
      (void)hipStreamSynchronize(ks);
      if (c->stream_gather) (void)hipStreamSynchronize(c->stream_gather);
      (void)hipGetLastError();
      (void)take_device_error(c);
    }
    return rc;
  });
}
inline bool is_multi(const fec_ctx* ctx) { return ctx && !ctx->children.empty(); }

}  // namespace

extern "C" {

int fec_point_limbs(fec_curve curve) { return curve_ok(curve) ? plimbs(curve) : 0; }

const char* fec_strerror(int status) try {
  switch (status) {
    case FEC_OK: return "ok";
    case FEC_E_ARG: return "invalid argument";
    case FEC_E_DEVICE: return "no usable gfx950 GPU / HIP runtime error (there is no CPU fallback)";
    case FEC_E_OOM: return "out of device memory";
    case FEC_E_LAUNCH: return "kernel launch or execution failed";
    case FEC_E_UNSUPPORTED: return "operation not supported for this curve or for a multi-device ctx";
    case FEC_E_COMM: return "multi-device ctx: a shard worker could not be started, or a copy between two devices failed";
    default: return "unknown fecgpu status";
  }
} FEC_ABI_CATCH_NULL

int fec_ctx_create(fec_ctx** out, int device) try {
  if (!out) return FEC_E_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return FEC_E_DEVICE;
  }
  if (device < 0 || device >= count) return FEC_E_ARG;
  if (hipSetDevice(device) != hipSuccess) return FEC_E_DEVICE;
  fec_ctx* ctx = new (std::nothrow) fec_ctx();
  if (!ctx) return FEC_E_OOM;
  ctx->device = device;
  {
    const char* e = std::getenv("FEC_CANON_COMB4");
    ctx->canon_use_comb8 = !(e && e[0] == '1');
    const char* w = std::getenv("FEC_FIXED_PREFIX_BITS");   // fixed-base prefix tables (ensure_gen_prefix): 0 = off
    ctx->prefix_bits = kDefaultPrefixBits;
    if (w && *w) {
      const unsigned long v = std::strtoul(w, nullptr, 10);
      ctx->prefix_bits = v > kMaxPrefixBits ? kMaxPrefixBits : (unsigned)v;
    }
    const char* after = std::getenv("FEC_FIXED_PREFIX_AFTER");
    ctx->prefix_after = after && *after ? (size_t)std::strtoull(after, nullptr, 10) : kPrefixAfter;
    const char* side = std::getenv("FEC_SIDE_STREAM_MAX");
    if (side && *side) ctx->side_stream_max = (size_t)std::strtoull(side, nullptr, 10);
    // (the three variables are overrides for experiments, read once here; the interface is fec_ctx_set_fixed_prefix_bits /
    // _after / _budget and fec_ctx_set_side_stream_max)
  }
  if (hipGetDeviceProperties(&ctx->prop, device) != hipSuccess ||
      std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0 ||
      hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
    (void)hipGetLastError();
    fec_ctx_destroy(ctx);
    return FEC_E_DEVICE;
  }
  for (int c = 0; c < 3; ++c) {
    if (hipMalloc(&ctx->d_gen[c], (size_t)plimbs(c) * 8) != hipSuccess) {
      (void)hipGetLastError();
      fec_ctx_destroy(ctx);
      return FEC_E_OOM;
    }
  }
  {  // the device error word: pinned host memory the kernels can write (kernels.hpp: SchedEnv)
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      (void)hipGetLastError();
      if (h) (void)hipHostFree(h);
      fec_ctx_destroy(ctx);
      return FEC_E_OOM;
    }
    std::memset(h, 0, 64);
    ctx->h_err = static_cast<unsigned*>(h);
    ctx->d_err = static_cast<unsigned*>(d);
  }
  {
    // every copy goes to the ctx stream: it is non-blocking, i.e. NOT ordered with NULL-stream work
    bool ok = ensure(ctx, 0, 64) == FEC_OK && ensure(ctx, 1, 64) == FEC_OK;
    ok = ok && hipMemcpyAsync(ctx->d_buf[0], SECP_GXY, 64, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(ctx->d_buf[1], SECP_R2X2, 64, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(ctx->d_gen[0] + 8, FE_ONE, 32, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(ctx->d_gen[1], GEN_P256, 96, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(ctx->d_gen[2], GEN_ED, 128, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    // secp256k1: (gx, gy) * R_SQUARED;  Ed25519: t = x * y
    ok = ok && launch_field(ctx, FEC_SECP256K1, FEC_F_MUL, (const u64*)ctx->d_buf[0],
                            (const u64*)ctx->d_buf[1], ctx->d_gen[0], 2) == FEC_OK;
    ok = ok && launch_field(ctx, FEC_ED25519, FEC_F_MUL, ctx->d_gen[2], ctx->d_gen[2] + 4,
                            ctx->d_gen[2] + 12, 1) == FEC_OK;
    ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess;
    ok = ok && hipMemcpy(ctx->h_gen_ed, ctx->d_gen[2], 128, hipMemcpyDeviceToHost) == hipSuccess;
    for (int c = 0; c < 3 && ok; ++c)
      ok = hipMemcpy(ctx->h_gen[c], ctx->d_gen[c], (size_t)plimbs(c) * 8, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      fec_ctx_destroy(ctx);
      return FEC_E_LAUNCH;
    }
  }
  *out = ctx;
  return FEC_OK;
} FEC_ABI_CATCH_STATUS

int fec_ctx_create_multi(fec_ctx** out, const int* devices, int n_devices) try {
  if (!out) return FEC_E_ARG;
  *out = nullptr;
  if (n_devices < 1 || n_devices > 16) return FEC_E_ARG;
  fec_ctx* parent = new (std::nothrow) fec_ctx();
  if (!parent) return FEC_E_OOM;
  for (int g = 0; g < n_devices; ++g) {
    fec_ctx* child = nullptr;
    int rc = fec_ctx_create(&child, devices ? devices[g] : g);
    if (rc != FEC_OK) {
      fec_ctx_destroy(parent);
      return rc;
    }
    try {
      parent->children.push_back(child);
    } catch (...) {
      fec_ctx_destroy(child);
      fec_ctx_destroy(parent);
      return FEC_E_OOM;
    }
  }
  parent->device = parent->children[0]->device;
  *out = parent;
  return FEC_OK;
} FEC_ABI_CATCH_STATUS

int fec_ctx_device_count(fec_ctx* ctx) { return !ctx ? 0 : (ctx->children.empty() ? 1 : (int)ctx->children.size()); }

void fec_ctx_destroy(fec_ctx* ctx) try {
  if (!ctx) return;
  if (!ctx->children.empty()) {
    for (fec_ctx* c : ctx->children) fec_ctx_destroy(c);
    delete ctx;
    return;
  }
  if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
  // Wipe BEFORE anything is freed: nothing a caller passed in outlives the ctx in device memory, and no memset is ever
  // issued on an address that has gone back to the allocator (round 2 freed the staging buffers first and wiped
  // afterwards: the wipe then either failed on the first freed pointer and skipped everything else, or zeroed
  // memory that another ctx had been handed in the meantime).
  (void)fec_ctx_wipe(ctx);
  for (int i = 0; i < 8; ++i) {
    if (ctx->d_buf[i]) (void)hipFree(ctx->d_buf[i]);
    ctx->d_buf[i] = nullptr;
    ctx->d_cap[i] = 0;
  }
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  ctx->stream2 = nullptr;
  for (int i = 0; i < 3; ++i) {
    if (ctx->d_gen[i]) (void)hipFree(ctx->d_gen[i]);
    ctx->d_gen[i] = nullptr;
  }
  if (ctx->h_err) (void)hipHostFree(ctx->h_err);
  ctx->h_err = ctx->d_err = nullptr;
  if (ctx->d_ed_table) (void)hipFree(ctx->d_ed_table);
  drop_gen_prefix(ctx);
  for (auto& e : ctx->stream_scratch)
    if (e.buf) (void)hipFree(e.buf);
  if (ctx->ev_order) (void)hipEventDestroy(ctx->ev_order);
  if (ctx->ev_gather) (void)hipEventDestroy(ctx->ev_gather);
  if (ctx->stream_gather) (void)hipStreamDestroy(ctx->stream_gather);
  for (int i = 0; i < 3; ++i)
    if (ctx->d_canon_comb[i]) (void)hipFree(ctx->d_canon_comb[i]);
  for (int i = 0; i < 3; ++i)
    if (ctx->d_canon_comb8[i]) (void)hipFree(ctx->d_canon_comb8[i]);
  if (ctx->d_win_scratch) (void)hipFree(ctx->d_win_scratch);
  if (ctx->d_zbuf) (void)hipFree(ctx->d_zbuf);
  if (ctx->d_tbuf) (void)hipFree(ctx->d_tbuf);
  if (ctx->d_verify) (void)hipFree(ctx->d_verify);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
} FEC_ABI_CATCH_VOID

int fec_batch_mul_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* ds, const uint64_t* dp, uint64_t* dout,
                      size_t n, void* stream) try {
  if (is_multi(ctx)) return FEC_E_UNSUPPORTED;  // device pointers belong to one device
  if (!ctx || !curve_ok(curve) || (n && (!ds || !dp || !dout))) return FEC_E_ARG;
  if (!aligned16(ds) || !aligned16(dp) || !aligned16(dout)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return launch_mul(ctx, curve, false, ds, dp, dout, n, stream);
} FEC_ABI_CATCH_STATUS

int fec_batch_mul_fixed_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* ds, const uint64_t* dbase,
                            uint64_t* dout, size_t n, void* stream) try {
  if (is_multi(ctx)) return FEC_E_UNSUPPORTED;  // device pointers belong to one device
  if (!ctx || !curve_ok(curve) || (n && (!ds || !dbase || !dout))) return FEC_E_ARG;
  if (!aligned16(ds) || !aligned16(dbase) || !aligned16(dout)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  if (curve == FEC_ED25519) {
    // the ctx's own generator (fec_generator_dev) is recognised by address, so its table is built once
    const u64* host_base = dbase == ctx->d_gen[FEC_ED25519] ? ctx->h_gen_ed : nullptr;
    return launch_ed_fixed(ctx, ds, dbase, host_base, dout, n, stream);
  }
  return launch_mul(ctx, curve, true, ds, dbase, dout, n, stream);
} FEC_ABI_CATCH_STATUS

int fec_batch_double_mul_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d1, const uint64_t* d2,
                             const uint64_t* dq, uint64_t* dout, size_t n, void* stream) try {
  if (is_multi(ctx)) return FEC_E_UNSUPPORTED;  // device pointers belong to one device
  if (!ctx || !curve_ok(curve) || (n && (!d1 || !d2 || !dq || !dout))) return FEC_E_ARG;
  if (!aligned16(d1) || !aligned16(d2) || !aligned16(dq) || !aligned16(dout)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return launch_double_mul(ctx, curve, d1, d2, dq, dout, n, stream);
} FEC_ABI_CATCH_STATUS

// ---- device-resident shards of a multi-device ctx (include/fecgpu.h: fec_multi_batch_*_dev) ----
namespace {
int multi_dev_args(fec_ctx* ctx, fec_curve curve, const void* const* a, const void* const* b, uint64_t* const* out,
                   const size_t* counts) {
  if (!ctx || !curve_ok(curve) || !counts || !a || !out) return FEC_E_ARG;
  if (!is_multi(ctx)) return FEC_E_UNSUPPORTED;   // a single-device ctx has fec_batch_*_dev
  for (size_t g = 0; g < ctx->children.size(); ++g) {
    if (counts[g] == 0) continue;
    if (!a[g] || !out[g] || (b && !b[g])) return FEC_E_ARG;
    if (!aligned16(a[g]) || !aligned16(out[g]) || (b && !aligned16(b[g]))) return FEC_E_ARG;
  }
  return FEC_OK;
}
}  // namespace

int fec_multi_batch_mul_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* const* scalars, const uint64_t* const* points,
                            uint64_t* const* out, const size_t* counts, uint64_t* gathered, int consumer,
                            void* const* streams) try {
  int rc = multi_dev_args(ctx, curve, (const void* const*)scalars, (const void* const*)points, out, counts);
  if (rc != FEC_OK) return rc;
  if (gathered && !aligned16(gathered)) return FEC_E_ARG;
  const size_t pl = (size_t)plimbs(curve);
  return multi_dev_run(ctx, curve, counts, out, gathered, consumer, streams,
                       [&](fec_ctx* c, size_t g, size_t lo, size_t m, uint64_t* o, void* s) {
                         return launch_mul(c, curve, false, scalars[g] + lo * 4, points[g] + lo * pl, o, m, s);
                       });
} FEC_ABI_CATCH_STATUS

int fec_multi_batch_mul_fixed_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* const* scalars, const uint64_t* const* bases,
                                  uint64_t* const* out, const size_t* counts, uint64_t* gathered, int consumer,
                                  void* const* streams) try {
  int rc = multi_dev_args(ctx, curve, (const void* const*)scalars, nullptr, out, counts);
  if (rc != FEC_OK) return rc;
  if (gathered && !aligned16(gathered)) return FEC_E_ARG;
  if (bases)
    for (size_t g = 0; g < ctx->children.size(); ++g)
      if (bases[g] && !aligned16(bases[g])) return FEC_E_ARG;
  return multi_dev_run(ctx, curve, counts, out, gathered, consumer, streams,
                       [&](fec_ctx* c, size_t g, size_t lo, size_t m, uint64_t* o, void* s) {
                         // no base given: the device's own copy of the reference's generator() (and its prefix table)
                         const uint64_t* base = bases && bases[g] ? bases[g] : c->d_gen[curve];
                         if (curve == FEC_ED25519)
                           return launch_ed_fixed(c, scalars[g] + lo * 4, base, base == c->d_gen[FEC_ED25519] ? c->h_gen_ed : nullptr, o, m, s);
                         return launch_mul(c, curve, true, scalars[g] + lo * 4, base, o, m, s);
                       });
} FEC_ABI_CATCH_STATUS

int fec_multi_batch_double_mul_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* const* u1, const uint64_t* const* u2,
                                   const uint64_t* const* q, uint64_t* const* out, const size_t* counts, uint64_t* gathered,
                                   int consumer, void* const* streams) try {
  int rc = multi_dev_args(ctx, curve, (const void* const*)u1, (const void* const*)u2, out, counts);
  if (rc != FEC_OK) return rc;
  if (!q || (gathered && !aligned16(gathered))) return FEC_E_ARG;
  for (size_t g = 0; g < ctx->children.size(); ++g)
    if (counts[g] && (!q[g] || !aligned16(q[g]))) return FEC_E_ARG;
  const size_t pl = (size_t)plimbs(curve);
  return multi_dev_run(ctx, curve, counts, out, gathered, consumer, streams,
                       [&](fec_ctx* c, size_t g, size_t lo, size_t m, uint64_t* o, void* s) {
                         return launch_double_mul(c, curve, u1[g] + lo * 4, u2[g] + lo * 4, q[g] + lo * pl, o, m, s);
                       });
} FEC_ABI_CATCH_STATUS

int fec_batch_mul(fec_ctx* ctx, fec_curve curve, const uint64_t* scalars, const uint64_t* points,
                  uint64_t* out, size_t n) try {
  if (is_multi(ctx)) {
    if (!curve_ok(curve) || (n && (!scalars || !points || !out))) return FEC_E_ARG;
    const size_t pl = (size_t)plimbs(curve);
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return fec_batch_mul(c, curve, scalars + lo * 4, points + lo * pl, out + lo * pl, cnt);
    });
  }
  if (!ctx || !curve_ok(curve) || (n && (!scalars || !points || !out))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  size_t pb = (size_t)plimbs(curve) * 8;
  const HostIn in[3] = {{scalars, 32, 0}, {points, pb, 0}, {nullptr, 0, 0}};
  return host_pipeline(ctx, n, in, out, pb, [&](void* a, void* b, void*, void* o, size_t cnt, void* s) {
    return launch_mul(ctx, curve, false, (const u64*)a, (const u64*)b, (u64*)o, cnt, s);
  });
} FEC_ABI_CATCH_STATUS

int fec_batch_mul_fixed(fec_ctx* ctx, fec_curve curve, const uint64_t* scalars, const uint64_t* base,
                        uint64_t* out, size_t n) try {
  if (is_multi(ctx)) {
    if (!curve_ok(curve) || !base || (n && (!scalars || !out))) return FEC_E_ARG;
    const size_t pl = (size_t)plimbs(curve);
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return fec_batch_mul_fixed(c, curve, scalars + lo * 4, base, out + lo * pl, cnt);
    });
  }
  if (!ctx || !curve_ok(curve) || !base || (n && (!scalars || !out))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  size_t pb = (size_t)plimbs(curve) * 8;
  if (curve == FEC_ED25519) {
    // the addend table is built once, on the ctx's first stream, before the pipeline starts
    if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
    int rc = drained(ctx, [&]() -> int {
      int r = ensure(ctx, 1, pb);
      if (r != FEC_OK) return r;
      if (hipMemcpyAsync(ctx->d_buf[1], base, pb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return FEC_E_DEVICE;
      r = ensure_ed_table(ctx, (const u64*)ctx->d_buf[1], base, ctx->stream);
      if (r != FEC_OK) return r;
      return hipStreamSynchronize(ctx->stream) == hipSuccess ? FEC_OK : FEC_E_LAUNCH;
    });
    if (rc != FEC_OK) return rc;
  }
  // the reference's generator() is recognised by value: the launches then name the ctx's own device copy, whose
  // prefix table (ensure_gen_prefix) they can start from
  const bool is_gen = std::memcmp(base, ctx->h_gen[curve], pb) == 0;
  const HostIn in[3] = {{scalars, 32, 0}, {base, 0, pb}, {nullptr, 0, 0}};
  return host_pipeline(ctx, n, in, out, pb, [&](void* a, void* b, void*, void* o, size_t cnt, void* s) {
    const u64* db = is_gen ? ctx->d_gen[curve] : (const u64*)b;
    if (curve == FEC_ED25519) return launch_ed_fixed(ctx, (const u64*)a, db, base, (u64*)o, cnt, s);
    return launch_mul(ctx, curve, true, (const u64*)a, db, (u64*)o, cnt, s);
  });
} FEC_ABI_CATCH_STATUS

int fec_batch_double_mul(fec_ctx* ctx, fec_curve curve, const uint64_t* u1, const uint64_t* u2,
                         const uint64_t* q, uint64_t* out, size_t n) try {
  if (is_multi(ctx)) {
    if (!curve_ok(curve) || (n && (!u1 || !u2 || !q || !out))) return FEC_E_ARG;
    const size_t pl = (size_t)plimbs(curve);
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return fec_batch_double_mul(c, curve, u1 + lo * 4, u2 + lo * 4, q + lo * pl, out + lo * pl, cnt);
    });
  }
  if (!ctx || !curve_ok(curve) || (n && (!u1 || !u2 || !q || !out))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  size_t pb = (size_t)plimbs(curve) * 8;
  const HostIn in[3] = {{u1, 32, 0}, {u2, 32, 0}, {q, pb, 0}};
  return host_pipeline(ctx, n, in, out, pb, [&](void* a, void* b, void* c, void* o, size_t cnt, void* s) {
    return launch_double_mul(ctx, curve, (const u64*)a, (const u64*)b, (const u64*)c, (u64*)o, cnt, s);
  });
} FEC_ABI_CATCH_STATUS

int fec_multi_scalar_mul(fec_ctx* ctx, fec_curve curve, const uint64_t* scalars, const uint64_t* points,
                         uint64_t* out, size_t n) try {
  FEC_FIRST_DEVICE(ctx);
  if (!ctx || !curve_ok(curve) || !out || (n && (!scalars || !points))) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  const size_t pb = (size_t)plimbs(curve) * 8;
  return drained(ctx, [&]() -> int {
  int rc = ensure(ctx, 6, (n ? n : 1) * pb);   // products stay on the device
  if (rc == FEC_OK) rc = ensure(ctx, 7, pb);
  if (rc != FEC_OK) return rc;
  const size_t msm_chunk = pipeline_chunk(ctx);
  for (size_t lo = 0; lo < n; lo += msm_chunk) {  // the independent products, chunked
    const size_t cnt = lo + msm_chunk <= n ? msm_chunk : n - lo;
    rc = ensure(ctx, 0, cnt * 32);
    if (rc == FEC_OK) rc = ensure(ctx, 1, cnt * pb);
    if (rc != FEC_OK) return rc;
    if (hipMemcpyAsync(ctx->d_buf[0], scalars + lo * 4, cnt * 32, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(ctx->d_buf[1], points + lo * (pb / 8), cnt * pb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
      return FEC_E_DEVICE;
    rc = launch_mul(ctx, curve, false, (const u64*)ctx->d_buf[0], (const u64*)ctx->d_buf[1],
                    (u64*)((char*)ctx->d_buf[6] + lo * pb), cnt, nullptr);
    if (rc != FEC_OK) return rc;
  }
  {
    Launch L(ctx, nullptr, "k_fold_sum");
    const u32* prod = reinterpret_cast<const u32*>(ctx->d_buf[6]);
    u32* o = reinterpret_cast<u32*>(ctx->d_buf[7]);
    switch (curve) {
      case FEC_SECP256K1: hipLaunchKernelGGL((k_fold_sum<Secp>), dim3(1), dim3(64), 0, L.s, prod, o, n); break;
      case FEC_P256: hipLaunchKernelGGL((k_fold_sum<P256>), dim3(1), dim3(64), 0, L.s, prod, o, n); break;
      default: hipLaunchKernelGGL((k_fold_sum<Ed>), dim3(1), dim3(64), 0, L.s, prod, o, n); break;
    }
    rc = L.done();
    if (rc != FEC_OK) return rc;
  }
  if (hipMemcpyAsync(out, ctx->d_buf[7], pb, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return FEC_E_DEVICE;
  return sync_and_check(ctx, ctx->stream);
  });
} FEC_ABI_CATCH_STATUS

namespace {
int ecdsa_verify_dev(fec_ctx* ctx, int curve, const uint8_t* d_digests, const uint64_t* d_r, const uint64_t* d_s,
                     const uint64_t* d_pk_xy, const uint8_t* d_pk_inf, uint8_t* d_status, size_t n, void* stream) {
  if (is_multi(ctx)) return FEC_E_UNSUPPORTED;  // device pointers belong to one device
  if (!ctx || (n && (!d_digests || !d_r || !d_s || !d_pk_xy || !d_status))) return FEC_E_ARG;
  if (!aligned16(d_digests) || !aligned16(d_r) || !aligned16(d_s) || !aligned16(d_pk_xy)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return launch_ecdsa_verify(ctx, curve, d_digests, d_r, d_s, d_pk_xy, d_pk_inf, d_status, n, stream);
}

int ecdsa_verify_host(fec_ctx* ctx, int curve, const uint8_t* digests, const uint64_t* r, const uint64_t* s,
                      const uint64_t* pk_xy, const uint8_t* pk_inf, uint8_t* status, size_t n) {
  if (is_multi(ctx)) {
    if (n && (!digests || !r || !s || !pk_xy || !status)) return FEC_E_ARG;
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return ecdsa_verify_host(c, curve, digests + lo * 32, r + lo * 4, s + lo * 4, pk_xy + lo * 8,
                               pk_inf ? pk_inf + lo : nullptr, status + lo, cnt);
    });
  }
  if (!ctx || (n && (!digests || !r || !s || !pk_xy || !status))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return drained(ctx, [&]() -> int {
  const void* hin[5] = {digests, r, s, pk_xy, pk_inf};
  const size_t bytes[5] = {n * 32, n * 32, n * 32, n * 64, n};
  const int slot[5] = {0, 1, 2, 4, 5};
  for (int i = 0; i < 5; ++i) {
    if (!hin[i]) continue;
    int rc = ensure(ctx, slot[i], bytes[i]);
    if (rc != FEC_OK) return rc;
    if (hipMemcpyAsync(ctx->d_buf[slot[i]], hin[i], bytes[i], hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
      return FEC_E_DEVICE;
  }
  int rc = ensure(ctx, 3, n);
  if (rc != FEC_OK) return rc;
  rc = launch_ecdsa_verify(ctx, curve, (const unsigned char*)ctx->d_buf[0], (const u64*)ctx->d_buf[1],
                           (const u64*)ctx->d_buf[2], (const u64*)ctx->d_buf[4],
                           pk_inf ? (const unsigned char*)ctx->d_buf[5] : nullptr, (unsigned char*)ctx->d_buf[3], n, nullptr);
  if (rc != FEC_OK) return rc;
  if (hipMemcpyAsync(status, ctx->d_buf[3], n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return FEC_E_DEVICE;
  return sync_and_check(ctx, ctx->stream);
  });
}
}  // namespace

int fec_ecdsa_verify_secp256k1_dev(fec_ctx* ctx, const uint8_t* d_digests, const uint64_t* d_r, const uint64_t* d_s,
                                   const uint64_t* d_pk_xy, const uint8_t* d_pk_inf, uint8_t* d_status, size_t n,
                                   void* stream) try {
  return ecdsa_verify_dev(ctx, FEC_SECP256K1, d_digests, d_r, d_s, d_pk_xy, d_pk_inf, d_status, n, stream);
} FEC_ABI_CATCH_STATUS
int fec_ecdsa_verify_secp256k1(fec_ctx* ctx, const uint8_t* digests, const uint64_t* r, const uint64_t* s,
                               const uint64_t* pk_xy, const uint8_t* pk_inf, uint8_t* status, size_t n) try {
  return ecdsa_verify_host(ctx, FEC_SECP256K1, digests, r, s, pk_xy, pk_inf, status, n);
} FEC_ABI_CATCH_STATUS
int fec_ecdsa_verify_p256_dev(fec_ctx* ctx, const uint8_t* d_digests, const uint64_t* d_r, const uint64_t* d_s,
                              const uint64_t* d_pk_xy, const uint8_t* d_pk_inf, uint8_t* d_status, size_t n,
                              void* stream) try {
  return ecdsa_verify_dev(ctx, FEC_P256, d_digests, d_r, d_s, d_pk_xy, d_pk_inf, d_status, n, stream);
} FEC_ABI_CATCH_STATUS
int fec_ecdsa_verify_p256(fec_ctx* ctx, const uint8_t* digests, const uint64_t* r, const uint64_t* s,
                          const uint64_t* pk_xy, const uint8_t* pk_inf, uint8_t* status, size_t n) try {
  return ecdsa_verify_host(ctx, FEC_P256, digests, r, s, pk_xy, pk_inf, status, n);
} FEC_ABI_CATCH_STATUS

// Ecdsa::<C, D>::batch_verify (forge-ec-signature/src/ecdsa.rs:287-391), C = Secp256k1 / P256, with the digests
// and the weights a_i (302-306) supplied.  The per-signature scalars and the 2n multiplications run in
// parallel; the loop's early returns (first failing signature in index order), the ORDERED fold
// r_sum += r_i (358) and the ordered scalar sum (368-372) are reproduced exactly.
int fec_ecdsa_batch_verify(fec_ctx* ctx, fec_curve curve, const uint8_t* digests, const uint64_t* r, const uint64_t* s,
                           const uint64_t* pk_xy, const uint8_t* pk_inf, const uint64_t* a, size_t n, uint8_t* result,
                           uint64_t* detail) try {
  FEC_FIRST_DEVICE(ctx);
  if (!ctx || !result || (n && (!digests || !r || !s || !pk_xy || !a))) return FEC_E_ARG;
  if (curve != FEC_SECP256K1 && curve != FEC_P256) return FEC_E_UNSUPPORTED;
  *result = 0;
  if (detail) std::memset(detail, 0, 16 * sizeof(uint64_t));
  if (n == 0) return FEC_OK;                                   // 289-291: false
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  // (the host targets of the device-to-host copies live outside the drained scope: a copy queued before a failure is
  // waited for while they still exist)
  std::unique_ptr<unsigned char[]> flags(new (std::nothrow) unsigned char[n]);  // no exception may cross the C ABI
  if (!flags) return FEC_E_OOM;
  unsigned char res = 0;
  uint64_t det[16];
  return drained(ctx, [&]() -> int {
  // slots: 0 digests, 1 r, 2 s, 3 pk, 4 a, 5 pk_inf, 6 work area, 7 r_sum + detail + result
  const void* hin[6] = {digests, r, s, pk_xy, a, pk_inf};
  const size_t bytes[6] = {n * 32, n * 32, n * 32, n * 64, n * 32, n};
  int rc = FEC_OK;
  for (int i = 0; i < 6 && rc == FEC_OK; ++i)
    if (hin[i]) rc = ensure(ctx, i, bytes[i]);
  if (rc == FEC_OK) rc = ensure(ctx, 6, ecdsa_batch_work_bytes(n));
  if (rc == FEC_OK) rc = ensure(ctx, 7, 96 + 128 + 16);
  if (rc != FEC_OK) return rc;
  for (int i = 0; i < 6; ++i)
    if (hin[i] && hipMemcpyAsync(ctx->d_buf[i], hin[i], bytes[i], hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
      return FEC_E_DEVICE;
  char* work = static_cast<char*>(ctx->d_buf[6]);
  char* tail = static_cast<char*>(ctx->d_buf[7]);
  {
    Launch L(ctx, nullptr, "k_ecdsa_pre");
    ecdsa_batch_pre_launch(curve, (const unsigned char*)ctx->d_buf[0], (const u32*)ctx->d_buf[1], (const u32*)ctx->d_buf[2],
                           (const u32*)ctx->d_buf[3], pk_inf ? (const unsigned char*)ctx->d_buf[5] : nullptr,
                           (const u32*)ctx->d_buf[4], work, n, L.s);
    rc = L.done();
    if (rc != FEC_OK) return rc;
  }
  // the loop returns at the first signature that fails a check (317-342): nothing after it is computed
  if (hipMemcpyAsync(flags.get(), work + n * 352, n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return FEC_E_DEVICE;
  if (int rc_sync = sync_and_check(ctx, ctx->stream)) return rc_sync;
  for (size_t i = 0; i < n; ++i)
    if (flags[i] != 0) {
      *result = flags[i] == 2 ? 2 : 0;
      return FEC_OK;
    }
  u32* ta = reinterpret_cast<u32*>(work + n * 160);
  u32* tb = reinterpret_cast<u32*>(work + n * 256);
  ensure_gen_prefix(ctx, curve, ctx->stream, n);
  {
    Launch L(ctx, nullptr, curve == FEC_SECP256K1 ? "k_secp_mul x2 + k_point_op + k_fold_sum + k_ecdsa_batch_finish"
                                                  : "k_p256_mul_sched x2 + k_point_op + k_fold_sum + k_ecdsa_batch_finish");
    ecdsa_batch_mul_launch(sched_env(ctx), curve, reinterpret_cast<const u32*>(ctx->d_gen[curve]), work, n, L.s, ctx->stream2);
    const dim3 g(grid_for(n)), b(TPB);
    if (curve == FEC_SECP256K1) {  // r_i = r1 + r2 (355), then r_sum += r_i in index order (358)
      hipLaunchKernelGGL((k_point_op<Secp>), g, b, 0, L.s, (int)FEC_P_ADD, (const u32*)ta, (const u32*)tb, ta, n);
      hipLaunchKernelGGL((k_fold_sum<Secp>), dim3(1), dim3(64), 0, L.s, (const u32*)ta, (u32*)tail, n);
    } else {
      hipLaunchKernelGGL((k_point_op<P256>), g, b, 0, L.s, (int)FEC_P_ADD, (const u32*)ta, (const u32*)tb, ta, n);
      hipLaunchKernelGGL((k_fold_sum<P256>), dim3(1), dim3(64), 0, L.s, (const u32*)ta, (u32*)tail, n);
    }
    ecdsa_batch_finish_launch(curve, (const u32*)tail, work, n, (unsigned char*)(tail + 96 + 128), (u32*)(tail + 96), L.s);
    rc = L.done();
    if (rc != FEC_OK) return rc;
  }
  if (hipMemcpyAsync(&res, tail + 96 + 128, 1, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(det, tail + 96, 128, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
    return FEC_E_DEVICE;
  if (int rc_sync = sync_and_check(ctx, ctx->stream)) return rc_sync;
  *result = res;
  if (detail) std::memcpy(detail, det, 128);
  return FEC_OK;
  });
} FEC_ABI_CATCH_STATUS

int fec_batch_validate_point_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_xy, const uint8_t* d_inf, uint8_t* d_ok,
                                 size_t n, void* stream) try {
  if (is_multi(ctx)) return FEC_E_UNSUPPORTED;  // device pointers belong to one device
  if (!ctx || !curve_ok(curve) || (n && (!d_xy || !d_ok))) return FEC_E_ARG;
  if (!aligned16(d_xy)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return launch_validate(ctx, curve, d_xy, d_inf, d_ok, n, stream);
} FEC_ABI_CATCH_STATUS

int fec_batch_validate_point(fec_ctx* ctx, fec_curve curve, const uint64_t* xy, const uint8_t* inf, uint8_t* ok, size_t n) try {
  if (!curve_ok(curve)) return FEC_E_ARG;
  if (is_multi(ctx)) {
    if (n && (!xy || !ok)) return FEC_E_ARG;
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return fec_batch_validate_point(c, curve, xy + lo * 8, inf ? inf + lo : nullptr, ok + lo, cnt);
    });
  }
  if (!ctx || (n && (!xy || !ok))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return drained(ctx, [&]() -> int {
  int rc = ensure(ctx, 0, n * 64);
  if (rc == FEC_OK && inf) rc = ensure(ctx, 1, n);
  if (rc == FEC_OK) rc = ensure(ctx, 2, n);
  if (rc != FEC_OK) return rc;
  if (hipMemcpyAsync(ctx->d_buf[0], xy, n * 64, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return FEC_E_DEVICE;
  if (inf && hipMemcpyAsync(ctx->d_buf[1], inf, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return FEC_E_DEVICE;
  rc = launch_validate(ctx, curve, (const u64*)ctx->d_buf[0], inf ? (const unsigned char*)ctx->d_buf[1] : nullptr,
                       (unsigned char*)ctx->d_buf[2], n, nullptr);
  if (rc != FEC_OK) return rc;
  if (hipMemcpyAsync(ok, ctx->d_buf[2], n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return FEC_E_DEVICE;
  return sync_and_check(ctx, ctx->stream);
  });
} FEC_ABI_CATCH_STATUS

int fec_batch_ecdh_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_private_keys, const uint64_t* d_pk_xy,
                       const uint8_t* d_pk_inf, uint8_t* d_secrets, uint8_t* d_status, size_t n, void* stream) try {
  if (is_multi(ctx)) return FEC_E_UNSUPPORTED;  // device pointers belong to one device
  if (!ctx || (n && (!d_private_keys || !d_pk_xy || !d_secrets || !d_status))) return FEC_E_ARG;
  if (curve != FEC_SECP256K1 && curve != FEC_P256) return FEC_E_UNSUPPORTED;   // Ed25519 has no KeyExchange impl
  if (!aligned16(d_private_keys) || !aligned16(d_pk_xy) || !aligned16(d_secrets)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return launch_ecdh(ctx, curve, d_private_keys, d_pk_xy, d_pk_inf, d_secrets, d_status, n, stream);
} FEC_ABI_CATCH_STATUS

int fec_batch_ecdh(fec_ctx* ctx, fec_curve curve, const uint64_t* private_keys, const uint64_t* pk_xy, const uint8_t* pk_inf,
                   uint8_t* secrets, uint8_t* status, size_t n) try {
  if (curve != FEC_SECP256K1 && curve != FEC_P256) return FEC_E_UNSUPPORTED;
  if (is_multi(ctx)) {
    if (n && (!private_keys || !pk_xy || !secrets || !status)) return FEC_E_ARG;
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return fec_batch_ecdh(c, curve, private_keys + lo * 4, pk_xy + lo * 8, pk_inf ? pk_inf + lo : nullptr, secrets + lo * 32,
                            status + lo, cnt);
    });
  }
  if (!ctx || (n && (!private_keys || !pk_xy || !secrets || !status))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  // Chunked like the other element-wise calls (device staging bounded by one chunk).  The private keys (slot 0), the
  // secrets (slot 4) and the shared points (the stream's scratch) sit in ctx-owned device memory while the call runs:
  // they are cleared on EVERY way out of it, error returns included.
  struct Wipe {
    fec_ctx* c;
    ~Wipe() {
      if (c->d_buf[0]) (void)hipMemsetAsync(c->d_buf[0], 0, c->d_cap[0], c->stream);
      if (c->d_buf[4]) (void)hipMemsetAsync(c->d_buf[4], 0, c->d_cap[4], c->stream);
      for (auto& e : c->stream_scratch)
        if (e.stream == c->stream && e.buf) (void)hipMemsetAsync(e.buf, 0, e.cap, c->stream);
      (void)hipStreamSynchronize(c->stream);
      (void)hipGetLastError();
    }
  } wipe{ctx};
  const void* const in[4] = {private_keys, pk_xy, pk_inf, nullptr};
  const size_t in_stride[4] = {32, 64, 1, 0};
  void* const outs[2] = {secrets, status};
  const size_t out_stride[2] = {32, 1};
  return host_chunked(ctx, n, in, in_stride, outs, out_stride, [&](void* const d[4], void* const o[2], size_t cnt) {
    return launch_ecdh(ctx, curve, (const u64*)d[0], (const u64*)d[1], (const unsigned char*)d[2], (unsigned char*)o[0],
                       (unsigned char*)o[1], cnt, nullptr);
  });
} FEC_ABI_CATCH_STATUS

int fec_eddsa_verify_ed25519_dev(fec_ctx* ctx, const uint64_t* d_r_xy, const uint8_t* d_r_inf, const uint64_t* d_pk_xy,
                                 const uint8_t* d_pk_inf, const uint64_t* d_s, const uint64_t* d_k, uint8_t* d_status,
                                 size_t n, void* stream) try {
  if (is_multi(ctx)) return FEC_E_UNSUPPORTED;  // device pointers belong to one device
  if (!ctx || (n && (!d_r_xy || !d_pk_xy || !d_s || !d_k || !d_status))) return FEC_E_ARG;
  if (!aligned16(d_r_xy) || !aligned16(d_pk_xy) || !aligned16(d_s) || !aligned16(d_k)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return launch_eddsa_verify(ctx, d_r_xy, d_r_inf, d_pk_xy, d_pk_inf, d_s, d_k, d_status, n, stream);
} FEC_ABI_CATCH_STATUS

int fec_eddsa_verify_ed25519(fec_ctx* ctx, const uint64_t* r_xy, const uint8_t* r_inf, const uint64_t* pk_xy,
                             const uint8_t* pk_inf, const uint64_t* s, const uint64_t* k, uint8_t* status, size_t n) try {
  if (is_multi(ctx)) {
    if (n && (!r_xy || !pk_xy || !s || !k || !status)) return FEC_E_ARG;
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return fec_eddsa_verify_ed25519(c, r_xy + lo * 8, r_inf ? r_inf + lo : nullptr, pk_xy + lo * 8,
                                      pk_inf ? pk_inf + lo : nullptr, s + lo * 4, k + lo * 4, status + lo, cnt);
    });
  }
  if (!ctx || (n && (!r_xy || !pk_xy || !s || !k || !status))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return drained(ctx, [&]() -> int {
  const void* hin[6] = {r_xy, pk_xy, s, k, r_inf, pk_inf};
  const size_t bytes[6] = {n * 64, n * 64, n * 32, n * 32, n, n};
  const int slot[6] = {0, 1, 2, 4, 5, 6};
  for (int i = 0; i < 6; ++i) {
    if (!hin[i]) continue;
    int rc = ensure(ctx, slot[i], bytes[i]);
    if (rc != FEC_OK) return rc;
    if (hipMemcpyAsync(ctx->d_buf[slot[i]], hin[i], bytes[i], hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
      return FEC_E_DEVICE;
  }
  int rc = ensure(ctx, 3, n);
  if (rc != FEC_OK) return rc;
  rc = launch_eddsa_verify(ctx, (const u64*)ctx->d_buf[0], r_inf ? (const unsigned char*)ctx->d_buf[5] : nullptr,
                           (const u64*)ctx->d_buf[1], pk_inf ? (const unsigned char*)ctx->d_buf[6] : nullptr,
                           (const u64*)ctx->d_buf[2], (const u64*)ctx->d_buf[4], (unsigned char*)ctx->d_buf[3], n, nullptr);
  if (rc != FEC_OK) return rc;
  if (hipMemcpyAsync(status, ctx->d_buf[3], n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return FEC_E_DEVICE;
  return sync_and_check(ctx, ctx->stream);
  });
} FEC_ABI_CATCH_STATUS

int fec_schnorr_verify_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_pk_xy, const uint8_t* d_pk_inf,
                           const uint64_t* d_r_xy, const uint8_t* d_r_inf, const uint64_t* d_s, const uint64_t* d_e,
                           uint8_t* d_status, size_t n, void* stream) try {
  if (is_multi(ctx)) return FEC_E_UNSUPPORTED;  // device pointers belong to one device
  if (!ctx || !curve_ok(curve) || (n && (!d_pk_xy || !d_r_xy || !d_s || !d_e || !d_status))) return FEC_E_ARG;
  if (!aligned16(d_pk_xy) || !aligned16(d_r_xy) || !aligned16(d_s) || !aligned16(d_e)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return launch_schnorr_verify(ctx, curve, d_pk_xy, d_pk_inf, d_r_xy, d_r_inf, d_s, d_e, d_status, n, stream);
} FEC_ABI_CATCH_STATUS

int fec_schnorr_verify(fec_ctx* ctx, fec_curve curve, const uint64_t* pk_xy, const uint8_t* pk_inf, const uint64_t* r_xy,
                       const uint8_t* r_inf, const uint64_t* s, const uint64_t* e, uint8_t* status, size_t n) try {
  if (!curve_ok(curve)) return FEC_E_ARG;
  if (is_multi(ctx)) {
    if (n && (!pk_xy || !r_xy || !s || !e || !status)) return FEC_E_ARG;
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return fec_schnorr_verify(c, curve, pk_xy + lo * 8, pk_inf ? pk_inf + lo : nullptr, r_xy + lo * 8,
                                r_inf ? r_inf + lo : nullptr, s + lo * 4, e + lo * 4, status + lo, cnt);
    });
  }
  if (!ctx || (n && (!pk_xy || !r_xy || !s || !e || !status))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  // chunked like the other element-wise calls; slots: 0 pk, 1 r, 2 s, 4 e, 5 pk_inf, 6 r_inf, 3 status
  const size_t pc = pipeline_chunk(ctx);
  const size_t chunk = pc < n ? pc : n;
  return drained(ctx, [&]() -> int {
  for (size_t lo = 0; lo < n; lo += chunk) {
    const size_t cnt = lo + chunk <= n ? chunk : n - lo;
    const void* hin[6] = {pk_xy + lo * 8, r_xy + lo * 8, s + lo * 4, e + lo * 4, pk_inf ? pk_inf + lo : nullptr,
                          r_inf ? r_inf + lo : nullptr};
    const size_t bytes[6] = {cnt * 64, cnt * 64, cnt * 32, cnt * 32, cnt, cnt};
    const int slot[6] = {0, 1, 2, 4, 5, 6};
    for (int i = 0; i < 6; ++i) {
      if (!hin[i]) continue;
      int rc = ensure(ctx, slot[i], bytes[i]);
      if (rc != FEC_OK) return rc;
      if (hipMemcpyAsync(ctx->d_buf[slot[i]], hin[i], bytes[i], hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return FEC_E_DEVICE;
    }
    int rc = ensure(ctx, 3, cnt);
    if (rc != FEC_OK) return rc;
    rc = launch_schnorr_verify(ctx, curve, (const u64*)ctx->d_buf[0], pk_inf ? (const unsigned char*)ctx->d_buf[5] : nullptr,
                               (const u64*)ctx->d_buf[1], r_inf ? (const unsigned char*)ctx->d_buf[6] : nullptr,
                               (const u64*)ctx->d_buf[2], (const u64*)ctx->d_buf[4], (unsigned char*)ctx->d_buf[3], cnt, nullptr);
    if (rc != FEC_OK) return rc;
    if (hipMemcpyAsync(status + lo, ctx->d_buf[3], cnt, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return FEC_E_DEVICE;
    if (int rc_sync = sync_and_check(ctx, ctx->stream)) return rc_sync;
  }
  return FEC_OK;
  });
} FEC_ABI_CATCH_STATUS

namespace {
// schnorr::batch_verify::<C, D> (forge-ec-signature/src/schnorr.rs:194-290) for C = Secp256k1 / P256
int schnorr_batch_verify(fec_ctx* ctx, int curve, const uint64_t* pk_xy, const uint8_t* pk_inf, const uint64_t* r_xy,
                         const uint8_t* r_inf, const uint64_t* s, const uint64_t* a, const uint64_t* e, size_t n,
                         uint8_t* result, uint64_t* sides_xy, uint8_t* sides_inf, uint8_t* debug_build_panics = nullptr) {
  FEC_FIRST_DEVICE(ctx);
  if (!ctx || !result || !curve_ok(curve) || (n && (!pk_xy || !r_xy || !s || !a || !e))) return FEC_E_ARG;
  *result = 0;
  if (debug_build_panics) *debug_build_panics = 0;
  if (sides_xy) std::memset(sides_xy, 0, 16 * sizeof(uint64_t));
  if (sides_inf) sides_inf[0] = sides_inf[1] = 0;
  if (n == 0) return FEC_OK;                                   // 197-199
  for (size_t i = 0; i < n; ++i)                               // 204-225 (only the identity tests can reject)
    if ((pk_inf && pk_inf[i]) || (r_inf && r_inf[i])) return FEC_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  const bool secp = curve == FEC_SECP256K1, edw = curve == FEC_ED25519;
  const size_t pb = (size_t)plimbs(curve) * 8;
  unsigned char flags[8] = {0};  // (host targets of the last copies: outside the drained scope, see fec_ecdsa_batch_verify)
  uint64_t sides[16];
  return drained(ctx, [&]() -> int {
  // slots: 0 pk, 1 r, 2 s, 3 a, 4 e, 5 A terms, 6 B terms, 7 sums + affine sides + flags + counter
  const size_t bytes[5] = {n * 64, n * 64, n * 32, n * 32, n * 32};
  const void* src[5] = {pk_xy, r_xy, s, a, e};
  int rc = FEC_OK;
  for (int i = 0; i < 5 && rc == FEC_OK; ++i) rc = ensure(ctx, i, bytes[i]);
  if (rc == FEC_OK) rc = ensure(ctx, 5, n * pb);
  if (rc == FEC_OK) rc = ensure(ctx, 6, n * pb);
  if (rc == FEC_OK) rc = ensure(ctx, 7, 2 * pb + 128 + 32);
  if (rc != FEC_OK) return rc;
  for (int i = 0; i < 5; ++i)
    if (hipMemcpyAsync(ctx->d_buf[i], src[i], bytes[i], hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
      return FEC_E_DEVICE;
  char* tail = (char*)ctx->d_buf[7];
  u32* d_sums = (u32*)tail;
  u32* d_sides = (u32*)(tail + 2 * pb);
  unsigned char* d_flags = (unsigned char*)(tail + 2 * pb + 128);
  unsigned int* d_done = (unsigned int*)(tail + 2 * pb + 128 + 8);
  unsigned char* d_wrapped = (unsigned char*)(tail + 2 * pb + 128 + 16);   // Ed25519: some s_i * a_i wrapped a u128 sum
  if (hipMemsetAsync(tail + 2 * pb + 128, 0, 32, ctx->stream) != hipSuccess) return FEC_E_DEVICE;
  if (edw) {
    // Ed25519 (the generic batch_verify::<Ed25519, D>): the same pipeline with that curve's kernels -- multiply(G, .) by
    // the table kernel, the two variable-base products by the scheduler kernel -- one after the other on the ctx stream
    // work area: s*a (32 n), from_affine(P) (128 n), e*P (128 n), R + e*P (128 n)
    char* work = static_cast<char*>(scratch_for(ctx, ctx->stream, n * (32 + 3 * pb)));
    if (!work) return FEC_E_OOM;
    u32* sa = reinterpret_cast<u32*>(work);
    u32* pp = reinterpret_cast<u32*>(work + n * 32);
    u32* ep = reinterpret_cast<u32*>(work + n * (32 + pb));
    u32* qq = reinterpret_cast<u32*>(work + n * (32 + 2 * pb));
    const dim3 g(grid_for(n)), b(TPB);
    {
      Launch L(ctx, nullptr, "k_schnorr_pre");
      hipLaunchKernelGGL((k_schnorr_pre<Ed>), g, b, 0, L.s, (const u32*)ctx->d_buf[0], (const u32*)ctx->d_buf[2], (const u32*)ctx->d_buf[3], sa, pp, d_wrapped, n);
      rc = L.done();
      if (rc != FEC_OK) return rc;
    }
    rc = launch_ed_fixed(ctx, (const u64*)sa, ctx->d_gen[FEC_ED25519], ctx->h_gen_ed, (u64*)ctx->d_buf[5], n, nullptr);   // A_i (266-268)
    if (rc != FEC_OK) return rc;
    {
      Launch L(ctx, nullptr, "k_schnorr_pre + k_ed_fixed_base + k_ed_mul_pers x2 + k_schnorr_mid");
      const SchedEnv env = sched_env(ctx);
      ed_launch_mul(env, (const u32*)ctx->d_buf[4], pp, ep, n, L.s);                                                        // e_i P_i (276)
      hipLaunchKernelGGL((k_schnorr_mid<Ed>), g, b, 0, L.s, (const u32*)ctx->d_buf[1], (const u32*)ep, qq, n);
      ed_launch_mul(env, (const u32*)ctx->d_buf[3], qq, (u32*)ctx->d_buf[6], n, L.s);                                       // B_i (282)
      rc = L.done();
      if (rc != FEC_OK) return rc;
    }
  } else {
    // work area: s*a (32 n), from_affine(P) (96 n), e*P (96 n), R + e*P (96 n)
    char* work = static_cast<char*>(scratch_for(ctx, ctx->stream, n * 320));
    if (!work) return FEC_E_OOM;
    u32* sa = reinterpret_cast<u32*>(work);
    u32* pp = reinterpret_cast<u32*>(work + n * 32);
    u32* ep = reinterpret_cast<u32*>(work + n * 128);
    u32* qq = reinterpret_cast<u32*>(work + n * 224);
    const u32* gen = reinterpret_cast<const u32*>(ctx->d_gen[curve]);
    ensure_gen_prefix(ctx, curve, ctx->stream, n);
    const SchedEnv env = sched_env(ctx);
    Launch L(ctx, nullptr, secp ? "k_schnorr_pre + k_secp_mul x3 + k_schnorr_mid" : "k_schnorr_pre + k_p256_mul_sched x3 + k_schnorr_mid");
    const dim3 g(grid_for(n)), b(TPB);
    auto mul = [&](bool fixed, const u32* k, const u32* p, u32* o, hipStream_t st, const SchedEnv& e) {
      if (secp) secp_launch_mul(e, fixed, k, p, o, n, st);
      else p256_launch_mul(e, fixed, k, p, o, n, st);
    };
    if (secp) hipLaunchKernelGGL((k_schnorr_pre<Secp>), g, b, 0, L.s, (const u32*)ctx->d_buf[0], (const u32*)ctx->d_buf[2], (const u32*)ctx->d_buf[3], sa, pp, (unsigned char*)nullptr, n);
    else hipLaunchKernelGGL((k_schnorr_pre<P256>), g, b, 0, L.s, (const u32*)ctx->d_buf[0], (const u32*)ctx->d_buf[2], (const u32*)ctx->d_buf[3], sa, pp, (unsigned char*)nullptr, n);
    // the A terms do not depend on the B chain: they run on the ctx's second stream beside it (at the moderate n
    // this entry point is meant for, a launch fills a fraction of the chip and is bound by one ladder's latency)
    hipEvent_t ev_pre = nullptr, ev_a = nullptr;
    const bool side = ctx->stream2 != nullptr && hipEventCreateWithFlags(&ev_pre, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&ev_a, hipEventDisableTiming) == hipSuccess;
    hipStream_t sa_stream = L.s;
    if (side) {
      (void)hipEventRecord(ev_pre, L.s);
      (void)hipStreamWaitEvent(ctx->stream2, ev_pre, 0);
      sa_stream = ctx->stream2;
    }
    // the persistent P-256 kernels: the CUs in proportion to the work on either stream (one fixed-base launch beside
    // two variable-base ones) while they run side by side
    SchedEnv ef = env, ev = env;
    if (side && !secp) p256_cu_split(env, n, kP256VarAffineMs + kP256VarMs, ef, ev);
    mul(true, sa, gen, (u32*)ctx->d_buf[5], sa_stream, ef);                              // A_i (266-268)
    if (side) (void)hipEventRecord(ev_a, ctx->stream2);
    mul(false, (const u32*)ctx->d_buf[4], pp, ep, L.s, ev);                              // e_i P_i (276)
    if (secp) hipLaunchKernelGGL((k_schnorr_mid<Secp>), g, b, 0, L.s, (const u32*)ctx->d_buf[1], (const u32*)ep, qq, n);
    else hipLaunchKernelGGL((k_schnorr_mid<P256>), g, b, 0, L.s, (const u32*)ctx->d_buf[1], (const u32*)ep, qq, n);
    mul(false, (const u32*)ctx->d_buf[3], qq, (u32*)ctx->d_buf[6], L.s, ev);             // B_i (282)
    if (side) (void)hipStreamWaitEvent(L.s, ev_a, 0);
    rc = L.done();
    if (ev_pre) (void)hipEventDestroy(ev_pre);
    if (ev_a) (void)hipEventDestroy(ev_a);
    if (rc != FEC_OK) return rc;
  }
  if (secp) hipLaunchKernelGGL((k_schnorr_fold_compare<Secp>), dim3(2), dim3(64), 0, ctx->stream, (const u32*)ctx->d_buf[5], (const u32*)ctx->d_buf[6], d_sums, d_sides, d_flags, d_done, n);
  else if (edw) hipLaunchKernelGGL((k_schnorr_fold_compare<Ed>), dim3(2), dim3(64), 0, ctx->stream, (const u32*)ctx->d_buf[5], (const u32*)ctx->d_buf[6], d_sums, d_sides, d_flags, d_done, n);
  else hipLaunchKernelGGL((k_schnorr_fold_compare<P256>), dim3(2), dim3(64), 0, ctx->stream, (const u32*)ctx->d_buf[5], (const u32*)ctx->d_buf[6], d_sums, d_sides, d_flags, d_done, n);
  if (hipGetLastError() != hipSuccess) return FEC_E_LAUNCH;
  if (hipMemcpyAsync(flags + 4, d_wrapped, 1, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return FEC_E_DEVICE;
  if (hipMemcpyAsync(flags, d_flags, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(sides, d_sides, 128, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
    return FEC_E_DEVICE;
  if (int rc_sync = sync_and_check(ctx, ctx->stream)) return rc_sync;
  *result = flags[0];
  if (debug_build_panics) *debug_build_panics = flags[4];
  if (flags[0] == 2) return FEC_OK;   // (the reference panics in to_affine: no sides)
  if (sides_xy) std::memcpy(sides_xy, sides, 128);
  if (sides_inf) { sides_inf[0] = flags[1]; sides_inf[1] = flags[2]; }
  return FEC_OK;
  });
}
}  // namespace

int fec_schnorr_batch_verify_secp256k1(fec_ctx* ctx, const uint64_t* pk_xy, const uint8_t* pk_inf,
                                       const uint64_t* r_xy, const uint8_t* r_inf, const uint64_t* s,
                                       const uint64_t* a, const uint64_t* e, size_t n, uint8_t* result,
                                       uint64_t* sides_xy, uint8_t* sides_inf) try {
  return schnorr_batch_verify(ctx, FEC_SECP256K1, pk_xy, pk_inf, r_xy, r_inf, s, a, e, n, result, sides_xy, sides_inf);
} FEC_ABI_CATCH_STATUS

int fec_schnorr_batch_verify(fec_ctx* ctx, fec_curve curve, const uint64_t* pk_xy, const uint8_t* pk_inf,
                             const uint64_t* r_xy, const uint8_t* r_inf, const uint64_t* s, const uint64_t* a,
                             const uint64_t* e, size_t n, uint8_t* result, uint64_t* sides_xy, uint8_t* sides_inf) try {
  return schnorr_batch_verify(ctx, curve, pk_xy, pk_inf, r_xy, r_inf, s, a, e, n, result, sides_xy, sides_inf);
} FEC_ABI_CATCH_STATUS

int fec_schnorr_batch_verify_ed25519(fec_ctx* ctx, const uint64_t* pk_xy, const uint8_t* pk_inf, const uint64_t* r_xy,
                                     const uint8_t* r_inf, const uint64_t* s, const uint64_t* a, const uint64_t* e, size_t n,
                                     uint8_t* result, uint64_t* sides_xy, uint8_t* sides_inf, uint8_t* debug_build_panics) try {
  return schnorr_batch_verify(ctx, FEC_ED25519, pk_xy, pk_inf, r_xy, r_inf, s, a, e, n, result, sides_xy, sides_inf,
                              debug_build_panics);
} FEC_ABI_CATCH_STATUS

int fec_batch_compress_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_xy, const uint8_t* d_inf,
                           uint8_t* d_out, size_t n, void* stream) try {
  if (is_multi(ctx)) return FEC_E_UNSUPPORTED;  // device pointers belong to one device
  if (!ctx || !curve_ok(curve) || (n && (!d_xy || !d_out))) return FEC_E_ARG;
  if (!aligned16(d_xy) || (reinterpret_cast<uintptr_t>(d_out) & 3u)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return launch_compress(ctx, curve, d_xy, d_inf, d_out, n, stream);
} FEC_ABI_CATCH_STATUS

int fec_batch_compress(fec_ctx* ctx, fec_curve curve, const uint64_t* xy, const uint8_t* inf, uint8_t* out,
                       size_t n) try {
  if (is_multi(ctx)) {
    if (!curve_ok(curve) || (n && (!xy || !out))) return FEC_E_ARG;
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return fec_batch_compress(c, curve, xy + lo * 8, inf ? inf + lo : nullptr, out + lo * 33, cnt);
    });
  }
  if (!ctx || !curve_ok(curve) || (n && (!xy || !out))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  const void* const in[4] = {xy, inf, nullptr, nullptr};
  const size_t in_stride[4] = {64, 1, 0, 0};
  void* const outs[2] = {out, nullptr};
  const size_t out_stride[2] = {33, 0};
  return host_chunked(ctx, n, in, in_stride, outs, out_stride, [&](void* const d[4], void* const o[2], size_t cnt) {
    return launch_compress(ctx, curve, (const u64*)d[0], (const unsigned char*)d[1], (unsigned char*)o[0], cnt, nullptr);
  });
} FEC_ABI_CATCH_STATUS

// decode entry points: in -> (xy, inf, ok).  The device writes inf and ok into one staging area
// (inf at [0, cnt), ok at [chunk, chunk + cnt)), so host_chunked's two output slots suffice.
static int decode_host(fec_ctx* ctx, int op, fec_curve curve, const uint8_t* in, size_t in_stride, uint64_t* xy,
                       uint8_t* inf, uint8_t* ok, size_t n) {
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  const size_t chunk = ctx->chunk < n ? ctx->chunk : n;
  return drained(ctx, [&]() -> int {
  for (size_t lo = 0; lo < n; lo += chunk) {
    const size_t cnt = lo + chunk <= n ? chunk : n - lo;
    int rc = ensure(ctx, 0, chunk * in_stride);
    if (rc == FEC_OK) rc = ensure(ctx, 4, chunk * 64);
    if (rc == FEC_OK) rc = ensure(ctx, 5, chunk * 2);
    if (rc != FEC_OK) return rc;
    if (hipMemcpyAsync(ctx->d_buf[0], in + lo * in_stride, cnt * in_stride, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
      return FEC_E_DEVICE;
    unsigned char* flags = (unsigned char*)ctx->d_buf[5];
    {
      Launch L(ctx, nullptr, op == 0 ? "k_decompress" : "k_decode_uncompressed");
      codec_launch(op, curve, ctx->d_buf[0], nullptr, ctx->d_buf[4], flags, flags + chunk, cnt, L.s);
      rc = L.done();
      if (rc != FEC_OK) return rc;
    }
    if (hipMemcpyAsync(xy + lo * 8, ctx->d_buf[4], cnt * 64, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(inf + lo, flags, cnt, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(ok + lo, flags + chunk, cnt, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
      return FEC_E_DEVICE;
    rc = sync_and_check(ctx, ctx->stream);
    if (rc != FEC_OK) return rc;
  }
  return FEC_OK;
  });
}

int fec_batch_decompress(fec_ctx* ctx, fec_curve curve, const uint8_t* in, uint64_t* xy, uint8_t* inf, uint8_t* ok,
                         size_t n) try {
  if (is_multi(ctx)) {
    if (!curve_ok(curve) || (n && (!in || !xy || !inf || !ok))) return FEC_E_ARG;
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return fec_batch_decompress(c, curve, in + lo * 33, xy + lo * 8, inf + lo, ok + lo, cnt);
    });
  }
  if (!ctx || !curve_ok(curve) || (n && (!in || !xy || !inf || !ok))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  return decode_host(ctx, 0, curve, in, 33, xy, inf, ok, n);
} FEC_ABI_CATCH_STATUS

int fec_batch_decode_uncompressed(fec_ctx* ctx, fec_curve curve, const uint8_t* in, uint64_t* xy, uint8_t* inf,
                                  uint8_t* ok, size_t n) try {
  if (is_multi(ctx)) {
    if (!curve_ok(curve) || (n && (!in || !xy || !inf || !ok))) return FEC_E_ARG;
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return fec_batch_decode_uncompressed(c, curve, in + lo * 65, xy + lo * 8, inf + lo, ok + lo, cnt);
    });
  }
  if (!ctx || !curve_ok(curve) || (n && (!in || !xy || !inf || !ok))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  return decode_host(ctx, 1, curve, in, 65, xy, inf, ok, n);
} FEC_ABI_CATCH_STATUS

int fec_batch_encode_uncompressed(fec_ctx* ctx, fec_curve curve, const uint64_t* xy, const uint8_t* inf, uint8_t* out,
                                  size_t n) try {
  if (is_multi(ctx)) {
    if (!curve_ok(curve) || (n && (!xy || !out))) return FEC_E_ARG;
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return fec_batch_encode_uncompressed(c, curve, xy + lo * 8, inf ? inf + lo : nullptr, out + lo * 65, cnt);
    });
  }
  if (!ctx || !curve_ok(curve) || (n && (!xy || !out))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  const void* const in[4] = {xy, inf, nullptr, nullptr};
  const size_t in_stride[4] = {64, 1, 0, 0};
  void* const outs[2] = {out, nullptr};
  const size_t out_stride[2] = {65, 0};
  return host_chunked(ctx, n, in, in_stride, outs, out_stride, [&](void* const d[4], void* const o[2], size_t cnt) {
    Launch L(ctx, nullptr, "k_encode_uncompressed");
    codec_launch(2, curve, d[0], d[1], o[0], nullptr, nullptr, cnt, L.s);
    return L.done();
  });
} FEC_ABI_CATCH_STATUS

int fec_batch_to_affine_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_points, uint64_t* d_xy,
                            uint8_t* d_inf, size_t n, void* stream) try {
  if (is_multi(ctx)) return FEC_E_UNSUPPORTED;  // device pointers belong to one device
  if (!ctx || !curve_ok(curve) || (n && (!d_points || !d_xy || !d_inf))) return FEC_E_ARG;
  if (!aligned16(d_points) || !aligned16(d_xy)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return launch_to_affine(ctx, curve, d_points, d_xy, d_inf, n, stream);
} FEC_ABI_CATCH_STATUS

int fec_batch_to_affine(fec_ctx* ctx, fec_curve curve, const uint64_t* points, uint64_t* xy, uint8_t* inf,
                        size_t n) try {
  if (is_multi(ctx)) {
    if (!curve_ok(curve) || (n && (!points || !xy || !inf))) return FEC_E_ARG;
    const size_t pl = (size_t)plimbs(curve);
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return fec_batch_to_affine(c, curve, points + lo * pl, xy + lo * 8, inf + lo, cnt);
    });
  }
  if (!ctx || !curve_ok(curve) || (n && (!points || !xy || !inf))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  size_t pb = (size_t)plimbs(curve) * 8;
  return drained(ctx, [&]() -> int {
  int rc = ensure(ctx, 0, n * pb);
  if (rc == FEC_OK) rc = ensure(ctx, 3, n * 64);
  if (rc == FEC_OK) rc = ensure(ctx, 1, n);
  if (rc != FEC_OK) return rc;
  if (hipMemcpyAsync(ctx->d_buf[0], points, n * pb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
    return FEC_E_DEVICE;
  rc = launch_to_affine(ctx, curve, (const u64*)ctx->d_buf[0], (u64*)ctx->d_buf[3],
                        (unsigned char*)ctx->d_buf[1], n, nullptr);
  if (rc != FEC_OK) return rc;
  if (hipMemcpyAsync(xy, ctx->d_buf[3], n * 64, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(inf, ctx->d_buf[1], n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
    return FEC_E_DEVICE;
  return sync_and_check(ctx, ctx->stream);
  });
} FEC_ABI_CATCH_STATUS

int fec_field_op(fec_ctx* ctx, fec_curve curve, fec_field_opcode op, const uint64_t* a, const uint64_t* b,
                 uint64_t* out, size_t n) try {
  if (is_multi(ctx)) {
    if (!curve_ok(curve) || (n && (!a || !out))) return FEC_E_ARG;
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return fec_field_op(c, curve, op, a + lo * 4, b ? b + lo * 4 : nullptr, out + lo * 4, cnt);
    });
  }
  if (!ctx || !curve_ok(curve) || op < FEC_F_ADD || op > FEC_F_NEG || (n && (!a || !out))) return FEC_E_ARG;
  bool binary = op == FEC_F_ADD || op == FEC_F_SUB || op == FEC_F_MUL;
  if (binary && n && !b) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  const HostIn in[3] = {{a, 32, 0}, {binary ? b : nullptr, 32, 0}, {nullptr, 0, 0}};
  return host_pipeline(ctx, n, in, out, 32, [&](void* x, void* y, void*, void* o, size_t cnt, void* s) {
    return launch_field(ctx, curve, op, (const u64*)x, (const u64*)y, (u64*)o, cnt, s);
  });
} FEC_ABI_CATCH_STATUS

int fec_point_op(fec_ctx* ctx, fec_curve curve, fec_point_opcode op, const uint64_t* p, const uint64_t* q,
                 uint64_t* out, size_t n) try {
  if (is_multi(ctx)) {
    if (!curve_ok(curve) || (n && (!p || !out))) return FEC_E_ARG;
    const size_t pl = (size_t)plimbs(curve);
    return multi_shard(ctx, n, [=](fec_ctx* c, size_t lo, size_t cnt) {
      return fec_point_op(c, curve, op, p + lo * pl, q ? q + lo * pl : nullptr, out + lo * pl, cnt);
    });
  }
  if (!ctx || !curve_ok(curve) || op < FEC_P_ADD || op > FEC_P_DOUBLE_TRAIT || (n && (!p || !out)))
    return FEC_E_ARG;
  if (op == FEC_P_DOUBLE_TRAIT && curve != FEC_SECP256K1) return FEC_E_UNSUPPORTED;
  if (op == FEC_P_ADD && n && !q) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  size_t pb = (size_t)plimbs(curve) * 8;
  const HostIn in[3] = {{p, pb, 0}, {op == FEC_P_ADD ? q : nullptr, pb, 0}, {nullptr, 0, 0}};
  return host_pipeline(ctx, n, in, out, pb, [&](void* x, void* y, void*, void* o, size_t cnt, void* s) {
    return launch_point(ctx, curve, op, (const u64*)x, (const u64*)y, (u64*)o, cnt, s);
  });
} FEC_ABI_CATCH_STATUS

int fec_generator(fec_ctx* ctx, fec_curve curve, uint64_t* out) try {
  FEC_FIRST_DEVICE(ctx);
  if (!ctx || !curve_ok(curve) || !out) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  if (hipMemcpy(out, ctx->d_gen[curve], (size_t)plimbs(curve) * 8, hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return FEC_E_DEVICE;
  }
  return FEC_OK;
} FEC_ABI_CATCH_STATUS

const uint64_t* fec_generator_dev(fec_ctx* ctx, fec_curve curve) try {
  FEC_FIRST_DEVICE(ctx);
  if (!ctx || !curve_ok(curve)) return nullptr;
  return ctx->d_gen[curve];
} FEC_ABI_CATCH_NULL

int fec_ctx_wipe(fec_ctx* ctx) try {
  if (!ctx) return FEC_E_ARG;
  if (is_multi(ctx)) {
    int rc = FEC_OK;
    for (fec_ctx* c : ctx->children) {
      const int r = fec_ctx_wipe(c);
      if (rc == FEC_OK) rc = r;
    }
    return rc;
  }
  if (ctx->device < 0 || !ctx->stream) return FEC_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  (void)hipDeviceSynchronize();
  // hipMemset is asynchronous with respect to the host and runs on the NULL stream, which the ctx's non-blocking
  // stream does not synchronise with: a launch issued right after the wipe could run before the tail of the
  // memsets and have its work area zeroed under it (seen as an intermittent "point at infinity" from the canonical
  // mul_base that followed a signing helper).  The wipe therefore goes to the ctx stream and is waited for.
  hipStream_t st = ctx->stream;
  // every buffer is attempted whatever happened to the ones before it (no short-circuit: a failure must not leave
  // the remaining buffers -- u1/u2, ECDH work areas, signing nonces -- uncleared)
  int failed = 0;
  auto zero = [&](void* p, size_t bytes) {
    if (p && bytes && hipMemsetAsync(p, 0, bytes, st) != hipSuccess) {
      (void)hipGetLastError();
      ++failed;
    }
  };
  for (int i = 0; i < 8; ++i) zero(ctx->d_buf[i], ctx->d_cap[i]);
  for (auto& e : ctx->stream_scratch) zero(e.buf, e.cap);
  zero(ctx->d_win_scratch, ctx->win_scratch_cap);
  zero(ctx->d_zbuf, ctx->zbuf_cap);
  zero(ctx->d_tbuf, ctx->tbuf_cap);
  zero(ctx->d_verify, ctx->verify_cap);
  if (hipStreamSynchronize(st) != hipSuccess) {
    (void)hipGetLastError();
    ++failed;
  }
  return failed ? FEC_E_DEVICE : FEC_OK;
} FEC_ABI_CATCH_STATUS

// The sticky device error state of the ctx (see kernels.hpp: SchedEnv), for callers of the *_dev entry points: those
// return as soon as the work is enqueued, so a fault a kernel reports can only be seen afterwards.  Synchronises the
// ctx's device, then returns FEC_E_LAUNCH if any kernel launched through this ctx since the last check reported a
// fault (its outputs must not be used), FEC_OK otherwise; reading clears the state.
int fec_ctx_check(fec_ctx* ctx) try {
  if (!ctx) return FEC_E_ARG;
  if (is_multi(ctx)) {
    int rc = FEC_OK;
    for (fec_ctx* c : ctx->children) {
      const int r = fec_ctx_check(c);
      if (rc == FEC_OK) rc = r;
    }
    return rc;
  }
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  if (hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    (void)take_device_error(ctx);
    return FEC_E_LAUNCH;
  }
  return take_device_error(ctx);
} FEC_ABI_CATCH_STATUS

// Debug hook: while enabled, every launch of a scheduler kernel (P-256, Ed25519 variable base -- also inside the
// composed entry points) raises its fault word at once, exactly as the watchdog would.  Lets a test assert that a
// scheduler fault comes back as FEC_E_LAUNCH.
int fec_ctx_debug_force_fault(fec_ctx* ctx, int enabled) try {
  if (!ctx) return FEC_E_ARG;
  if (is_multi(ctx)) {
    for (fec_ctx* c : ctx->children) c->debug_force_fault = enabled ? 1u : 0u;
    return FEC_OK;
  }
  ctx->debug_force_fault = enabled ? 1u : 0u;
  return FEC_OK;
} FEC_ABI_CATCH_STATUS

int fec_ctx_fixed_prefix_bits(fec_ctx* ctx, fec_curve curve) try {
  FEC_FIRST_DEVICE(ctx);
  if (!ctx || !curve_ok(curve)) return FEC_E_ARG;
  return ctx->d_gen_prefix[curve] ? (int)ctx->gen_prefix_bits[curve] : 0;
} FEC_ABI_CATCH_STATUS

int fec_ctx_set_fixed_prefix_bits(fec_ctx* ctx, unsigned bits) try {
  if (!ctx || bits > kMaxPrefixBits) return FEC_E_ARG;
  if (is_multi(ctx)) {
    int rc = FEC_OK;
    for (fec_ctx* c : ctx->children) {
      const int r = fec_ctx_set_fixed_prefix_bits(c, bits);
      if (rc == FEC_OK) rc = r;
    }
    return rc;
  }
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  drop_gen_prefix(ctx);   // the ctx's references go; tables of the new size are attached / built by the next fixed-base launches
  ctx->prefix_bits = bits;
  ctx->prefix_after = 0;        // asked for explicitly: no waiting for the ctx to have multiplied enough,
  ctx->prefix_explicit = true;  // and any launch -- a *_dev one too -- may build
  return FEC_OK;
} FEC_ABI_CATCH_STATUS

int fec_ctx_build_fixed_prefix(fec_ctx* ctx, fec_curve curve) try {
  if (!ctx || !curve_ok(curve)) return FEC_E_ARG;
  if (is_multi(ctx)) {
    int rc = FEC_OK;
    for (fec_ctx* c : ctx->children) {
      const int r = fec_ctx_build_fixed_prefix(c, curve);
      if (rc == FEC_OK) rc = r;
    }
    return rc;
  }
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  (void)attach_gen_prefix(ctx, curve, ctx->stream, true);   // (refused memory is not an error: fec_ctx_fixed_prefix_bits says what there is)
  return sync_and_check(ctx, ctx->stream);
} FEC_ABI_CATCH_STATUS

int fec_ctx_set_fixed_prefix_after(fec_ctx* ctx, size_t elements) try {
  if (!ctx) return FEC_E_ARG;
  if (is_multi(ctx)) {
    for (fec_ctx* c : ctx->children) c->prefix_after = elements;
    return FEC_OK;
  }
  ctx->prefix_after = elements;
  return FEC_OK;
} FEC_ABI_CATCH_STATUS

int fec_ctx_set_fixed_prefix_budget(fec_ctx* ctx, unsigned percent_of_free_memory) try {
  if (!ctx || percent_of_free_memory > 100) return FEC_E_ARG;
  if (is_multi(ctx)) {
    for (fec_ctx* c : ctx->children) c->prefix_budget_pct = percent_of_free_memory;
    return FEC_OK;
  }
  ctx->prefix_budget_pct = percent_of_free_memory;
  return FEC_OK;
} FEC_ABI_CATCH_STATUS

int fec_ctx_set_side_stream_max(fec_ctx* ctx, size_t elements) try {
  if (!ctx) return FEC_E_ARG;
  if (is_multi(ctx)) {
    for (fec_ctx* c : ctx->children) c->side_stream_max = elements;
    return FEC_OK;
  }
  ctx->side_stream_max = elements;
  return FEC_OK;
} FEC_ABI_CATCH_STATUS

int fec_ctx_set_chunk(fec_ctx* ctx, size_t elements) try {
  if (is_multi(ctx)) {
    if (elements == 0) return FEC_E_ARG;
    for (fec_ctx* c : ctx->children) c->chunk = elements;
    return FEC_OK;
  }
  if (!ctx || elements == 0) return FEC_E_ARG;
  ctx->chunk = elements;
  return FEC_OK;
} FEC_ABI_CATCH_STATUS

int fec_ctx_set_timing(fec_ctx* ctx, int enabled) try {
  FEC_FIRST_DEVICE(ctx);
  if (!ctx) return FEC_E_ARG;
  ctx->timing = enabled != 0;
  ctx->timed = false;
  return FEC_OK;
} FEC_ABI_CATCH_STATUS

int fec_ctx_last_kernel_ms(fec_ctx* ctx, float* ms, const char** kernel_name) try {
  FEC_FIRST_DEVICE(ctx);
  if (!ctx || !ms) return FEC_E_ARG;
  if (!ctx->timed) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  if (hipEventSynchronize(ctx->ev1) != hipSuccess || hipEventElapsedTime(ms, ctx->ev0, ctx->ev1) != hipSuccess) {
    (void)hipGetLastError();
    return FEC_E_LAUNCH;
  }
  if (kernel_name) *kernel_name = ctx->last_kernel;
  return FEC_OK;
} FEC_ABI_CATCH_STATUS

int fec_measure_peak_mad32(fec_ctx* ctx, double* mad32_per_sec) try {
  FEC_FIRST_DEVICE(ctx);
  if (!ctx || !mad32_per_sec) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  const int blocks = ctx->prop.multiProcessorCount * 8;  // 8 wavefronts per SIMD
  int rc = ensure(ctx, 3, (size_t)blocks * TPB * sizeof(u32));
  if (rc != FEC_OK) return rc;
  u32* out = (u32*)ctx->d_buf[3];
  double best = 0;
  for (int rep = 0; rep < 4; ++rep) {  // first rep warms up
    (void)hipEventRecord(ctx->ev0, ctx->stream);
    hipLaunchKernelGGL(k_peak_mad32, dim3(blocks), dim3(TPB), 0, ctx->stream, out, (u32)rep);
    (void)hipEventRecord(ctx->ev1, ctx->stream);
    if (hipGetLastError() != hipSuccess || hipEventSynchronize(ctx->ev1) != hipSuccess) return FEC_E_LAUNCH;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    double rate = (double)blocks * TPB * (double)PEAK_ITERS * 64.0 / (ms * 1e-3);
    if (rep > 0 && rate > best) best = rate;
  }
  *mad32_per_sec = best;
  return FEC_OK;
} FEC_ABI_CATCH_STATUS

int fec_ctx_device_info(fec_ctx* ctx, char* name, size_t name_len, int* compute_units, int* clock_khz) try {
  FEC_FIRST_DEVICE(ctx);
  if (!ctx) return FEC_E_ARG;
  if (name && name_len) {
    std::snprintf(name, name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
  }
  if (compute_units) *compute_units = ctx->prop.multiProcessorCount;
  if (clock_khz) *clock_khz = ctx->prop.clockRate;
  return FEC_OK;
} FEC_ABI_CATCH_STATUS

}  // extern "C"
