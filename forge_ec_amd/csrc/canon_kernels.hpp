// canon_kernels.hpp -- kernels of the CANONICAL-MATH MODE (secp256k1, P-256), included by fecgpu.hip.
//
// NOT reference parity (see canon_curves.hpp).  Same mapping as the parity kernels: one scalar
// per lane, 256-thread workgroups, coalesced 16-byte HBM<->LDS staging of word-major columns.
#pragma once
#include "canon_curves.hpp"

namespace fecgpu {

enum { CANON_FINITE = canon::ST_FINITE, CANON_INFINITY = canon::ST_INFINITY, CANON_BAD_POINT = canon::ST_BAD_POINT };

// table[(i * 15 + j - 1) * COMB_STRIDE ...] = affine j * 16^i * G, i = 0..63, j = 1..15.
// One wavefront, once per context: lane 0 walks the 16^i * G chain (252 doublings), then lane i
// derives its 15 multiples by repeated addition and normalises each.
template <class W>
__global__ __launch_bounds__(64) void k_canon_build_comb(u32* __restrict__ table) {
  __shared__ u32 lds_b[24 * 64];
  const int lane = threadIdx.x;
  if (lane == 0) {
    canon::aff g = W::generator();
    canon::jac b;
    b.x = g.x;
    b.y = g.y;
    b.z = fe_small(1);
#pragma unroll 1
    for (int i = 0; i < canon::COMB_WINDOWS; ++i) {
      store_fe(lds_b + i, 64, b.x);
      store_fe(lds_b + 8 * 64 + i, 64, b.y);
      store_fe(lds_b + 16 * 64 + i, 64, b.z);
#pragma unroll 1
      for (int d = 0; d < 4; ++d) b = W::jdouble(b);
    }
  }
  __syncthreads();
  canon::jac acc;
  acc.x = load_fe(lds_b + lane, 64);
  acc.y = load_fe(lds_b + 8 * 64 + lane, 64);
  acc.z = load_fe(lds_b + 16 * 64 + lane, 64);
  canon::aff base;
  W::to_affine(acc, base);
  W::comb_fill_window(table, lane, base);
}

// Phase 1 of key generation: Jacobian scalars[i] * G by the comb; X, Y go to out_xy[i], Z to zbuf[i].
template <class W>
__global__ __launch_bounds__(TPB) void k_canon_mul_base(const u32* __restrict__ scalars,
                                                        const u32* __restrict__ table,
                                                        u32* __restrict__ out_xy, u32* __restrict__ zbuf,
                                                        unsigned char* __restrict__ status, size_t n) {
  __shared__ u32 lds_k[8 * TPB];
  __shared__ u32 lds_t[canon::COMB_WORDS];
  const int valid = block_valid(n);
  const size_t first = (size_t)blockIdx.x * TPB;
  stage_in<8>(lds_k, scalars + first * 8, valid);
  for (int v = threadIdx.x; v < canon::COMB_WORDS; v += TPB) lds_t[v] = table[v];
  __syncthreads();
  const int e = threadIdx.x;
  if (e < valid) {
    canon::jac r = W::mul_base_comb(lds_t, lds_k + e);
    const size_t i = first + e;
    canon::st8(out_xy + i * 16, r.x);
    canon::st8(out_xy + i * 16 + 8, r.y);
    canon::st8(zbuf + i * 8, r.z);
    status[i] = CANON_FINITE;
  }
}

// Phase 1 of ECDH: Jacobian scalars[i] * points_xy[i] by the windowed ladder.  `scratch` holds one
// 15-entry window table per element (WIN_ENTRIES * WIN_ENTRY_WORDS words each), private to the lane
// that builds it.  Rejected input points get status CANON_BAD_POINT (phase 2 zeroes them).
// ACCUM: out_xy / zbuf already hold a Jacobian point per element (u1 * G from the comb kernel); the
// ladder's result is added to it -- u1*G + u2*P, the verification pattern, with G's half free of doublings.
template <class W, bool ACCUM>
__global__ __launch_bounds__(TPB, 2) void k_canon_mul(const u32* __restrict__ scalars,
                                                   const u32* __restrict__ points_xy,
                                                   u32* __restrict__ scratch, u32* __restrict__ out_xy,
                                                   u32* __restrict__ zbuf, unsigned char* __restrict__ status,
                                                   size_t n) {
  __shared__ u32 lds_k[8 * TPB];
  __shared__ u32 lds_p[16 * TPB];
  const int valid = block_valid(n);
  const size_t first = (size_t)blockIdx.x * TPB;
  stage_in<8>(lds_k, scalars + first * 8, valid);
  stage_in<16>(lds_p, points_xy + first * 16, valid);
  __syncthreads();
  const int e = threadIdx.x;
  if (e < valid) {
    canon::aff base;
    base.x = load_fe(lds_p + e, TPB);
    base.y = load_fe(lds_p + 8 * TPB + e, TPB);
    const lmask ok = W::on_curve(base);
    // a bad point still runs the ladder (on garbage; the arithmetic is total) and is zeroed in phase 2
    const size_t i = first + e;
    u32* table = scratch + i * (size_t)(canon::WIN_ENTRIES * canon::WIN_ENTRY_WORDS);
    canon::jac r;
    if constexpr (W::P_HAS_GLV) r = cglv::mul_window(base, lds_k + e, table);  // secp256k1: half the doublings
    else r = W::mul_window(base, lds_k + e, table);
    if (ACCUM) {
      canon::jac a;
      a.x = canon::ld8(out_xy + i * 16);
      a.y = canon::ld8(out_xy + i * 16 + 8);
      a.z = canon::ld8(zbuf + i * 8);
      r = W::jadd(a, r);
    }
    canon::st8(out_xy + i * 16, r.x);
    canon::st8(out_xy + i * 16 + 8, r.y);
    canon::st8(zbuf + i * 8, r.z);
    status[i] = lane_of(ok) ? CANON_FINITE : CANON_BAD_POINT;
  }
}

// Phase 2: Jacobian -> affine in place, one inversion per NORM_GROUP elements per lane.
template <class W>
__global__ __launch_bounds__(TPB, 2) void k_canon_normalize(u32* __restrict__ xy, const u32* __restrict__ zbuf,
                                                         unsigned char* __restrict__ status, size_t n,
                                                         size_t stride) {
  const size_t g = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (g < stride) W::normalize_group(xy, zbuf, status, g, stride, n);
}

// canonical field ops for tests and callers: op = fec_field_opcode, plus FEC_F_NEG + 1 = inverse
template <class W>
__global__ __launch_bounds__(TPB) void k_canon_field_op(int op, const u32* __restrict__ a,
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
      case FEC_F_ADD: r = W::add(x, y); break;
      case FEC_F_SUB: r = W::sub(x, y); break;
      case FEC_F_MUL: r = W::mul(x, y); break;
      case FEC_F_SQR: r = W::sqr(x); break;
      case FEC_F_NEG: r = W::neg(x); break;
      default: r = W::inv(x); break;
    }
    store_fe(lds_a + e, TPB, r);
  }
  __syncthreads();
  stage_out<8>(out + first * 8, lds_a, valid);
}

// ---- 8-bit comb in global memory (L2-resident), Weierstrass curves -------------------------------
// table[(w * 255 + j - 1) * 16 ...] = affine j * 256^w * G.  Block w: thread 0 walks to 256^w * G
// (8 w doublings), thread j derives j times it by an 8-step double-and-add and normalises.
template <class W>
__global__ __launch_bounds__(TPB, 2) void k_canon_build_comb8(u32* __restrict__ table) {
  __shared__ u32 lds_b[16];
  const int w = blockIdx.x;
  const u32 j = threadIdx.x;
  if (j == 0) {
    canon::aff g = W::generator();
    canon::jac b;
    b.x = g.x;
    b.y = g.y;
    b.z = fe_small(1);
#pragma unroll 1
    for (int d = 0; d < 8 * w; ++d) b = W::jdouble(b);
    canon::aff base;
    W::to_affine(b, base);
    store_fe(lds_b, 1, base.x);
    store_fe(lds_b + 8, 1, base.y);
  }
  __syncthreads();
  if (j == 0) return;
  canon::aff base;
  base.x = load_fe(lds_b, 1);
  base.y = load_fe(lds_b + 8, 1);
  canon::jac acc = W::small_multiple(base, j);
  canon::aff e;
  W::to_affine(acc, e);
  W::comb8_store(table, w, j, e);
}

// Phase 1 of key generation with the 8-bit comb: 32 gathers + 32 mixed additions per key, no LDS table.
template <class W>
__global__ __launch_bounds__(TPB, 2) void k_canon_mul_base8(const u32* __restrict__ scalars,
                                                            const u32* __restrict__ table,
                                                            u32* __restrict__ out_xy, u32* __restrict__ zbuf,
                                                            unsigned char* __restrict__ status, size_t n) {
  __shared__ u32 lds_k[8 * TPB];
  const int valid = block_valid(n);
  const size_t first = (size_t)blockIdx.x * TPB;
  stage_in<8>(lds_k, scalars + first * 8, valid);
  __syncthreads();
  const int e = threadIdx.x;
  if (e < valid) {
    canon::jac r = W::mul_base_comb8(table, lds_k + e);
    const size_t i = first + e;
    canon::st8(out_xy + i * 16, r.x);
    canon::st8(out_xy + i * 16 + 8, r.y);
    canon::st8(zbuf + i * 8, r.z);
    status[i] = CANON_FINITE;
  }
}

// ---- Ed25519 (extended coordinates, signed comb) ------------------------------------------------
// table: 64 windows x 8 affine Niels multiples of 16^w * B, plus 2^256 * B (ED_COMB_WORDS words).
__global__ __launch_bounds__(64) void k_ced_build_comb(u32* __restrict__ table) {
  __shared__ u32 lds_b[32 * 65];
  const int lane = threadIdx.x;
  if (lane == 0) {
    canon::ext b = ced::from_affine(ced::generator());
#pragma unroll 1
    for (int i = 0; i <= canon::COMB_WINDOWS; ++i) {
      store_fe(lds_b + i, 65, b.x);
      store_fe(lds_b + 8 * 65 + i, 65, b.y);
      store_fe(lds_b + 16 * 65 + i, 65, b.z);
#pragma unroll 1
      for (int d = 0; d < 4; ++d) b = ced::dbl<true>(b);
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {  // pass 1: lane 0 alone stores the 2^256 * B entry
    const int idx = pass == 0 ? lane : canon::COMB_WINDOWS;
    if (pass == 1 && lane != 0) break;
    canon::ext acc;
    acc.x = load_fe(lds_b + idx, 65);
    acc.y = load_fe(lds_b + 8 * 65 + idx, 65);
    acc.z = load_fe(lds_b + 16 * 65 + idx, 65);
    acc.t = fe_zero();
    canon::aff base = ced::to_affine(acc);
    if (pass == 0) ced::comb_fill_window(table, lane, base);
    else ced::comb_store(table, canon::COMB_WINDOWS * canon::ED_COMB_ENTRIES, base);
  }
}

// tbuf (may be null): also keep T, for a following accumulate pass
__global__ __launch_bounds__(TPB) void k_ced_mul_base(const u32* __restrict__ scalars, const u32* __restrict__ table,
                                                      u32* __restrict__ out_xy, u32* __restrict__ zbuf,
                                                      u32* __restrict__ tbuf, unsigned char* __restrict__ status,
                                                      size_t n) {
  __shared__ u32 lds_k[8 * TPB];
  __shared__ u32 lds_t[canon::ED_COMB_WORDS];
  const int valid = block_valid(n);
  const size_t first = (size_t)blockIdx.x * TPB;
  stage_in<8>(lds_k, scalars + first * 8, valid);
  for (int v = threadIdx.x; v < canon::ED_COMB_WORDS; v += TPB) lds_t[v] = table[v];
  __syncthreads();
  const int e = threadIdx.x;
  if (e < valid) {
    canon::ext r = ced::mul_base_comb(lds_t, lds_k + e);
    const size_t i = first + e;
    canon::st8(out_xy + i * 16, r.x);
    canon::st8(out_xy + i * 16 + 8, r.y);
    canon::st8(zbuf + i * 8, r.z);
    if (tbuf) canon::st8(tbuf + i * 8, r.t);
    status[i] = CANON_FINITE;
  }
}

// Signed 8-bit comb table for Ed25519: block w < 32 fills j * 256^w * B for j = 1..128, block 32 the
// single entry 2^256 * B.
__global__ __launch_bounds__(TPB, 2) void k_ced_build_comb8(u32* __restrict__ table) {
  __shared__ u32 lds_b[16];
  const int w = blockIdx.x;
  const u32 j = threadIdx.x;
  if (j == 0) {
    canon::ext b = ced::from_affine(ced::generator());
#pragma unroll 1
    for (int d = 0; d < 8 * w; ++d) b = ced::dbl<true>(b);
    canon::aff base = ced::to_affine(b);
    store_fe(lds_b, 1, base.x);
    store_fe(lds_b + 8, 1, base.y);
  }
  __syncthreads();
  const u32 last = w == canon::COMB8_WINDOWS ? 1u : (u32)canon::ED_COMB8_ENTRIES;
  if (j == 0 || j > last) return;
  canon::aff base;
  base.x = load_fe(lds_b, 1);
  base.y = load_fe(lds_b + 8, 1);
  canon::aff e = ced::to_affine(ced::small_multiple(base, j));
  ced::comb8_store(table, (size_t)w * canon::ED_COMB8_ENTRIES + j - 1, e);
}

__global__ __launch_bounds__(TPB, 2) void k_ced_mul_base8(const u32* __restrict__ scalars, const u32* __restrict__ table,
                                                          u32* __restrict__ out_xy, u32* __restrict__ zbuf,
                                                          u32* __restrict__ tbuf, unsigned char* __restrict__ status,
                                                          size_t n) {
  __shared__ u32 lds_k[8 * TPB];
  const int valid = block_valid(n);
  const size_t first = (size_t)blockIdx.x * TPB;
  stage_in<8>(lds_k, scalars + first * 8, valid);
  __syncthreads();
  const int e = threadIdx.x;
  if (e < valid) {
    canon::ext r = ced::mul_base_comb8(table, lds_k + e);
    const size_t i = first + e;
    canon::st8(out_xy + i * 16, r.x);
    canon::st8(out_xy + i * 16 + 8, r.y);
    canon::st8(zbuf + i * 8, r.z);
    if (tbuf) canon::st8(tbuf + i * 8, r.t);
    status[i] = CANON_FINITE;
  }
}

// `scratch`: ED_WIN_ENTRIES * 32 words per element
// tbuf != null: accumulate onto the extended point already in out_xy / zbuf / tbuf
__global__ __launch_bounds__(TPB, 2) void k_ced_mul(const u32* __restrict__ scalars, const u32* __restrict__ points_xy,
                                                    u32* __restrict__ scratch, u32* __restrict__ out_xy,
                                                    u32* __restrict__ zbuf, const u32* __restrict__ tbuf,
                                                    unsigned char* __restrict__ status, size_t n) {
  __shared__ u32 lds_k[8 * TPB];
  __shared__ u32 lds_p[16 * TPB];
  const int valid = block_valid(n);
  const size_t first = (size_t)blockIdx.x * TPB;
  stage_in<8>(lds_k, scalars + first * 8, valid);
  stage_in<16>(lds_p, points_xy + first * 16, valid);
  __syncthreads();
  const int e = threadIdx.x;
  if (e < valid) {
    canon::aff base;
    base.x = load_fe(lds_p + e, TPB);
    base.y = load_fe(lds_p + 8 * TPB + e, TPB);
    const lmask ok = ced::on_curve(base);
    const size_t i = first + e;
    u32* table = scratch + i * (size_t)(canon::ED_WIN_ENTRIES * 32);
    canon::ext r = ced::mul_window(base, lds_k + e, table);
    if (tbuf) {
      canon::ext a;
      a.x = canon::ld8(out_xy + i * 16);
      a.y = canon::ld8(out_xy + i * 16 + 8);
      a.z = canon::ld8(zbuf + i * 8);
      a.t = canon::ld8(tbuf + i * 8);
      r = ced::add_pniels(r, ced::to_pniels(a), 0, 0);
    }
    canon::st8(out_xy + i * 16, r.x);
    canon::st8(out_xy + i * 16 + 8, r.y);
    canon::st8(zbuf + i * 8, r.z);
    status[i] = lane_of(ok) ? CANON_FINITE : CANON_BAD_POINT;
  }
}

// ---- canonical ECDSA verification (secp256k1, P-256) ---------------------------------------------
// scalar half: range check, w = s^-1 mod n (one inversion per NORM_GROUP signatures per lane), u1 = z w, u2 = r w
template <class N>
__global__ __launch_bounds__(TPB, 2) void k_canon_ecdsa_scalars(const u32* __restrict__ zs, const u32* __restrict__ rs,
                                                                const u32* __restrict__ ss, u32* __restrict__ u1,
                                                                u32* __restrict__ u2, unsigned char* __restrict__ ok,
                                                                size_t n, size_t stride) {
  const size_t g = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (g < stride) canon::ecdsa_scalars_group<N>(zs, rs, ss, u1, u2, ok, g, stride, n);
}

// final comparison: valid iff the range check passed, the public key was accepted, R is finite and
// x(R) mod n == r
template <class N>
__global__ __launch_bounds__(TPB) void k_canon_ecdsa_finish(const u32* __restrict__ xy, const u32* __restrict__ rs,
                                                            const unsigned char* __restrict__ ok,
                                                            const unsigned char* __restrict__ point_status,
                                                            unsigned char* __restrict__ result, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const fe x = canon::ld8(xy + i * 16), r = canon::ld8(rs + i * 8);
  const bool same = lane_of(canon::ecdsa_x_matches<N>(x, r));
  result[i] = (ok[i] != 0 && point_status[i] == CANON_FINITE && same) ? 1 : 0;
}

// ---- BIP-340 (secp256k1) and EdDSA (Ed25519) verification: decode / scalar half and final test --------
__global__ __launch_bounds__(TPB, 2) void k_canon_bip340_prepare(const u32* __restrict__ pkx, const u32* __restrict__ rs,
                                                                 const u32* __restrict__ ss, const u32* __restrict__ es,
                                                                 u32* __restrict__ pxy, u32* __restrict__ u2,
                                                                 unsigned char* __restrict__ ok, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  canon::aff P;
  fe v;
  const lmask good = canon::bip340_prepare(canon::ld8(pkx + i * 8), canon::ld8(rs + i * 8), canon::ld8(ss + i * 8),
                                           canon::ld8(es + i * 8), P, v);
  const bool mine = lane_of(good);
  if (!mine) {  // keep the ladder on a valid point; the lane is rejected by `ok`
    canon::aff g = csecp::generator();
    P = g;
  }
  canon::st8(pxy + i * 16, P.x);
  canon::st8(pxy + i * 16 + 8, P.y);
  canon::st8(u2 + i * 8, v);
  ok[i] = mine ? 1 : 0;
}
__global__ __launch_bounds__(TPB) void k_canon_bip340_finish(const u32* __restrict__ xy, const u32* __restrict__ rs,
                                                             const unsigned char* __restrict__ ok,
                                                             const unsigned char* __restrict__ point_status,
                                                             unsigned char* __restrict__ result, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const bool acc = lane_of(canon::bip340_accept(canon::ld8(xy + i * 16), canon::ld8(xy + i * 16 + 8), canon::ld8(rs + i * 8)));
  result[i] = (ok[i] != 0 && point_status[i] == CANON_FINITE && acc) ? 1 : 0;
}
__global__ __launch_bounds__(TPB, 2) void k_ced_eddsa_prepare(const u32* __restrict__ a_enc, const u32* __restrict__ r_enc,
                                                              const u32* __restrict__ ss, const u32* __restrict__ hs,
                                                              u32* __restrict__ axy, u32* __restrict__ rxy,
                                                              u32* __restrict__ u2, unsigned char* __restrict__ ok,
                                                              size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  canon::aff A, R;
  fe v;
  const lmask good = canon::eddsa_prepare(canon::ld8(a_enc + i * 8), canon::ld8(r_enc + i * 8), canon::ld8(ss + i * 8),
                                          canon::ld8(hs + i * 8), A, R, v);
  const bool mine = lane_of(good);
  if (!mine) A = ced::generator();
  canon::st8(axy + i * 16, A.x);
  canon::st8(axy + i * 16 + 8, A.y);
  canon::st8(rxy + i * 16, R.x);
  canon::st8(rxy + i * 16 + 8, R.y);
  canon::st8(u2 + i * 8, v);
  ok[i] = mine ? 1 : 0;
}
// S B - h A == R, compared as affine points
__global__ __launch_bounds__(TPB) void k_ced_eddsa_finish(const u32* __restrict__ xy, const u32* __restrict__ rxy,
                                                          const unsigned char* __restrict__ ok,
                                                          const unsigned char* __restrict__ point_status,
                                                          unsigned char* __restrict__ result, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const bool same = lane_of(fe_eq(canon::ld8(xy + i * 16), canon::ld8(rxy + i * 16)) &
                            fe_eq(canon::ld8(xy + i * 16 + 8), canon::ld8(rxy + i * 16 + 8)));
  result[i] = (ok[i] != 0 && point_status[i] == CANON_FINITE && same) ? 1 : 0;
}

// scalar-field arithmetic modulo the group order (signing side): op 0 = a*b + c, 1 = a^-1
template <class N>
__global__ __launch_bounds__(TPB, 2) void k_canon_scalar_op(int op, const u32* __restrict__ a, const u32* __restrict__ b,
                                                            const u32* __restrict__ c, u32* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const fe x = canon::ld8(a + i * 8);
  const fe y = b ? canon::ld8(b + i * 8) : fe_zero();
  const fe z = c ? canon::ld8(c + i * 8) : fe_zero();
  canon::st8(out + i * 8, canon::scalar_op<N>(op, x, y, z));
}

}  // namespace fecgpu
