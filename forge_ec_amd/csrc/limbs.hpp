// limbs.hpp -- 256-bit limb primitives for gfx950 (CDNA4) integer VALU.
//
// Representation: a field element / scalar is 8 x uint32 little-endian words held in VGPRs (one
// scalar-mul per lane, 64 per wavefront).  The reference's u64 limb i is the word pair
// (w[2i], w[2i+1]).  All loops are fully unrolled with compile-time indices so nothing lands in
// scratch.
//
// Per-lane predicates are LANE MASKS (`lmask`, one bit per lane, held in an SGPR pair): carry
// chains export their carry-out as a mask, masks are combined with scalar ALU ops, selects
// consume them with VOP3 v_cndmask, and rare paths branch on `mask != 0` (wave-uniform).
//
// Why inline asm for carry chains and selects (measured on MI355X, profiles/valu_rates*_r01.txt):
//  * v_mad_u64_u32, v_add_co/v_addc_co, v_cmp and 3-operand VOP3 integer ops all issue at ~4.5
//    cycles per wave-instruction, plain VOP2 at ~2.5: the cost model is "count instructions".
//  * VOP2 `v_cndmask_b32_e32 ..., vcc` -- what hipcc emits for every select whose condition sits
//    in VCC -- measures 23 cycles; the VOP3 form on an SGPR pair 4.6.
//  * hipcc pads every VCC write->read with s_nop (6.3 vs 4.4 cycles per chained v_addc); a
//    2*10^10-chain stress of nop-less v_addc/v_subb chains (tools/microbench/valu_rates3.hip)
//    found no mismatch at 1..8 waves/SIMD.
//  * ROCm 7.2 hipcc MISCOMPILES a __builtin_subc chain followed by a __builtin_addc chain: it
//    fuses the top words into v_addc(x, borrow ? -1 : 0, carry), whose carry-out is wrong
//    whenever the borrow reaches the top word (caught by the edge-operand parity test).
#pragma once
#include <stdint.h>
#ifdef FEC_HOST_EMUL
// Host emulation of the device headers (tools/host_emul.cpp): logic checks and sanitizers on the
// CPU build; never part of the shipped library.  One "lane": lmask is 0 or all-ones.
#define FEC_DEV static inline
#define FEC_DEV_NOINLINE static __attribute__((noinline))
#else
#include <hip/hip_runtime.h>
#include "field_asm.inc"  // generated asm statements of the field multiplications (tools/gen_field_asm.py)
#define FEC_DEV __device__ __forceinline__
#define FEC_DEV_NOINLINE __device__ __attribute__((noinline))
#endif

namespace fecgpu {

typedef uint32_t u32;
typedef uint64_t u64;
typedef u64 lmask;  // one bit per lane

// scalars are staged in LDS word-major: word k of this lane's scalar is kw[k * KSTRIDE]
constexpr int KSTRIDE = 256;

struct fe {
  u32 w[8];
};

#define FEC_UNROLL _Pragma("unroll")
// Region markers for tools/isa_mix.py --region: comment-only asm statements, compiled in ONLY for the analysis build
// (-DFEC_ISA_MARKERS); the shipped library does not contain them.
#ifdef FEC_ISA_MARKERS
#define FEC_MARK(name) asm volatile("; FEC_MARK " name)
#else
#define FEC_MARK(name) \
  do {                 \
  } while (0)
#endif

#ifdef FEC_HOST_EMUL
FEC_DEV lmask lanes_where(bool c) { return c ? ~0ull : 0ull; }
#else
FEC_DEV lmask lanes_where(bool c) { return __builtin_amdgcn_ballot_w64(c); }
#endif

// high 64 bits of a 64 x 64 product
#ifdef FEC_HOST_EMUL
FEC_DEV u64 mulhi64(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) >> 64); }
#else
FEC_DEV u64 mulhi64(u64 a, u64 b) { return __umul64hi(a, b); }
#endif

FEC_DEV fe fe_zero() {
  fe r;
  FEC_UNROLL for (int i = 0; i < 8; ++i) r.w[i] = 0;
  return r;
}
FEC_DEV fe fe_small(u32 x) {
  fe r = fe_zero();
  r.w[0] = x;
  return r;
}
FEC_DEV lmask fe_eq(const fe& a, const fe& b) {
  u32 d = 0;
  FEC_UNROLL for (int i = 0; i < 8; ++i) d |= a.w[i] ^ b.w[i];
  return lanes_where(d == 0);
}
FEC_DEV lmask fe_is_zero(const fe& a) {
  u32 d = 0;
  FEC_UNROLL for (int i = 0; i < 8; ++i) d |= a.w[i];
  return lanes_where(d == 0);
}

// ------------------------------------------------------------------------------------------
// carry chains and selects
// ------------------------------------------------------------------------------------------
#ifdef FEC_HOST_EMUL

FEC_DEV lmask add256(fe& r, const fe& a, const fe& b) {
  u64 c = 0;
  for (int i = 0; i < 8; ++i) {
    u64 s = (u64)a.w[i] + b.w[i] + c;
    r.w[i] = (u32)s;
    c = s >> 32;
  }
  return c ? ~0ull : 0ull;
}
FEC_DEV lmask sub256(fe& r, const fe& a, const fe& b) {
  u64 bo = 0;
  for (int i = 0; i < 8; ++i) {
    u64 d = (u64)a.w[i] - b.w[i] - bo;
    r.w[i] = (u32)d;
    bo = (d >> 32) & 1;
  }
  return bo ? ~0ull : 0ull;
}
// r = a + a mod 2^256; returns the carry-out lane mask
FEC_DEV lmask dbl256(fe& r, const fe& a) { return add256(r, a, a); }
FEC_DEV lmask add256_cin(fe& r, const fe& a, const fe& b, lmask cin) {
  u64 c = cin ? 1 : 0;
  for (int i = 0; i < 8; ++i) {
    u64 s = (u64)a.w[i] + b.w[i] + c;
    r.w[i] = (u32)s;
    c = s >> 32;
  }
  return c ? ~0ull : 0ull;
}
// r = a + b mod 2^256, top += carry-out;  r = a - b mod 2^256, top -= borrow-out
FEC_DEV void add256c(fe& r, const fe& a, const fe& b, u32& top) { top += add256(r, a, b) ? 1u : 0u; }
FEC_DEV void sub256c(fe& r, const fe& a, const fe& b, u32& top) { top -= sub256(r, a, b) ? 1u : 0u; }
#define FEC_SDEV static inline
FEC_DEV fe fe_k8(u32 k0, u32 k1, u32 k2, u32 k3, u32 k4, u32 k5, u32 k6, u32 k7) {
  fe k;
  k.w[0] = k0; k.w[1] = k1; k.w[2] = k2; k.w[3] = k3; k.w[4] = k4; k.w[5] = k5; k.w[6] = k6; k.w[7] = k7;
  return k;
}
#define FEC_K8(k0, k1, k2, k3, k4, k5, k6, k7) \
  fe_k8((u32)(k0), (u32)(k1), (u32)(k2), (u32)(k3), (u32)(k4), (u32)(k5), (u32)(k6), (u32)(k7))
#define FEC_ADDK256_(r, a, c, k0, k1, k2, k3, k4, k5, k6, k7) c = add256(r, a, FEC_K8(k0, k1, k2, k3, k4, k5, k6, k7))
#define FEC_SUBK256_(r, a, c, k0, k1, k2, k3, k4, k5, k6, k7) c = sub256(r, a, FEC_K8(k0, k1, k2, k3, k4, k5, k6, k7))
#define FEC_KSUB256_(r, a, c, k0, k1, k2, k3, k4, k5, k6, k7) c = sub256(r, FEC_K8(k0, k1, k2, k3, k4, k5, k6, k7), a)
// r = a - (hi:lo) mod 2^256 (lo, hi per-lane words)
FEC_DEV lmask sub_lohi256(fe& r, const fe& a, u32 lo, u32 hi) { return sub256(r, a, FEC_K8(lo, hi, 0, 0, 0, 0, 0, 0)); }
// r = a + (hi:lo) mod 2^256 (lo, hi per-lane words)
FEC_DEV lmask add_lohi256(fe& r, const fe& a, u32 lo, u32 hi) { return add256(r, a, FEC_K8(lo, hi, 0, 0, 0, 0, 0, 0)); }
// r = a + k mod 2^256 (k a per-lane word)
FEC_DEV lmask add_word256(fe& r, const fe& a, u32 k) { return add256(r, a, FEC_K8(k, 0, 0, 0, 0, 0, 0, 0)); }
// r = a - k mod 2^256 (k a per-lane word)
FEC_DEV lmask sub_word256(fe& r, const fe& a, u32 k) { return sub256(r, a, FEC_K8(k, 0, 0, 0, 0, 0, 0, 0)); }
// m ? b : a on single words
FEC_DEV u32 word_select(u32 a, u32 b, lmask m) { return m ? b : a; }
// r = m ? b : a   (subtle::ConditionallySelectable::conditional_select(a, b, choice))
FEC_DEV fe fe_select(const fe& a, const fe& b, lmask m) { return m ? b : a; }

#else  // ---- gfx950 ----

// All multi-instruction asm statements below are IN-PLACE: each output word is tied ("+v") to the
// input word of the same index, so no instruction can overwrite a register a later instruction
// still reads, and the allocator may reuse the first operand's registers when it is dead.
#define FEC_RW8(x) "+v"(x.w[0]), "+v"(x.w[1]), "+v"(x.w[2]), "+v"(x.w[3]), "+v"(x.w[4]), "+v"(x.w[5]), "+v"(x.w[6]), "+v"(x.w[7])
#define FEC_V8(x) "v"(x.w[0]), "v"(x.w[1]), "v"(x.w[2]), "v"(x.w[3]), "v"(x.w[4]), "v"(x.w[5]), "v"(x.w[6]), "v"(x.w[7])

// r = a + b mod 2^256; returns the carry-out lane mask.
FEC_DEV lmask add256(fe& r, const fe& a, const fe& b) {
  lmask c;
  fe x = a;
  asm("v_add_co_u32_e32 %0, vcc, %0, %9\n\t"
      "v_addc_co_u32_e32 %1, vcc, %1, %10, vcc\n\t"
      "v_addc_co_u32_e32 %2, vcc, %2, %11, vcc\n\t"
      "v_addc_co_u32_e32 %3, vcc, %3, %12, vcc\n\t"
      "v_addc_co_u32_e32 %4, vcc, %4, %13, vcc\n\t"
      "v_addc_co_u32_e32 %5, vcc, %5, %14, vcc\n\t"
      "v_addc_co_u32_e32 %6, vcc, %6, %15, vcc\n\t"
      "v_addc_co_u32_e32 %7, vcc, %7, %16, vcc\n\t"
      "s_mov_b64 %8, vcc"
      : FEC_RW8(x), "=s"(c)
      : FEC_V8(b)
      : "vcc");
  r = x;
  return c;
}
// r = a + a mod 2^256; returns the carry-out lane mask.  One operand list: with add256(r, a, a) the tied copy of a
// and a itself (still an input) have to sit in different registers, eight v_mov.
FEC_DEV lmask dbl256(fe& r, const fe& a) {
  lmask c;
  fe x = a;
  asm("v_add_co_u32_e32 %0, vcc, %0, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\t"
      "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc\n\t"
      "v_addc_co_u32_e32 %3, vcc, %3, %3, vcc\n\t"
      "v_addc_co_u32_e32 %4, vcc, %4, %4, vcc\n\t"
      "v_addc_co_u32_e32 %5, vcc, %5, %5, vcc\n\t"
      "v_addc_co_u32_e32 %6, vcc, %6, %6, vcc\n\t"
      "v_addc_co_u32_e32 %7, vcc, %7, %7, vcc\n\t"
      "s_mov_b64 %8, vcc"
      : FEC_RW8(x), "=s"(c)
      :
      : "vcc");
  r = x;
  return c;
}
// r = a - b mod 2^256; returns the borrow-out lane mask.
FEC_DEV lmask sub256(fe& r, const fe& a, const fe& b) {
  lmask c;
  fe x = a;
  asm("v_sub_co_u32_e32 %0, vcc, %0, %9\n\t"
      "v_subb_co_u32_e32 %1, vcc, %1, %10, vcc\n\t"
      "v_subb_co_u32_e32 %2, vcc, %2, %11, vcc\n\t"
      "v_subb_co_u32_e32 %3, vcc, %3, %12, vcc\n\t"
      "v_subb_co_u32_e32 %4, vcc, %4, %13, vcc\n\t"
      "v_subb_co_u32_e32 %5, vcc, %5, %14, vcc\n\t"
      "v_subb_co_u32_e32 %6, vcc, %6, %15, vcc\n\t"
      "v_subb_co_u32_e32 %7, vcc, %7, %16, vcc\n\t"
      "s_mov_b64 %8, vcc"
      : FEC_RW8(x), "=s"(c)
      : FEC_V8(b)
      : "vcc");
  r = x;
  return c;
}
// r = a + b mod 2^256 and top += carry-out, the carry absorbed by a ninth chain instruction
FEC_DEV void add256c(fe& r, const fe& a, const fe& b, u32& top) {
  fe x = a;
  asm("v_add_co_u32_e32 %0, vcc, %0, %9\n\t"
      "v_addc_co_u32_e32 %1, vcc, %1, %10, vcc\n\t"
      "v_addc_co_u32_e32 %2, vcc, %2, %11, vcc\n\t"
      "v_addc_co_u32_e32 %3, vcc, %3, %12, vcc\n\t"
      "v_addc_co_u32_e32 %4, vcc, %4, %13, vcc\n\t"
      "v_addc_co_u32_e32 %5, vcc, %5, %14, vcc\n\t"
      "v_addc_co_u32_e32 %6, vcc, %6, %15, vcc\n\t"
      "v_addc_co_u32_e32 %7, vcc, %7, %16, vcc\n\t"
      "v_addc_co_u32_e32 %8, vcc, 0, %8, vcc"
      : FEC_RW8(x), "+v"(top)
      : FEC_V8(b)
      : "vcc");
  r = x;
}
// r = a - b mod 2^256 and top -= borrow-out
FEC_DEV void sub256c(fe& r, const fe& a, const fe& b, u32& top) {
  fe x = a;
  asm("v_sub_co_u32_e32 %0, vcc, %0, %9\n\t"
      "v_subb_co_u32_e32 %1, vcc, %1, %10, vcc\n\t"
      "v_subb_co_u32_e32 %2, vcc, %2, %11, vcc\n\t"
      "v_subb_co_u32_e32 %3, vcc, %3, %12, vcc\n\t"
      "v_subb_co_u32_e32 %4, vcc, %4, %13, vcc\n\t"
      "v_subb_co_u32_e32 %5, vcc, %5, %14, vcc\n\t"
      "v_subb_co_u32_e32 %6, vcc, %6, %15, vcc\n\t"
      "v_subb_co_u32_e32 %7, vcc, %7, %16, vcc\n\t"
      "v_subbrev_co_u32_e32 %8, vcc, 0, %8, vcc"
      : FEC_RW8(x), "+v"(top)
      : FEC_V8(b)
      : "vcc");
  r = x;
}
#define FEC_SDEV __device__ __forceinline__ static
// r = a + K, K = {k0..k7} compile-time words riding in the VOP2 src0 slot (no registers).  k0 may
// be any 32-bit literal; k1..k7 must be inline constants (-16..64): a literal plus the VCC
// carry-in would be two constant-bus reads.  c receives the carry-out mask.
#define FEC_ADDK256_(r, a, c, k0, k1, k2, k3, k4, k5, k6, k7) \
  do { \
    fe fec_x_ = (a); \
    asm("v_add_co_u32_e32 %0, vcc, " #k0 ", %0\n\t" \
        "v_addc_co_u32_e32 %1, vcc, " #k1 ", %1, vcc\n\t" \
        "v_addc_co_u32_e32 %2, vcc, " #k2 ", %2, vcc\n\t" \
        "v_addc_co_u32_e32 %3, vcc, " #k3 ", %3, vcc\n\t" \
        "v_addc_co_u32_e32 %4, vcc, " #k4 ", %4, vcc\n\t" \
        "v_addc_co_u32_e32 %5, vcc, " #k5 ", %5, vcc\n\t" \
        "v_addc_co_u32_e32 %6, vcc, " #k6 ", %6, vcc\n\t" \
        "v_addc_co_u32_e32 %7, vcc, " #k7 ", %7, vcc\n\t" \
        "s_mov_b64 %8, vcc" \
        : FEC_RW8(fec_x_), "=s"(c) : : "vcc"); \
    (r) = fec_x_; \
  } while (0)
// r = a - K; c receives the borrow-out mask.
#define FEC_SUBK256_(r, a, c, k0, k1, k2, k3, k4, k5, k6, k7) \
  do { \
    fe fec_x_ = (a); \
    asm("v_subrev_co_u32_e32 %0, vcc, " #k0 ", %0\n\t" \
        "v_subbrev_co_u32_e32 %1, vcc, " #k1 ", %1, vcc\n\t" \
        "v_subbrev_co_u32_e32 %2, vcc, " #k2 ", %2, vcc\n\t" \
        "v_subbrev_co_u32_e32 %3, vcc, " #k3 ", %3, vcc\n\t" \
        "v_subbrev_co_u32_e32 %4, vcc, " #k4 ", %4, vcc\n\t" \
        "v_subbrev_co_u32_e32 %5, vcc, " #k5 ", %5, vcc\n\t" \
        "v_subbrev_co_u32_e32 %6, vcc, " #k6 ", %6, vcc\n\t" \
        "v_subbrev_co_u32_e32 %7, vcc, " #k7 ", %7, vcc\n\t" \
        "s_mov_b64 %8, vcc" \
        : FEC_RW8(fec_x_), "=s"(c) : : "vcc"); \
    (r) = fec_x_; \
  } while (0)
// r = K - a; c receives the borrow-out mask.
#define FEC_KSUB256_(r, a, c, k0, k1, k2, k3, k4, k5, k6, k7) \
  do { \
    fe fec_x_ = (a); \
    asm("v_sub_co_u32_e32 %0, vcc, " #k0 ", %0\n\t" \
        "v_subb_co_u32_e32 %1, vcc, " #k1 ", %1, vcc\n\t" \
        "v_subb_co_u32_e32 %2, vcc, " #k2 ", %2, vcc\n\t" \
        "v_subb_co_u32_e32 %3, vcc, " #k3 ", %3, vcc\n\t" \
        "v_subb_co_u32_e32 %4, vcc, " #k4 ", %4, vcc\n\t" \
        "v_subb_co_u32_e32 %5, vcc, " #k5 ", %5, vcc\n\t" \
        "v_subb_co_u32_e32 %6, vcc, " #k6 ", %6, vcc\n\t" \
        "v_subb_co_u32_e32 %7, vcc, " #k7 ", %7, vcc\n\t" \
        "s_mov_b64 %8, vcc" \
        : FEC_RW8(fec_x_), "=s"(c) : : "vcc"); \
    (r) = fec_x_; \
  } while (0)
// r = a - (hi:lo) mod 2^256 (lo, hi per-lane words); returns the borrow-out mask.
FEC_DEV lmask sub_lohi256(fe& r, const fe& a, u32 lo, u32 hi) {
  lmask c;
  fe x = a;
  asm("v_sub_co_u32_e32 %0, vcc, %0, %9\n\t"
      "v_subb_co_u32_e32 %1, vcc, %1, %10, vcc\n\t"
      "v_subbrev_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
      "v_subbrev_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
      "v_subbrev_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
      "v_subbrev_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
      "v_subbrev_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
      "v_subbrev_co_u32_e32 %7, vcc, 0, %7, vcc\n\t"
      "s_mov_b64 %8, vcc"
      : FEC_RW8(x), "=s"(c)
      : "v"(lo), "v"(hi)
      : "vcc");
  r = x;
  return c;
}
// r = a + (hi:lo) mod 2^256 (lo, hi per-lane words); returns the carry-out mask.
FEC_DEV lmask add_lohi256(fe& r, const fe& a, u32 lo, u32 hi) {
  lmask c;
  fe x = a;
  asm("v_add_co_u32_e32 %0, vcc, %0, %9\n\t"
      "v_addc_co_u32_e32 %1, vcc, %1, %10, vcc\n\t"
      "v_addc_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
      "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
      "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
      "v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
      "v_addc_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
      "v_addc_co_u32_e32 %7, vcc, 0, %7, vcc\n\t"
      "s_mov_b64 %8, vcc"
      : FEC_RW8(x), "=s"(c)
      : "v"(lo), "v"(hi)
      : "vcc");
  r = x;
  return c;
}
// r = a + k mod 2^256 (k a per-lane word); returns the carry-out mask.
FEC_DEV lmask add_word256(fe& r, const fe& a, u32 k) {
  lmask c;
  fe x = a;
  asm("v_add_co_u32_e32 %0, vcc, %0, %9\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_addc_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
      "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
      "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
      "v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
      "v_addc_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
      "v_addc_co_u32_e32 %7, vcc, 0, %7, vcc\n\t"
      "s_mov_b64 %8, vcc"
      : FEC_RW8(x), "=s"(c)
      : "v"(k)
      : "vcc");
  r = x;
  return c;
}
// r = a - k mod 2^256 (k a per-lane word); returns the borrow-out mask.
FEC_DEV lmask sub_word256(fe& r, const fe& a, u32 k) {
  lmask c;
  fe x = a;
  asm("v_sub_co_u32_e32 %0, vcc, %0, %9\n\t"
      "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_subbrev_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
      "v_subbrev_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
      "v_subbrev_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
      "v_subbrev_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
      "v_subbrev_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
      "v_subbrev_co_u32_e32 %7, vcc, 0, %7, vcc\n\t"
      "s_mov_b64 %8, vcc"
      : FEC_RW8(x), "=s"(c)
      : "v"(k)
      : "vcc");
  r = x;
  return c;
}
// m ? b : a on single words
FEC_DEV u32 word_select(u32 a, u32 b, lmask m) {
  u32 r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
  return r;
}

// r = m ? b : a per lane (subtle::conditional_select(a, b, choice)); VOP3 v_cndmask on the mask.
FEC_DEV fe fe_select(const fe& a, const fe& b, lmask m) {
  fe r = a;
  asm("v_cndmask_b32_e64 %0, %0, %8, %16\n\t"
      "v_cndmask_b32_e64 %1, %1, %9, %16\n\t"
      "v_cndmask_b32_e64 %2, %2, %10, %16\n\t"
      "v_cndmask_b32_e64 %3, %3, %11, %16\n\t"
      "v_cndmask_b32_e64 %4, %4, %12, %16\n\t"
      "v_cndmask_b32_e64 %5, %5, %13, %16\n\t"
      "v_cndmask_b32_e64 %6, %6, %14, %16\n\t"
      "v_cndmask_b32_e64 %7, %7, %15, %16"
      : FEC_RW8(r)
      : FEC_V8(b), "s"(m));
  return r;
}
#endif

// The compiler treats the mask outputs of the multi-output asm chains as divergent values and may
// evaluate boolean combinations of them on the VALU; uniform_mask() pins such a combination back
// into SGPRs (it IS wave-uniform) before it is used as an "s" operand.  Only needed for
// combinations the instruction selector turns into VALU-only forms (and-not chains -> v_bfi).
#ifdef FEC_HOST_EMUL
FEC_DEV lmask uniform_mask(lmask m) { return m; }
#else
FEC_DEV lmask uniform_mask(lmask m) {
  u32 lo = __builtin_amdgcn_readfirstlane((u32)m), hi = __builtin_amdgcn_readfirstlane((u32)(m >> 32));
  return ((u64)hi << 32) | lo;
}
#endif

// this lane's bit of a lane mask.  Goes through word_select so that the mask stays in SGPRs: a
// 64-bit shift by the lane id would let the compiler move the whole mask computation to the VALU,
// and the "s" operands of the asm selects above would then be handed VGPRs.
FEC_DEV bool lane_of(lmask m) { return word_select(0u, 1u, uniform_mask(m)) != 0; }

// ---- short carry chains -------------------------------------------------------------------------------------
// Adding or subtracting a one- or two-word constant on the lanes of a mask touches words 0..1; the carry (borrow)
// leaves them only when the word it enters is all ones (zero): ~2^-32 per lane.  The chain therefore stops after the
// constant's words and hands the outgoing carry back as a lane mask; the caller continues it through the remaining
// words behind a wave-uniform branch (carry_from / borrow_from).  Exact for every operand.
#ifdef FEC_HOST_EMUL
// x.w[0..1] += m ? (1:k0) : 0; returns the carry out of word 1
FEC_DEV lmask add_short2(fe& x, lmask m, u32 k0) {
  if (!m) return 0;
  u64 s = (u64)x.w[0] + k0;
  x.w[0] = (u32)s;
  s = (u64)x.w[1] + 1u + (s >> 32);
  x.w[1] = (u32)s;
  return (s >> 32) ? ~0ull : 0ull;
}
// x.w[0..1] -= m ? (1:k0) : 0; returns the borrow out of word 1
FEC_DEV lmask sub_short2(fe& x, lmask m, u32 k0) {
  if (!m) return 0;
  u64 d = (u64)x.w[0] - k0;
  x.w[0] = (u32)d;
  d = (u64)x.w[1] - 1u - ((d >> 32) & 1);
  x.w[1] = (u32)d;
  return ((d >> 32) & 1) ? ~0ull : 0ull;
}
#define FEC_ADD_SHORT2(x, m, cy, k0) cy = add_short2(x, m, (u32)(k0))
#define FEC_SUB_SHORT2(x, m, bw, k0) bw = sub_short2(x, m, (u32)(k0))
// x.w[0] += k (a per-lane word); returns the carry out of word 0
FEC_DEV lmask add_short1(fe& x, u32 k) {
  u64 s = (u64)x.w[0] + k;
  x.w[0] = (u32)s;
  return (s >> 32) ? ~0ull : 0ull;
}
// x.w[0] -= k (a per-lane word); returns the borrow out of word 0
FEC_DEV lmask sub_short1(fe& x, u32 k) {
  u64 d = (u64)x.w[0] - k;
  x.w[0] = (u32)d;
  return ((d >> 32) & 1) ? ~0ull : 0ull;
}
// x += 2^(32 F) on the lanes of m, wrapping at 2^256;  x -= 2^(32 F) likewise
template <int F>
FEC_DEV void carry_from(fe& x, lmask m) {
  u64 c = m ? 1 : 0;
  for (int i = F; i < 8; ++i) {
    u64 s = (u64)x.w[i] + c;
    x.w[i] = (u32)s;
    c = s >> 32;
  }
}
template <int F>
FEC_DEV void borrow_from(fe& x, lmask m) {
  u64 b = m ? 1 : 0;
  for (int i = F; i < 8; ++i) {
    u64 d = (u64)x.w[i] - b;
    x.w[i] = (u32)d;
    b = (d >> 32) & 1;
  }
}
#else
// k0: any 32-bit literal (it rides in a VOP2 literal slot); the constant's word 1 is 1
#define FEC_ADD_SHORT2(x, m, cy, k0) \
  do { \
    u32 fec_t0_, fec_t1_; \
    asm("v_cndmask_b32_e64 %3, 0, 1, %5\n\t" \
        "v_mul_u32_u24_e32 %4, " #k0 ", %3\n\t" \
        "v_add_co_u32_e32 %0, vcc, %0, %4\n\t" \
        "v_addc_co_u32_e32 %1, vcc, %1, %3, vcc\n\t" \
        "s_mov_b64 %2, vcc" \
        : "+v"((x).w[0]), "+v"((x).w[1]), "=s"(cy), "=&v"(fec_t1_), "=&v"(fec_t0_) : "s"(m) : "vcc"); \
  } while (0)
#define FEC_SUB_SHORT2(x, m, bw, k0) \
  do { \
    u32 fec_t0_, fec_t1_; \
    asm("v_cndmask_b32_e64 %3, 0, 1, %5\n\t" \
        "v_mul_u32_u24_e32 %4, " #k0 ", %3\n\t" \
        "v_sub_co_u32_e32 %0, vcc, %0, %4\n\t" \
        "v_subb_co_u32_e32 %1, vcc, %1, %3, vcc\n\t" \
        "s_mov_b64 %2, vcc" \
        : "+v"((x).w[0]), "+v"((x).w[1]), "=s"(bw), "=&v"(fec_t1_), "=&v"(fec_t0_) : "s"(m) : "vcc"); \
  } while (0)
FEC_DEV lmask add_short1(fe& x, u32 k) {
  lmask c;
  asm("v_add_co_u32_e32 %0, vcc, %0, %2\n\t"
      "s_mov_b64 %1, vcc"
      : "+v"(x.w[0]), "=s"(c) : "v"(k) : "vcc");
  return c;
}
FEC_DEV lmask sub_short1(fe& x, u32 k) {
  lmask c;
  asm("v_sub_co_u32_e32 %0, vcc, %0, %2\n\t"
      "s_mov_b64 %1, vcc"
      : "+v"(x.w[0]), "=s"(c) : "v"(k) : "vcc");
  return c;
}
// the continuation as ONE statement (VCC must not be touched between the chain's instructions)
template <int F>
FEC_DEV void carry_from(fe& x, lmask m) {
  static_assert(F == 1 || F == 2, "continuations start at word 1 or word 2");
  if (F == 1)
    asm("s_mov_b64 vcc, %7\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
        "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
        "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
        "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
        "v_addc_co_u32_e32 %6, vcc, 0, %6, vcc"
        : "+v"(x.w[1]), "+v"(x.w[2]), "+v"(x.w[3]), "+v"(x.w[4]), "+v"(x.w[5]), "+v"(x.w[6]), "+v"(x.w[7])
        : "s"(m) : "vcc");
  else
    asm("s_mov_b64 vcc, %6\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
        "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
        "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
        "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32_e32 %5, vcc, 0, %5, vcc"
        : "+v"(x.w[2]), "+v"(x.w[3]), "+v"(x.w[4]), "+v"(x.w[5]), "+v"(x.w[6]), "+v"(x.w[7])
        : "s"(m) : "vcc");
}
template <int F>
FEC_DEV void borrow_from(fe& x, lmask m) {
  static_assert(F == 1 || F == 2, "continuations start at word 1 or word 2");
  if (F == 1)
    asm("s_mov_b64 vcc, %7\n\t"
        "v_subbrev_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "v_subbrev_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
        "v_subbrev_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
        "v_subbrev_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
        "v_subbrev_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
        "v_subbrev_co_u32_e32 %6, vcc, 0, %6, vcc"
        : "+v"(x.w[1]), "+v"(x.w[2]), "+v"(x.w[3]), "+v"(x.w[4]), "+v"(x.w[5]), "+v"(x.w[6]), "+v"(x.w[7])
        : "s"(m) : "vcc");
  else
    asm("s_mov_b64 vcc, %6\n\t"
        "v_subbrev_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "v_subbrev_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
        "v_subbrev_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
        "v_subbrev_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
        "v_subbrev_co_u32_e32 %5, vcc, 0, %5, vcc"
        : "+v"(x.w[2]), "+v"(x.w[3]), "+v"(x.w[4]), "+v"(x.w[5]), "+v"(x.w[6]), "+v"(x.w[7])
        : "s"(m) : "vcc");
}
#endif

// one level of indirection so that a constant list passed as a single macro (FEC_SECP_C, ...)
// is expanded before it is split into k0..k7
#define FEC_ADDK256(r, a, c, ...) FEC_ADDK256_(r, a, c, __VA_ARGS__)
#define FEC_SUBK256(r, a, c, ...) FEC_SUBK256_(r, a, c, __VA_ARGS__)
#define FEC_KSUB256(r, a, c, ...) FEC_KSUB256_(r, a, c, __VA_ARGS__)

// 32x32 + 32 + 32 -> 64 (never overflows)
FEC_DEV u64 mad2(u32 a, u32 b, u32 c, u32 d) { return (u64)a * b + c + d; }

// t[0..15] = a * b, exact 512-bit product.
#ifdef FEC_HOST_EMUL
FEC_DEV void mul_wide(u32 t[16], const fe& a, const fe& b) {
  for (int i = 0; i < 16; ++i) t[i] = 0;
  for (int i = 0; i < 8; ++i) {
    u32 carry = 0;
    for (int j = 0; j < 8; ++j) {
      u64 p = mad2(a.w[i], b.w[j], t[i + j], carry);
      t[i + j] = (u32)p;
      carry = (u32)(p >> 32);
    }
    t[i + 8] = carry;
  }
}
#else
// Product scanning: column k accumulates its (up to 8) partial products into a 96-bit
// accumulator {acc (64-bit VGPR pair), ovf}.  Each product is TWO instructions:
//   v_mad_u64_u32 acc, vcc, a_i, b_j, acc      ; acc += a_i*b_j, carry-out -> vcc
//   v_addc_co_u32 ovf, vcc, 0, ovf, vcc        ; ovf += carry
// The accumulator is only ever written by the mad (as a 64-bit pair) and only read in halves, so
// no 64-bit pair has to be assembled from separate registers inside a column -- that assembly
// cost hipcc's own lowering of the same arithmetic 179 v_mov per multiplication.
FEC_DEV void mac96(u64& acc, u32& ovf, u32 x, u32 y) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
      : "+v"(acc), "+v"(ovf)
      : "v"(x), "v"(y)
      : "vcc");
}
// First product of a column: a FRESH accumulator {acc, ovf} = x*y + cin.  Starting every column
// in new registers leaves the finished word t[k] where the previous column's mad put it and needs
// no re-zeroing of the overflow word; only the carry-in pair {acc_prev.hi, ovf_prev} is assembled
// (2 v_mov per column instead of 4).  All inputs are consumed by the first instruction.
FEC_DEV void mac96_first(u64& acc, u32& ovf, u32 x, u32 y, u64 cin) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc"
      : "=v"(acc), "=v"(ovf)
      : "v"(x), "v"(y), "v"(cin)
      : "vcc");
}
#ifdef FEC_MAC_PER_PRODUCT
FEC_DEV void mul_wide(u32 t[16], const fe& a, const fe& b) {
  u64 acc = (u64)a.w[0] * b.w[0];  // column 0: one product, cannot overflow
  u32 ovf = 0;
  t[0] = (u32)acc;
  FEC_UNROLL for (int k = 1; k < 15; ++k) {
    u64 cin = (acc >> 32) | ((u64)ovf << 32);
    bool first = true;
    FEC_UNROLL for (int i = 0; i < 8; ++i) {
      const int j = k - i;
      if (j >= 0 && j < 8) {
        if (first) {
          u64 nacc;
          u32 novf;
          mac96_first(nacc, novf, a.w[i], b.w[j], cin);
          acc = nacc;
          ovf = novf;
          first = false;
        } else {
          mac96(acc, ovf, a.w[i], b.w[j]);
        }
      }
    }
    t[k] = (u32)acc;
  }
  t[15] = (u32)(acc >> 32);
}
#else
// A whole column in ONE asm statement: {acc, ovf} = cin + sum of N products.  Keeping the column's
// mad/addc pairs inside one statement keeps the compiler's hazard padding (an s_nop after every
// asm statement that clobbers VCC) to one per column instead of one per product.  acc/ovf are
// early-clobber: the first instruction writes them while later ones still read their operands.
template <int N>
FEC_DEV void mcol(u64& acc, u32& ovf, u64 cin, const u32* x, const u32* y);
template <>
FEC_DEV void mcol<1>(u64& acc, u32& ovf, u64 cin, const u32* x, const u32* y) {
  asm("v_mad_u64_u32 %0, vcc, %3, %4, %2\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc"
      : "=&v"(acc), "=&v"(ovf)
      : "v"(cin), "v"(x[0]), "v"(y[0])
      : "vcc");
}
template <>
FEC_DEV void mcol<2>(u64& acc, u32& ovf, u64 cin, const u32* x, const u32* y) {
  asm("v_mad_u64_u32 %0, vcc, %3, %4, %2\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
      : "=&v"(acc), "=&v"(ovf)
      : "v"(cin), "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1])
      : "vcc");
}
template <>
FEC_DEV void mcol<3>(u64& acc, u32& ovf, u64 cin, const u32* x, const u32* y) {
  asm("v_mad_u64_u32 %0, vcc, %3, %4, %2\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
      : "=&v"(acc), "=&v"(ovf)
      : "v"(cin), "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2])
      : "vcc");
}
template <>
FEC_DEV void mcol<4>(u64& acc, u32& ovf, u64 cin, const u32* x, const u32* y) {
  asm("v_mad_u64_u32 %0, vcc, %3, %4, %2\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
      : "=&v"(acc), "=&v"(ovf)
      : "v"(cin), "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3])
      : "vcc");
}
template <>
FEC_DEV void mcol<5>(u64& acc, u32& ovf, u64 cin, const u32* x, const u32* y) {
  asm("v_mad_u64_u32 %0, vcc, %3, %4, %2\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
      : "=&v"(acc), "=&v"(ovf)
      : "v"(cin), "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4])
      : "vcc");
}
template <>
FEC_DEV void mcol<6>(u64& acc, u32& ovf, u64 cin, const u32* x, const u32* y) {
  asm("v_mad_u64_u32 %0, vcc, %3, %4, %2\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
      : "=&v"(acc), "=&v"(ovf)
      : "v"(cin), "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5])
      : "vcc");
}
template <>
FEC_DEV void mcol<7>(u64& acc, u32& ovf, u64 cin, const u32* x, const u32* y) {
  asm("v_mad_u64_u32 %0, vcc, %3, %4, %2\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
      : "=&v"(acc), "=&v"(ovf)
      : "v"(cin), "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6])
      : "vcc");
}
template <>
FEC_DEV void mcol<8>(u64& acc, u32& ovf, u64 cin, const u32* x, const u32* y) {
  asm("v_mad_u64_u32 %0, vcc, %3, %4, %2\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
      : "=&v"(acc), "=&v"(ovf)
      : "v"(cin), "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7])
      : "vcc");
}
FEC_DEV void mul_wide(u32 t[16], const fe& a, const fe& b) {
  u64 acc = (u64)a.w[0] * b.w[0];  // column 0: one product, cannot overflow
  u32 ovf = 0;
  t[0] = (u32)acc;
  FEC_UNROLL for (int k = 1; k < 15; ++k) {
    const u64 cin = (acc >> 32) | ((u64)ovf << 32);
    const int lo = k < 8 ? 0 : k - 7, n = (k < 8 ? k : 7) - lo + 1;
    u32 x[8], y[8];
    FEC_UNROLL for (int i = 0; i < 8; ++i) {
      x[i] = i < n ? a.w[lo + i] : 0;
      y[i] = i < n ? b.w[k - lo - i] : 0;
    }
    switch (n) {
      case 1: mcol<1>(acc, ovf, cin, x, y); break;
      case 2: mcol<2>(acc, ovf, cin, x, y); break;
      case 3: mcol<3>(acc, ovf, cin, x, y); break;
      case 4: mcol<4>(acc, ovf, cin, x, y); break;
      case 5: mcol<5>(acc, ovf, cin, x, y); break;
      case 6: mcol<6>(acc, ovf, cin, x, y); break;
      case 7: mcol<7>(acc, ovf, cin, x, y); break;
      default: mcol<8>(acc, ovf, cin, x, y); break;
    }
    t[k] = (u32)acc;
  }
  t[15] = (u32)(acc >> 32);
}
#endif
#endif

// t[0..15] = a^2, exact.  28 cross products a_i a_j (i < j) by product scanning, doubled with
// v_alignbit, plus the eight squares: 56 + 16 + 8 + 17 instructions against 128 for mul_wide(a, a).
#ifdef FEC_HOST_EMUL
FEC_DEV void sqr_wide(u32 t[16], const fe& a) { mul_wide(t, a, a); }
#else
// r = a + b + carry-in mask; returns the carry-out mask
FEC_DEV lmask add256_cin(fe& r, const fe& a, const fe& b, lmask cin) {
  lmask c;
  fe x = a;
  asm("s_mov_b64 vcc, %17\n\t"
      "v_addc_co_u32_e32 %0, vcc, %0, %9, vcc\n\t"
      "v_addc_co_u32_e32 %1, vcc, %1, %10, vcc\n\t"
      "v_addc_co_u32_e32 %2, vcc, %2, %11, vcc\n\t"
      "v_addc_co_u32_e32 %3, vcc, %3, %12, vcc\n\t"
      "v_addc_co_u32_e32 %4, vcc, %4, %13, vcc\n\t"
      "v_addc_co_u32_e32 %5, vcc, %5, %14, vcc\n\t"
      "v_addc_co_u32_e32 %6, vcc, %6, %15, vcc\n\t"
      "v_addc_co_u32_e32 %7, vcc, %7, %16, vcc\n\t"
      "s_mov_b64 %8, vcc"
      : FEC_RW8(x), "=s"(c)
      : FEC_V8(b), "s"(cin)
      : "vcc");
  r = x;
  return c;
}
FEC_DEV u64 sq64(u32 x) {
  u64 r;
  asm("v_mad_u64_u32 %0, vcc, %1, %1, 0" : "=v"(r) : "v"(x) : "vcc");
  return r;
}
FEC_DEV void sqr_wide(u32 t[16], const fe& a) {
  u32 x[16];
  x[0] = 0;
  u64 acc = (u64)a.w[0] * a.w[1];  // column 1: one product
  u32 ovf = 0;
  x[1] = (u32)acc;
  FEC_UNROLL for (int k = 2; k < 14; ++k) {
    const u64 cin = (acc >> 32) | ((u64)ovf << 32);
    const int lo = k < 8 ? 0 : k - 7, hi = (k - 1) / 2, n = hi - lo + 1;  // pairs (i, k - i), lo <= i <= hi
    u32 xs[4], ys[4];
    FEC_UNROLL for (int i = 0; i < 4; ++i) {
      xs[i] = i < n ? a.w[lo + i] : 0;
      ys[i] = i < n ? a.w[k - lo - i] : 0;
    }
    switch (n) {
      case 1: mcol<1>(acc, ovf, cin, xs, ys); break;
      case 2: mcol<2>(acc, ovf, cin, xs, ys); break;
      case 3: mcol<3>(acc, ovf, cin, xs, ys); break;
      default: mcol<4>(acc, ovf, cin, xs, ys); break;
    }
    x[k] = (u32)acc;
  }
  x[14] = (u32)(acc >> 32);
  x[15] = ovf;
  fe y0, y1, d0, d1;
  y0.w[0] = 0;
  FEC_UNROLL for (int k = 1; k < 8; ++k) y0.w[k] = __builtin_amdgcn_alignbit(x[k], x[k - 1], 31);
  FEC_UNROLL for (int k = 8; k < 16; ++k) y1.w[k - 8] = __builtin_amdgcn_alignbit(x[k], x[k - 1], 31);
  FEC_UNROLL for (int i = 0; i < 4; ++i) {
    const u64 s0 = sq64(a.w[i]), s1 = sq64(a.w[4 + i]);
    d0.w[2 * i] = (u32)s0;
    d0.w[2 * i + 1] = (u32)(s0 >> 32);
    d1.w[2 * i] = (u32)s1;
    d1.w[2 * i + 1] = (u32)(s1 >> 32);
  }
  fe r0, r1;
  const lmask c = add256(r0, y0, d0);
  add256_cin(r1, y1, d1, c);
  FEC_UNROLL for (int i = 0; i < 8; ++i) {
    t[i] = r0.w[i];
    t[8 + i] = r1.w[i];
  }
}
#endif

// t[0..7] = low 256 bits of a * b (columns 0..7 of the same product scanning)
#ifdef FEC_HOST_EMUL
FEC_DEV void mul_low256(u32 t[8], const fe& a, const fe& b) {
  u32 w[16];
  mul_wide(w, a, b);
  for (int i = 0; i < 8; ++i) t[i] = w[i];
}
#else
FEC_DEV void mul_low256(u32 t[8], const fe& a, const fe& b) {
  u64 acc = (u64)a.w[0] * b.w[0];
  u32 ovf = 0;
  t[0] = (u32)acc;
  FEC_UNROLL for (int k = 1; k < 8; ++k) {
    u64 cin = (acc >> 32) | ((u64)ovf << 32);
    bool first = true;
    FEC_UNROLL for (int i = 0; i <= k; ++i) {
      const int j = k - i;
      if (first) {
        u64 nacc;
        u32 novf;
        mac96_first(nacc, novf, a.w[i], b.w[j], cin);
        acc = nacc;
        ovf = novf;
        first = false;
      } else {
        mac96(acc, ovf, a.w[i], b.w[j]);
      }
    }
    t[k] = (u32)acc;
  }
}
#endif

// t[0..8] = a * k (k a 32-bit constant), t[9..15] = 0
FEC_DEV void mul_wide_small(u32 t[16], const fe& a, u32 k) {
  u32 carry = 0;
  FEC_UNROLL for (int j = 0; j < 8; ++j) {
    u64 p = (u64)a.w[j] * k + carry;
    t[j] = (u32)p;
    carry = (u32)(p >> 32);
  }
  t[8] = carry;
  FEC_UNROLL for (int j = 9; j < 16; ++j) t[j] = 0;
}

// p[0..3] = (x1:x0) * (y1:y0), exact 128-bit product as four words (product scanning: the first
// product of a column cannot overflow the 64-bit accumulator, and the top column is bounded by
// the true product, so only one overflow count is needed).
#ifdef FEC_HOST_EMUL
FEC_DEV void mul64_words(u32 p[4], u32 x0, u32 x1, u32 y0, u32 y1) {
  unsigned __int128 v = (unsigned __int128)(((u64)x1 << 32) | x0) * (((u64)y1 << 32) | y0);
  for (int i = 0; i < 4; ++i) p[i] = (u32)(v >> (32 * i));
}
#else
FEC_DEV void mul64_words(u32 p[4], u32 x0, u32 x1, u32 y0, u32 y1) {
  u64 c0 = (u64)x0 * y0;
  p[0] = (u32)c0;
  u64 acc = (u64)x0 * y1 + (c0 >> 32);
  u32 ovf = 0;
  mac96(acc, ovf, x1, y0);
  p[1] = (u32)acc;
  u64 top = (u64)x1 * y1 + ((acc >> 32) | ((u64)ovf << 32));
  p[2] = (u32)top;
  p[3] = (u32)(top >> 32);
}
#endif

// 64x64 -> 128 on 32-bit words: (lo, hi) of (a1:a0) * (b1:b0)
FEC_DEV void mul64wide(u32 a0, u32 a1, u32 b0, u32 b1, u64& lo, u64& hi) {
  u64 p00 = (u64)a0 * b0;
  u64 p01 = (u64)a0 * b1 + (u32)(p00 >> 32);
  u64 p10 = mad2(a1, b0, (u32)p01, 0);
  u64 p11 = mad2(a1, b1, (u32)(p01 >> 32), (u32)(p10 >> 32));
  lo = (u64)(u32)p00 | ((u64)(u32)p10 << 32);
  hi = p11;
}

FEC_DEV void set_limb64(fe& a, int i, u64 v) {
  a.w[2 * i] = (u32)v;
  a.w[2 * i + 1] = (u32)(v >> 32);
}

}  // namespace fecgpu
