// kernels_p256.hip -- P-256 Curve::multiply (p256.rs:2120-2156) as a workgroup-level task scheduler.
//
//   for i in 0..256 { result = result.double(); if bit(255 - i) == 1 { result = result + point } }
//
// Every step doubles, but only the lanes whose bit is set need the (twice as expensive) addition.
// Executing the addition for whole wavefronts under a mask wastes half of it; a step-synchronous
// compaction (round 1) packs it but leaves SIMDs idle at three barriers per step (PMC: 40 % of wave
// cycles parked, VALUBusy 82 %, profiles/pmc_r02_base_p256).  Here a workgroup keeps 864 elements'
// state in LDS slots, and its wavefronts pull BATCHES from two ready queues: 64 elements whose next step has a
// clear scalar bit (a doubling), or 64 elements whose next step has a set bit (a doubling and then the addition).  Elements therefore
// advance at their own pace (each one still sees exactly the reference's operation sequence, so
// results are bit-identical), every batch is full until the workgroup's whole range is done, there is no
// workgroup barrier inside the ladder, and a wavefront holds no point state between batches.
//
// The queues are lock-free rings in LDS driven by LDS fetch-and-add (sched_lf.hpp; rounds 2-3 guarded them with a FIFO
// ticket lock: a 1 020-cycle hand-over per batch): a third ring holds the slots whose element has finished, so that
// claims run 64 at a time.  Forward progress and the batch policy: sched_lf.hpp.
#include <hip/hip_runtime.h>

#include "../../include/fecgpu.h"
#include "p256.hpp"
#include "staging.hpp"
#include "kernels.hpp"
#include "sched_lf.hpp"

namespace fecgpu {

// Scheduler statistics for tools/microbench/sched_stats.hip (compiled in only with -DFEC_SCHED_STATS; the shipped
// library has none of it): [0] doubling batches, [1] lanes in them, [2] addition batches, [3] lanes in them,
// [6] sleeps and lost races of waiting wavefronts, [7] claim tasks (batches of free slots taking their next elements)
// (wave-local counters, flushed with one atomic per counter per wavefront when the kernel ends: counting with an atomic
// per event slowed the kernel 2000-fold and measured the atomics)
#ifdef FEC_SCHED_STATS
__device__ unsigned long long g_sched_stats[8];
#define FEC_STAT_DECL unsigned fec_st[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define FEC_STAT(i, v) fec_st[i] += (unsigned)(v)
#define FEC_STAT_FLUSH                                                                             \
  do {                                                                                             \
    if ((threadIdx.x & 63) == 0)                                                                   \
      for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_sched_stats[i_], (unsigned long long)fec_st[i_]); \
  } while (0)
#else
#define FEC_STAT_DECL \
  do {                \
  } while (0)
#define FEC_STAT(i, v) \
  do {                 \
  } while (0)
#define FEC_STAT_FLUSH \
  do {                 \
  } while (0)
#endif

namespace {

FEC_DEV p256::pt ld_pt(const u32* l, int stride) {
  p256::pt p;
  FEC_UNROLL for (int i = 0; i < 8; ++i) {
    p.x.w[i] = l[i * stride];
    p.y.w[i] = l[(8 + i) * stride];
    p.z.w[i] = l[(16 + i) * stride];
  }
  return p;
}
FEC_DEV void st_pt(u32* l, int stride, const p256::pt& p) {
  FEC_UNROLL for (int i = 0; i < 8; ++i) {
    l[i * stride] = p.x.w[i];
    l[(8 + i) * stride] = p.y.w[i];
    l[(16 + i) * stride] = p.z.w[i];
  }
}
// base point of element g straight from the caller's array (24 words, 16-byte loads; L2-resident:
// a workgroup re-reads its 48 KiB of base points about 128 times)
FEC_DEV p256::pt ld_base(const u32* points, size_t g) {
  const uint4* src = reinterpret_cast<const uint4*>(points + g * 24);
  uint4 v[6];
  FEC_UNROLL for (int i = 0; i < 6; ++i) v[i] = src[i];
  p256::pt p;
  FEC_UNROLL for (int i = 0; i < 2; ++i) {
    p.x.w[4 * i] = v[i].x; p.x.w[4 * i + 1] = v[i].y; p.x.w[4 * i + 2] = v[i].z; p.x.w[4 * i + 3] = v[i].w;
    p.y.w[4 * i] = v[2 + i].x; p.y.w[4 * i + 1] = v[2 + i].y; p.y.w[4 * i + 2] = v[2 + i].z; p.y.w[4 * i + 3] = v[2 + i].w;
    p.z.w[4 * i] = v[4 + i].x; p.z.w[4 * i + 1] = v[4 + i].y; p.z.w[4 * i + 2] = v[4 + i].z; p.z.w[4 * i + 3] = v[4 + i].w;
  }
  return p;
}

// Streaming accesses (a result is written once) carry the non-temporal hint so that they do not push the base points
// -- re-read by every addition -- out of the XCD's L2.
typedef u32 v4u_t __attribute__((ext_vector_type(4)));
FEC_DEV void st_stream(u32* g, u32 a, u32 b, u32 c, u32 d) {
  v4u_t v = {a, b, c, d};
  __builtin_nontemporal_store(v, reinterpret_cast<v4u_t*>(g));
}
FEC_DEV void st_out(u32* o, const p256::pt& r) {  // 24 words, 16-byte stores
  FEC_UNROLL for (int w = 0; w < 2; ++w) {
    st_stream(o + 4 * w, r.x.w[4 * w], r.x.w[4 * w + 1], r.x.w[4 * w + 2], r.x.w[4 * w + 3]);
    st_stream(o + 8 + 4 * w, r.y.w[4 * w], r.y.w[4 * w + 1], r.y.w[4 * w + 2], r.y.w[4 * w + 3]);
    st_stream(o + 16 + 4 * w, r.z.w[4 * w], r.z.w[4 * w + 1], r.z.w[4 * w + 2], r.z.w[4 * w + 3]);
  }
}

// one coordinate (8 words) of a slot / of a base point
FEC_DEV fe ld_coord(const u32* l, int stride, int c) {
  fe a;
  FEC_UNROLL for (int i = 0; i < 8; ++i) a.w[i] = l[(8 * c + i) * stride];
  return a;
}
FEC_DEV fe ld_base_coord(const u32* point, int c) {
  const uint4* src = reinterpret_cast<const uint4*>(point + 8 * c);
  const uint4 lo = src[0], hi = src[1];
  fe a;
  a.w[0] = lo.x; a.w[1] = lo.y; a.w[2] = lo.z; a.w[3] = lo.w;
  a.w[4] = hi.x; a.w[5] = hi.y; a.w[6] = hi.z; a.w[7] = hi.w;
  return a;
}

// p256::pdouble (1869-1912) on a slot, coordinates loaded where they are used (Y twice)
FEC_DEV p256::pt pdouble_in_place(const u32* lp, int stride) {
  using namespace p256;
  const fe x = ld_coord(lp, stride, 0);
  const fe xx = sqr(x);
  const fe yy = sqr(ld_coord(lp, stride, 1));
  const fe yyyy = sqr(yy);
  const fe xy2 = sqr(add(x, yy));
  const fe w = sub(sub(xy2, xx), yyyy);
  const fe d = dbl(w);
  const fe e = mul_small(xx, 3);
  const fe ee = sqr(e);
  pt r;
  r.x = sub(sub(ee, d), d);
  r.y = sub(mul(e, sub(d, r.x)), mul_small(yyyy, 8));
  const fe y = ld_coord(lp, stride, 1), z = ld_coord(lp, stride, 2);
  const fe z3 = dbl(y);
  r.z = mul(z3, z);
  // z.is_one() (1909: once per element, on the first doubling after the result became the base point) and the
  // identity early-out (1870) behind one-word tests every lane taking one must pass
  if (__builtin_expect(lanes_where(z.w[0] <= 1u) != 0, 0)) {
    r.z = fe_select(r.z, z3, fe_eq(z, fe_small(1)));
    r = pt_select(r, identity(), fe_is_zero(z));
  }
  return r;
}

// p256::padd (1938-2007) with its operands left in memory: p in the slot (LDS), q in the caller's array.  Every
// coordinate is loaded where it is first used and the early-outs (identity operands, equal or opposite points:
// never on random inputs) re-read both points inside their rare branch, so that neither point is live across the
// sixteen products -- the difference between 43 spilled registers and none at three wavefronts per SIMD.  The
// products, their operands and their order are those of p256::padd_nodouble / padd.
// `lzz` (when not null): z2z2 = q.z * q.z of this addend in LDS (word w at lzz[w * zstride]), computed once per element
// (or per launch for a fixed base) by the same sqr() -- the addend of an element never changes, and its square is one of
// the sixteen products of every addition.
//
// AFFINE (every lane's addend has z == 1 -- the generator, or a key that came through from_affine -- and every lane's
// x1, y1 are below p, which the caller has tested): z2z2 = 1 * 1 is 1, and u1 = x1 * 1, s1 = y1 * 1 * 1 are x1 and y1
// themselves, because Mul by the canonical 1 is reduce_wide of the operand followed by reduce() (544-704, 88-99), i.e.
// the operand's canonical form.  Four of the sixteen products are gone, nothing else changes.
template <bool AFFINE>
FEC_DEV p256::pt padd_body(const u32* lp, int stride, const u32* gq, const u32* lzz, int zstride) {
  using namespace p256;
  const fe z1 = ld_coord(lp, stride, 2), z2 = AFFINE ? fe_small(1) : ld_base_coord(gq, 2);
  const lmask idp = fe_is_zero(z1), idq = AFFINE ? (lmask)0 : fe_is_zero(z2);
  const fe z1z1 = sqr(z1), z2z2 = AFFINE ? fe_small(1) : (lzz ? ld_coord(lzz, zstride, 0) : sqr(z2));
  const fe zs = sub(sub(sqr(add(z1, z2)), z1z1), z2z2);
  const fe s1 = AFFINE ? ld_coord(lp, stride, 1) : mul(mul(ld_coord(lp, stride, 1), z2), z2z2);
  __builtin_amdgcn_sched_barrier(0);  // keep the loads of q's coordinates where they are used (register budget)
  const fe s2 = mul(mul(ld_base_coord(gq, 1), z1), z1z1);
  const fe u1 = AFFINE ? ld_coord(lp, stride, 0) : mul(ld_coord(lp, stride, 0), z2z2);
  __builtin_amdgcn_sched_barrier(0);
  const fe u2 = mul(ld_base_coord(gq, 0), z1z1);
  const lmask ueq = fe_eq(u1, u2);
  lmask same = 0, opposite = 0;
  if (__builtin_expect(ueq != 0, 0)) {
    same = ueq & fe_eq(s1, s2);
    opposite = ueq & fe_eq(s1, neg(s2));
  }
  const fe h = sub(u2, u1);
  const fe z3 = mul(zs, h);
  const fe s21 = sub(s2, s1);
  const fe r = dbl(s21);
  const fe i = sqr(dbl(h));
  const fe j = mul(h, i);
  const fe v = mul(u1, i);
  pt o;
  o.x = sub(sub(sub(sqr(r), j), v), v);
  o.y = sub(mul(r, sub(v, o.x)), mul(dbl(s1), j));
  o.z = z3;
  if (__builtin_expect((idp | idq | ueq) != 0, 0)) {
    pt p, q;
    p.x = ld_coord(lp, stride, 0); p.y = ld_coord(lp, stride, 1); p.z = ld_coord(lp, stride, 2);
    q.x = ld_base_coord(gq, 0); q.y = ld_base_coord(gq, 1); q.z = ld_base_coord(gq, 2);
    o = pt_select(o, identity(), uniform_mask(opposite));
    o = pt_select(o, p, idq);
    o = pt_select(o, q, idp);
    const lmask nd = uniform_mask(same & ~idp & ~idq);
    if (nd != 0) o = pt_select(o, pdouble(p), nd);
  }
  return o;
}
// `q_affine` is wave-uniform: the caller knows that the addend of every lane that counts has z == 1
FEC_DEV p256::pt padd_in_place(const u32* lp, int stride, const u32* gq, const u32* lzz, int zstride, bool q_affine) {
  if (q_affine) {
    // x1, y1 < p?  A value >= p has an all-ones top word (the reference's own Sub emits such values about once per
    // 2^20 scalar-muls): the wavefront then takes the general form, whose products canonicalise them
    const u32 x7 = lp[7 * stride], y7 = lp[15 * stride];
    if (__builtin_expect(lanes_where(x7 == 0xFFFFFFFFu || y7 == 0xFFFFFFFFu) == 0, 1))
      return padd_body<true>(lp, stride, gq, lzz, zstride);
  }
  return padd_body<false>(lp, stride, gq, lzz, zstride);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// Persistent workgroups (the shape of k_ed_mul_pers): one workgroup of TWELVE wavefronts per CU -- three per SIMD,
// 168 VGPRs, the asm field blocks at v[122:167] -- owns a contiguous RANGE of elements and keeps QS = 864 of them
// in LDS slots (point 96 B + scalar 32 B + z2z2 of the base point 32 B), refilling a slot from the range the moment its
// element finishes: no workgroup tail until the whole range is done.  History per 2^20 batch -- round 2: 512 elements
// per workgroup, two workgroups of four wavefronts per CU 28.55 ms; persistent, eight wavefronts 28.2-28.4 ms; twelve
// wavefronts with the operands of Add / double left in memory (padd_in_place, pdouble_in_place: no spills) 26.3 ms;
// rare legs in place, control words in LDS address space 24.9 ms -- round 3: the ladder's prefix answered by claim(),
// z2z2 once per element 24.3 ms; one task per ladder step 23.8 ms; 832 slots with the whole scalar in LDS: the same
// time at 1.05x the algorithmic bytes instead of 2.2x -- round 4: lock-free rings, claims of 64
// elements at full width, 864 slots: 23.2-23.4 ms, 13-21 % less at 2^14 .. 2^17 elements.
// ---------------------------------------------------------------------------------------------------
namespace {
constexpr int QT = 768;      // threads per workgroup: 12 wavefronts, three per SIMD
#ifndef FEC_P256_QS
#define FEC_P256_QS 864
#endif
constexpr int QS_MAIN = FEC_P256_QS;  // element slots per workgroup (12 x 64 in flight + 96 queued).  Same-box sweep with the lock-free rings
                                      // (profiles/sched_r04/ab_libs_r04b.txt): 800 -> 24.16 ms, 832 -> 24.06, 864 -> 23.21, 896 -> 23.25 (round 3's, with the lock:
                                      // profiles/slot_sweep_r03.txt)
// The second instantiation: 1 024 slots, the scalar NOT in LDS (there is no room for it beside 1 024 points and z2z2:
// a step's bit is read from the caller's array instead).  For launches whose workgroups get a little more than a
// whole number of QS_MAIN-element fills -- the late, thinly occupied last fill costs 5-22 % there (2^18 elements: 1 024 per
// workgroup, 8.25 ms against 6.8; 2^19: 13.3 against 12.7) -- p256_launch_mul picks it by the per-workgroup count.
constexpr int QS_WIDE = 1024;
// Fixed base: no z2z2 per slot (one for the workgroup), so the scalar fits beside the point at any slot count
#ifndef FEC_P256_QS_FIXED
#define FEC_P256_QS_FIXED FEC_P256_QS
#endif
constexpr int QS_FIXED = FEC_P256_QS_FIXED;
}  // namespace

// HOIST (fixed base): z2z2 of the base point is computed once per workgroup into LDS instead of by every addition.
template <bool FIXED, bool HOIST, int QS>
__global__ __launch_bounds__(QT, 1) void k_p256_mul_sched(const u32* __restrict__ scalars, const u32* __restrict__ points,
                                                      u32* __restrict__ out, size_t n, unsigned per_wg,
                                                      unsigned* __restrict__ err, unsigned force_fault,
                                                      const u32* __restrict__ prefix, int wbits) {
  constexpr int RING = QS < 1024 ? 1024 : 2048;   // ring positions: more than there are slots (sched_lf.hpp)
  static_assert(QS <= 1024, "a ring entry holds a ten-bit slot number");
  __shared__ u32 lds_st[24 * QS];               // X, Y, Z of slot e: word w at lds_st[w * QS + e]
  // z2z2 = base.z * base.z of slot e's element (variable base): one of the sixteen products of EVERY addition of an
  // element depends on its base point alone, so claim() computes it once with the same sqr() and the ~128 additions
  // read it back.  With 864 slots it fits beside the whole scalar (160 B per slot, 146 KiB in all).  (Round 2 parked
  // z2z2 in the element's output slot instead: 23 GB of L2-side traffic and no in-place calls -- removed; this form has
  // neither.)
  __shared__ u32 lds_zq[(FIXED ? 0 : 8 * QS) + 8];
  constexpr bool KLDS = FIXED || QS <= QS_MAIN;          // the whole scalar of every slot in LDS (not with QS_WIDE slots: no room)
  __shared__ u32 lds_k[KLDS ? 8 * QS : 8];      // scalar of slot e: word w at lds_k[w * QS + e]
  __shared__ u32 lds_gid[QS];                   // element of slot e, relative to the workgroup's range
  __shared__ unsigned short lds_step[QS];
  __shared__ __attribute__((aligned(16))) int lds_lf[LF_INTS<RING>];   // control words + the rings D, A, F (sched_lf.hpp)
  __shared__ __attribute__((aligned(16))) u32 lds_zz[8];   // FIXED && HOIST: base.z * base.z (one for the workgroup)
  const size_t lo = (size_t)blockIdx.x * per_wg;
  const int range = (n - lo) < (size_t)per_wg ? (int)(n - lo) : (int)per_wg;
  const int tid = threadIdx.x, lane = tid & 63;
  // control words through an LDS-address-space pointer in one opaque base register, immediate offsets
  lds_int_ptr ctl = (lds_int_ptr)lds_lf;
  asm volatile("" : "+v"(ctl));
  const unsigned ctl_addr = (unsigned)(size_t)ctl;  // LDS byte address of the control block
  if (FIXED && HOIST) {  // every lane computes the same square; one stores it
    const fe zz = p256::sqr(ld_base_coord(points, 2));
    if (tid == 0) {
      FEC_UNROLL for (int w = 0; w < 8; ++w) lds_zz[w] = zz.w[w];
    }
  }
  // every slot starts in the free ring: the wavefronts' first pops are claims of 64 elements each
  lf_init<RING>(lds_lf, tid, QT, range < QS ? range : QS, force_fault ? (unsigned)FEC_DEVERR_FORCED : 0u);
  __syncthreads();

  // Claims the next element of the range for slot `e`.  multiply's early-outs (2121-2124: identity point or zero
  // scalar) are answered at once.  The ladder's prefix is answered at once as well: until the scalar's top set bit
  // (position t) every doubling is double(identity) = identity (1870) and the first addition is identity + point =
  // point (Add's first early-out returns rhs as it is), so the slot starts with result = point at step 256 - t --
  // exactly the state the reference is in after that addition.  (Besides the skipped operations this keeps identity
  // results out of the queues: with them, 39 % of all batches held at least one lane on an early-out and ran Add's /
  // double's rare leg for the whole wavefront.)  Returns the scalar bit of the element's next step (= the ring it
  // goes to) or LF_NXT_DEAD when the range is used up (the slot dies).
  auto claim = [&](int e) -> int {
    for (;;) {
      const int rel = lds_fetch_add(ctl, LF_NEXT, 1);
      if (rel >= range) return LF_NXT_DEAD;
      const size_t g = lo + rel;
      const uint4* ks = reinterpret_cast<const uint4*>(scalars + g * 8);
      const uint4 k0 = ks[0], k1 = ks[1];
      const u32 any = k0.x | k0.y | k0.z | k0.w | k1.x | k1.y | k1.z | k1.w;
      u32 zany = 0;
      if (FIXED) {
        FEC_UNROLL for (int w = 0; w < 8; ++w) zany |= points[16 + w];
      } else {
        const uint4* z = reinterpret_cast<const uint4*>(points + g * 24 + 16);
        const uint4 z0 = z[0], z1 = z[1];
        zany = z0.x | z0.y | z0.z | z0.w | z1.x | z1.y | z1.z | z1.w;
      }
      if (any == 0 || zany == 0) {
        uint4* o = reinterpret_cast<uint4*>(out + g * 24);
        FEC_UNROLL for (int w = 0; w < 6; ++w) o[w] = make_uint4(w == 2 ? 1u : 0u, 0, 0, 0);  // identity (0, 1, 0)
        continue;
      }
      const u32 kw[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
      int t = 0;  // top set bit of the scalar
      FEC_UNROLL for (int w = 0; w < 8; ++w) t = kw[w] ? 32 * w + 31 - __builtin_clz(kw[w]) : t;
      const p256::pt base = ld_base(points, FIXED ? 0 : g);
      if (t == 0) {  // scalar == 1: 255 doublings of the identity, then identity + point
        st_out(out + g * 24, base);
        continue;
      }
      // Fixed base with a prefix table (fecgpu.hip: ensure_gen_prefix): the result after the first wbits steps --
      // bits 255 .. 256 - wbits -- depends on those bits alone; entry `idx` of the table is that result (24 words),
      // computed once per ctx by this very kernel (multiply(base, idx) performs the same doublings and additions after
      // its leading doublings of the identity).  Pattern 0 is the identity: the top-bit shortcut below covers it.
      const u32 idx = FIXED && wbits > 0 ? kw[7] >> (32 - wbits) : 0u;
      p256::pt start = base;
      if (idx != 0) {
        t = 256 - wbits;   // the slot starts at step wbits
        start = ld_base(prefix, idx);
      }
      st_pt(lds_st + e, QS, start);
      if (!FIXED) {
        const fe zz = p256::sqr(base.z);
        FEC_UNROLL for (int w = 0; w < 8; ++w) lds_zq[w * QS + e] = zz.w[w];
      }
      u32 cur = 0;  // the word that holds bit t - 1, the next one the ladder looks at
      FEC_UNROLL for (int w = 0; w < 8; ++w) {
        if (KLDS) lds_k[w * QS + e] = kw[w];
        cur = ((t - 1) >> 5) == w ? kw[w] : cur;
      }
      lds_gid[e] = (u32)rel;
      // bit 15 of the step word: this element's base point has z == 1 (Add then needs four products fewer)
      const bool z_one = lane_of(fe_eq(base.z, fe_small(1)));
      lds_step[e] = (unsigned short)((256 - t) | (z_one ? 0x8000 : 0));
      return (int)(cur >> ((t - 1) & 31)) & 1;   // the bit of step 256 - t
    }
  };
  // the scalar bit of step `step_new` (< 256): bit 255 - step_new (2127-2129)
  auto step_bit = [&](int e, int step_new) -> u32 {
    const int b = 255 - step_new;
    if (KLDS) return (lds_k[(b >> 5) * QS + e] >> (b & 31)) & 1u;
    return (scalars[(lo + lds_gid[e]) * 8 + (b >> 5)] >> (b & 31)) & 1u;   // (lds_gid[e] < range: written by claim())
  };

  int e = 0;
  int nxt = LF_NXT_NONE;
  unsigned watchdog = 0;
  FEC_STAT_DECL;
  for (;;) {
    // hand on what the last batch left (sched_lf.hpp: no lock, no turn to wait for), take the next one
    lf_push<RING>(ctl_addr, lane, nxt, e, (u32)FEC_DEVERR_SCHED_WATCHDOG);
    nxt = LF_NXT_NONE;
    const LfPop pop = lf_pop<RING>(ctl_addr, lane, watchdog, (u32)FEC_DEVERR_SCHED_WATCHDOG);
    if (pop.kind < 0) break;
    if (!lf_entry<RING>(ctl_addr, pop, lane, (u32)FEC_DEVERR_SCHED_WATCHDOG, e)) break;
    const bool active = lane < pop.count;
    if (pop.kind == LF_Q_F) {   // free slots: each takes the next element of the range (64 claims at full width)
      FEC_STAT(7, 1);
      if (active) nxt = claim(e);
      continue;
    }
    const int kind = pop.kind;
    const int step_word = active ? lds_step[e] : 0x8000;   // (a lane without an element does not veto the z == 1 form)
    int step = step_word & 0x7FFF;
    // Every global address below is formed from this index (written by claim(), always < range): a broken queue must
    // never address memory outside the workgroup's own range -- such a lane works on element 0, stores nothing, raises the error.
    u32 gid = active ? lds_gid[e] : 0u;
    const bool oob = gid >= (u32)range;
    gid = oob ? 0u : gid;
    const bool live = active && !oob;
    bool fin = false;
    p256::pt res = p256::identity();
    FEC_STAT(kind == 0 ? 0 : 2, 1);
    FEC_STAT(kind == 0 ? 1 : 3, pop.count);
    // ONE task = one step of the reference's loop (2126-2134) for 64 elements that agree on the step's scalar bit:
    //   ring D (bit clear): result = result.double()
    //   ring A (bit set):   result = result.double(); result = result + point
    // (Round 2 queued the doubling and the addition separately: 381 visits of the scheduler per element instead of 254.)
    FEC_MARK("task_double_begin");
#ifdef FEC_SCHED_STUB   // tools/microbench/sched_stats.hip: the scheduler alone (the task is a copy of the slot)
    res = ld_pt(lds_st + e, QS);
#else
    res = pdouble_in_place(lds_st + e, QS);
#endif
    FEC_MARK("task_double_end");
    if (kind == LF_Q_A) {
      // the addition reads its first operand from the slot (a lane's LDS accesses stay in order); inactive lanes add
      // slot 0 and element 0 of the range: harmless, never stored
      if (live) st_pt(lds_st + e, QS, res);
      const size_t g_el = lo + gid;
      FEC_MARK("task_add_begin");
#ifdef FEC_SCHED_STUB
      res = ld_pt(lds_st + e, QS);
      (void)g_el;
#else
      const bool all_affine = __builtin_amdgcn_ballot_w64(live && !(step_word & 0x8000)) == 0;
      res = padd_in_place(lds_st + e, QS, FIXED ? points : points + g_el * 24,
                          FIXED ? (HOIST ? lds_zz : nullptr) : lds_zq + e, FIXED ? 1 : QS, all_affine);
#endif
      FEC_MARK("task_add_end");
    }
    if (live) {
      ++step;
      lds_step[e] = (unsigned short)(step | (step_word & 0x8000));
      fin = step == 256;
      if (!fin) {
        st_pt(lds_st + e, QS, res);
        nxt = (int)step_bit(e, step);
      }
    }
    if (fin) {  // the element is done: its result goes out (16-byte stores), the slot joins the free ring
      st_out(out + (lo + gid) * 24, res);
      nxt = LF_NXT_FREE;
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(oob) != 0, 0)) {
      lf_raise(ctl_addr, lane, (u32)FEC_DEVERR_SCHED_INDEX);
      nxt = LF_NXT_NONE;
    }
  }
  FEC_STAT(6, watchdog);
  FEC_STAT_FLUSH;
  __syncthreads();
  if (const int ec = lds_lf[LF_ERR]) {
    // Scheduler fault (watchdog, index guard, or the debug hook): zero-fill the workgroup's results AND set the ctx's
    // error word, which the host reads after its synchronisation -- the call returns FEC_E_LAUNCH (see kernels_ed.hip).
    for (int el = tid; el < range; el += QT) {
      uint4* o = reinterpret_cast<uint4*>(out + (lo + el) * 24);
      FEC_UNROLL for (int w = 0; w < 6; ++w) o[w] = make_uint4(0, 0, 0, 0);
    }
    if (tid == 0 && err != nullptr) {
      *reinterpret_cast<volatile unsigned*>(err) = (unsigned)ec;
      __threadfence_system();
    }
  }
}

namespace {
// Which instantiation a launch whose workgroups own `per_wg` elements each takes: QS_WIDE when those elements are a
// little more than a whole number of QS_MAIN-element fills and a (near) whole number of QS_WIDE-element ones -- a fill
// of leftovers starts late and runs thinly occupied -- and only up to three fills (beyond that the refills overlap and
// the leaner LDS image with the scalar in it wins: 23.9 against 24.3 ms at 2^20).
inline bool wide_slots_pay(unsigned per_wg) {
  if (per_wg <= (unsigned)QS_MAIN || per_wg > 3u * QS_WIDE) return false;
  auto waste = [per_wg](unsigned q) { return (double)(((per_wg + q - 1) / q) * q) / (double)per_wg; };
  return waste(QS_WIDE) + 0.04 < waste(QS_MAIN);
}
}  // namespace

void p256_launch_mul(const SchedEnv& env, bool fixed, const u32* scalars, const u32* points, u32* out, size_t n, hipStream_t s) {
  // one workgroup per CU the launch may take (env.cus), each with a contiguous range of at least 64 elements
  const unsigned cus = env.cus ? env.cus : 256u;
  size_t grid = (n + 63) / 64;
  if (grid > cus) grid = cus;
  const unsigned per_wg = (unsigned)((n + grid - 1) / grid);
  grid = (n + per_wg - 1) / per_wg;
  const bool wide = wide_slots_pay(per_wg);
  const dim3 g((unsigned)grid), b(QT);
  if (fixed) {
    const bool tab = points == env.gen[FEC_P256] && env.gen_prefix[FEC_P256] != nullptr && env.gen_prefix_bits[FEC_P256] > 0;
    const u32* pre = tab ? env.gen_prefix[FEC_P256] : (const u32*)nullptr;
    const int wbits = tab ? (int)env.gen_prefix_bits[FEC_P256] : 0;
    hipLaunchKernelGGL((k_p256_mul_sched<true, true, QS_FIXED>), g, b, 0, s, scalars, points, out, n, per_wg, env.err,
                       env.force_fault, pre, wbits);
    return;
  }
  // Variable base: z2z2 is recomputed by every addition.  Parking it in the element's (still unused) output slot was
  // 2.6 % faster (25.0 -> 24.4 ms) but pushed a workgroup's working set out of its XCD's L2 -- 23 GB of L2-side
  // fetches per launch instead of 0.44 (profiles/pmc_r02br_p256_hoist.json) -- and broke in-place calls; not kept.
  if (wide)
    hipLaunchKernelGGL((k_p256_mul_sched<false, false, QS_WIDE>), g, b, 0, s, scalars, points, out, n, per_wg, env.err,
                       env.force_fault, (const u32*)nullptr, 0);
  else
    hipLaunchKernelGGL((k_p256_mul_sched<false, false, QS_MAIN>), g, b, 0, s, scalars, points, out, n, per_wg, env.err,
                       env.force_fault, (const u32*)nullptr, 0);
}

}  // namespace fecgpu
