// kernels_secp.hip -- the headline kernel: secp256k1 Curve::multiply (secp256k1.rs:2635-2692), one
// scalar multiplication per lane, at THREE wavefronts per SIMD.
//
// The ladder step is s = r0 + r1; d = double(bit ? r1 : r0); (r0, r1) = bit ? (s, d) : (d, s).
// Held entirely in registers it needs 214-256 VGPRs (2 waves per SIMD).  The measured issue cost of
// the v_mad_u64_u32 / v_addc pairs that make up most of the kernel drops from 5.3 to 5.0 cycles per
// instruction with a third wavefront on the SIMD (profiles/valu_latency_r02.txt), so this kernel is
// built for 168 VGPRs: both ladder points LIVE in LDS (48 words per lane) -- a step loads them for the
// addition, re-reads the doubling's operand, and stores sum and doubling to the slots the scalar bit
// selects (an address select instead of 72 register selects per step) -- the field multiplications'
// fixed register block sits at v[132:167] (tools/gen_field_asm.py), and the scalar is read from HBM
// one word per 32 steps instead of being staged in LDS.  LDS: 48 KiB per workgroup, three workgroups
// per CU.
#include <hip/hip_runtime.h>

#include "../../include/fecgpu.h"
#include "secp256k1.hpp"
#include "staging.hpp"
#include "kernels.hpp"

namespace fecgpu {

namespace {

FEC_DEV secp::pt ld3(const u32* l, int stride) {
  secp::pt p;
  FEC_UNROLL for (int i = 0; i < 8; ++i) {
    p.x.w[i] = l[i * stride];
    p.y.w[i] = l[(8 + i) * stride];
    p.z.w[i] = l[(16 + i) * stride];
  }
  return p;
}
FEC_DEV void st3(u32* l, int stride, const secp::pt& p) {
  FEC_UNROLL for (int i = 0; i < 8; ++i) {
    l[i * stride] = p.x.w[i];
    l[(8 + i) * stride] = p.y.w[i];
    l[(16 + i) * stride] = p.z.w[i];
  }
}

// one coordinate (c = 0, 1, 2: X, Y, Z) of a ladder point in its LDS slot
FEC_DEV fe ldc(const u32* l, int c) {
  fe a;
  FEC_UNROLL for (int i = 0; i < 8; ++i) a.w[i] = l[(8 * c + i) * TPB];
  return a;
}

// secp::padd_nodouble (Add, 1444-1498) with both operands left in their LDS slots: every coordinate is loaded where
// it is first used (Z twice), so that neither point is live across the sixteen products, and the early-outs
// (identity operands: Z == 0; u1 == u2) sit behind one-word tests every lane taking one must pass, their exact masks
// evaluated on the re-read points inside the rare branch.  The same products on the same operands as
// padd_nodouble(); independent products are merely issued in another order.
FEC_DEV secp::pt padd_slots(const u32* lp, const u32* lq, lmask& need_double) {
  using namespace secp;
  fe z1s, z1c, z2s, z2c;
  lmask maybe;
  {
    const fe z1 = ldc(lp, 2);
    maybe = lanes_where(z1.w[0] == 0u);
    z1s = sqr(z1);
    z1c = mul(z1s, z1);
  }
  __builtin_amdgcn_sched_barrier(0);  // keep the loads where they are used (register budget of three waves per SIMD)
  {
    const fe z2 = ldc(lq, 2);
    maybe |= lanes_where(z2.w[0] == 0u);
    z2s = sqr(z2);
    z2c = mul(z2s, z2);
  }
  __builtin_amdgcn_sched_barrier(0);
  const fe u1 = mul(ldc(lp, 0), z2s);
  const fe u2 = mul(ldc(lq, 0), z1s);
  __builtin_amdgcn_sched_barrier(0);
  const fe s1 = mul(ldc(lp, 1), z2c);
  const fe s2 = mul(ldc(lq, 1), z1c);
  __builtin_amdgcn_sched_barrier(0);
  lmask ueq = 0, seq = 0;
  if (__builtin_expect(lanes_where(u1.w[0] == u2.w[0]) != 0, 0)) {
    ueq = fe_eq(u1, u2);
    seq = fe_eq(s1, s2);
  }
  const fe h = sub(u2, u1);
  const fe r = sub(s2, s1);
  const fe h2 = sqr(h);
  const fe h3 = mul(h2, h);
  const fe u1h2 = mul(u1, h2);
  pt o;
  o.x = sub(sub(sub(sqr(r), h3), u1h2), u1h2);
  o.y = sub(mul(r, sub(u1h2, o.x)), mul(s1, h3));
  o.z = mul(mul(h, ldc(lp, 2)), ldc(lq, 2));
  need_double = 0;
  if (__builtin_expect((maybe | ueq) != 0, 0)) {  // early-outs: only the ladder's first steps
    const pt p = ld3(lp, TPB), q = ld3(lq, TPB);
    const lmask idp = is_identity(p), idq = is_identity(q);
    o = pt_select(o, identity(), ueq & ~seq);
    o = pt_select(o, p, idq);
    o = pt_select(o, q, idp);
    need_double = ueq & seq & ~idp & ~idq;
  }
  return o;
}

}  // namespace

// MODE 0: variable base (points[i]); 1: one base (points[0]); 2: one base whose ladder starts from the PREFIX TABLE;
// 3: builds one level of that table from the level below.
//
// The prefix table of a fixed base (fecgpu.hip: ensure_gen_prefix).  The ladder's state after its first w steps --
// the pair (r0, r1) -- depends on the base and on the first w scalar bits alone, so for the reference's generator() it
// is computed once per ctx for all 2^w bit patterns and kept in HBM: 2^w entries of 48 words, r0 then r1.  It grows
// level by level (MODE 3): entry g of level j is ONE ladder step -- this very loop body, the same addition and
// doubling on the same operands as a full multiplication runs -- from entry g >> 1 of level j - 1 with the bit g & 1;
// level 0 is the ladder's initial state (identity, base).  2^(w+1) steps in all, 3.6 ms for w = 24.  A multiplication
// by the generator (MODE 2) fetches its entry and runs the remaining 256 - w steps: w / 256 of the work is not redone
// for every element.  (The table is indexed by the bits in ladder order: step i reads bit 7 - i % 8 of byte i / 8 of
// the scalar's little-endian bytes, 2655-2659, so the first 32 steps are bswap32 of word 0, msb first.)
template <int MODE>
__global__ __launch_bounds__(TPB, 3) void k_secp_mul(const u32* __restrict__ scalars,
                                                  const u32* __restrict__ points,
                                                  u32* __restrict__ out, size_t n,
                                                  const u32* __restrict__ prefix, int wbits) {
  __shared__ u32 lds[48 * TPB];  // word w of lane e: r0 at lds[w * TPB + e], r1 at lds[(24 + w) * TPB + e]
  const int valid = block_valid(n);
  const size_t first = (size_t)blockIdx.x * TPB;
  const int e = threadIdx.x;
  if (MODE == 0) stage_in<24>(lds, points + first * 24, valid);
  __syncthreads();
  if (e < valid) {
    const size_t g = first + e;
    const u32* kg = scalars + g * 8;
    lmask early = 0;
    // The two ladder points LIVE in LDS: slot 0 (words 0..23) holds r0, slot 1 (words 24..47) r1.  A step reads both,
    // adds them, re-reads the one the reference's kept doubling takes, and writes the sum and the doubling back to
    // the slots the bit selects -- one address select per access instead of a register select per word.
    u32* const slot0 = lds + e;
    int i0 = 0, i1 = 256;
    if (MODE != 3) {
      u32 any = 0;
      FEC_UNROLL for (int i = 0; i < 8; ++i) any |= kg[i];
      early = lanes_where(any == 0);
    }
    if (MODE == 2) early |= secp::is_identity(ld3(points, 1));   // (2636-2639: a caller's own base may be the identity)
    if (MODE == 2 || MODE == 3) {
      // MODE 3: `prefix` is the level below, this element's parent entry is g >> 1
      const u32 idx = MODE == 3 ? (u32)(g >> 1) : __builtin_bswap32(kg[0]) >> (32 - wbits);
      const uint4* row = reinterpret_cast<const uint4*>(prefix + (size_t)idx * 48);
      FEC_UNROLL for (int q = 0; q < 12; ++q) {
        const uint4 x = row[q];
        slot0[(4 * q + 0) * TPB] = x.x;
        slot0[(4 * q + 1) * TPB] = x.y;
        slot0[(4 * q + 2) * TPB] = x.z;
        slot0[(4 * q + 3) * TPB] = x.w;
      }
      i0 = MODE == 3 ? 0 : wbits;
      if (MODE == 3) i1 = 1;
    } else {
      const secp::pt r1 = MODE != 0 ? ld3(points, 1) : ld3(lds + e, TPB);
      early |= secp::is_identity(r1);
      st3(slot0, TPB, secp::identity());
      st3(slot0 + 24 * TPB, TPB, r1);
    }
    u32 kword = MODE == 3 ? 0u : kg[i0 >> 5];
#pragma unroll 1
    for (int i = i0; i < i1; ++i) {
      u32 b;
      if (MODE == 3) {
        b = (u32)g & 1u;   // entry g of this level: the parent's bits, then this one
      } else {
        if ((i & 31) == 0) kword = kg[i >> 5];
        // bit i of the ladder (2655-2659): byte i/8 of the little-endian bytes, MSB first in the byte
        const int sh = (((i >> 3) & 3) << 3) + 7 - (i & 7);
        b = (kword >> sh) & 1u;
      }
      lmask nd;
      secp::pt s = padd_slots(slot0, slot0 + 24 * TPB, nd);
      if (__builtin_expect(nd != 0, 0)) {  // Add (1469-1473) returns self.double(): never on random inputs
        secp::pt d0 = secp::pdouble(ld3(slot0, TPB));
        s = secp::pt_select(s, d0, nd);
      }
      // (r0, r1) = bit ? (s, d) : (d, s) with d = double(bit ? r1 : r0), the only doubling the reference keeps
      // (2669-2684): the doubling's operand sits in slot `b`, which the doubling then overwrites; the sum goes to
      // the other slot, whose point is dead once the addition has read it
      u32* const slot_d = slot0 + b * (24u * TPB);
      u32* const slot_s = slot0 + (b ^ 1u) * (24u * TPB);
      const secp::pt din = ld3(slot_d, TPB);
      st3(slot_s, TPB, s);
      st3(slot_d, TPB, secp::pdouble(din));
    }
    if (MODE != 3) {
      const secp::pt r0 = secp::pt_select(ld3(slot0, TPB), secp::identity(), early);
      st3(slot0, TPB, r0);
    }
  }
  __syncthreads();
  if (MODE == 3) stage_out<48>(out + first * 48, lds, valid);
  else stage_out<24>(out + first * 24, lds, valid);
}

void secp_launch_mul(const SchedEnv& env, bool fixed, const u32* scalars, const u32* points, u32* out, size_t n, hipStream_t s) {
  const unsigned grid = (unsigned)((n + TPB - 1) / TPB);
  const int w = (int)env.gen_prefix_bits[FEC_SECP256K1];
  if (fixed && w > 0 && points == env.gen[FEC_SECP256K1] && env.gen_prefix[FEC_SECP256K1] != nullptr)
    hipLaunchKernelGGL((k_secp_mul<2>), dim3(grid), dim3(TPB), 0, s, scalars, points, out, n, env.gen_prefix[FEC_SECP256K1], w);
  else if (fixed) hipLaunchKernelGGL((k_secp_mul<1>), dim3(grid), dim3(TPB), 0, s, scalars, points, out, n, (const u32*)nullptr, 0);
  else hipLaunchKernelGGL((k_secp_mul<0>), dim3(grid), dim3(TPB), 0, s, scalars, points, out, n, (const u32*)nullptr, 0);
}

void secp_prefix_level_launch(const u32* parent, u32* child, size_t child_entries, hipStream_t s) {
  hipLaunchKernelGGL((k_secp_mul<3>), dim3((unsigned)((child_entries + TPB - 1) / TPB)), dim3(TPB), 0, s, (const u32*)nullptr,
                     (const u32*)nullptr, child, child_entries, parent, 0);
}

}  // namespace fecgpu
