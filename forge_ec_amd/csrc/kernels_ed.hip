// kernels_ed.hip -- Ed25519 variable-base Curve::multiply (ed25519.rs:2062-2097) as a persistent
// workgroup-level task scheduler (the queue design of kernels_p256.hip).
//
//   result = identity; addend = point
//   for i in 0..256 { s = result + addend; result = bit(i) ? s : result; addend = addend.double() }
//
// The reference computes `result + addend` every step and keeps it only where the bit is set: the
// additions of clear bits are dead work, and so is the last doubling.  Round 1 executed them all
// (lock-step lanes: 2 x 9 field multiplications per step).  Here every element runs exactly the
// operations whose results are used, in the reference's order -- A_i (only if bit i is set), then D_i
// (for i < 255) -- and a workgroup's wavefronts pull BATCHES of 64 elements that all need an
// addition or all need a doubling from two ready queues in LDS.  The doubling is the reference's
// double() = self + self (1828-1832), whose four self-products are formed with the exact squaring.
#include <hip/hip_runtime.h>

#include "../../include/fecgpu.h"
#include "ed25519.hpp"
#include "staging.hpp"
#include "kernels.hpp"
#include "sched_lf.hpp"

namespace fecgpu {

namespace {


FEC_DEV ed::pt ld_lds(const u32* l, int stride) {
  ed::pt p;
  FEC_UNROLL for (int i = 0; i < 8; ++i) {
    p.x.w[i] = l[i * stride];
    p.y.w[i] = l[(8 + i) * stride];
    p.z.w[i] = l[(16 + i) * stride];
    p.t.w[i] = l[(24 + i) * stride];
  }
  return p;
}
FEC_DEV void st_lds(u32* l, int stride, const ed::pt& p) {
  FEC_UNROLL for (int i = 0; i < 8; ++i) {
    l[i * stride] = p.x.w[i];
    l[(8 + i) * stride] = p.y.w[i];
    l[(16 + i) * stride] = p.z.w[i];
    l[(24 + i) * stride] = p.t.w[i];
  }
}
FEC_DEV ed::pt ld_glb(const u32* g) {  // 32 consecutive words, 16-byte loads
  const uint4* src = reinterpret_cast<const uint4*>(g);
  uint4 v[8];
  FEC_UNROLL for (int i = 0; i < 8; ++i) v[i] = src[i];
  ed::pt p;
  FEC_UNROLL for (int i = 0; i < 2; ++i) {
    p.x.w[4 * i] = v[i].x; p.x.w[4 * i + 1] = v[i].y; p.x.w[4 * i + 2] = v[i].z; p.x.w[4 * i + 3] = v[i].w;
    p.y.w[4 * i] = v[2 + i].x; p.y.w[4 * i + 1] = v[2 + i].y; p.y.w[4 * i + 2] = v[2 + i].z; p.y.w[4 * i + 3] = v[2 + i].w;
    p.z.w[4 * i] = v[4 + i].x; p.z.w[4 * i + 1] = v[4 + i].y; p.z.w[4 * i + 2] = v[4 + i].z; p.z.w[4 * i + 3] = v[4 + i].w;
    p.t.w[4 * i] = v[6 + i].x; p.t.w[4 * i + 1] = v[6 + i].y; p.t.w[4 * i + 2] = v[6 + i].z; p.t.w[4 * i + 3] = v[6 + i].w;
  }
  return p;
}
FEC_DEV void st_glb(u32* g, const ed::pt& p) {
  uint4* dst = reinterpret_cast<uint4*>(g);
  FEC_UNROLL for (int i = 0; i < 2; ++i) {
    dst[i] = make_uint4(p.x.w[4 * i], p.x.w[4 * i + 1], p.x.w[4 * i + 2], p.x.w[4 * i + 3]);
    dst[2 + i] = make_uint4(p.y.w[4 * i], p.y.w[4 * i + 1], p.y.w[4 * i + 2], p.y.w[4 * i + 3]);
    dst[4 + i] = make_uint4(p.z.w[4 * i], p.z.w[4 * i + 1], p.z.w[4 * i + 2], p.z.w[4 * i + 3]);
    dst[6 + i] = make_uint4(p.t.w[4 * i], p.t.w[4 * i + 1], p.t.w[4 * i + 2], p.t.w[4 * i + 3]);
  }
}

// bit i of scalar.to_raw() of element g (2075-2079: limb i / 64, bit i % 64)
FEC_DEV u32 scalar_bit(const u32* scalars, size_t g, int i) { return (scalars[g * 8 + (i >> 5)] >> (i & 31)) & 1u; }

}  // namespace

namespace {
// a point as 32 consecutive words (x, y, z, t), word loads (any alignment the callers use)
FEC_DEV ed::pt ld_words(const u32* g) {
  ed::pt p;
  FEC_UNROLL for (int i = 0; i < 8; ++i) { p.x.w[i] = g[i]; p.y.w[i] = g[8 + i]; p.z.w[i] = g[16 + i]; p.t.w[i] = g[24 + i]; }
  return p;
}
FEC_DEV void st_words(u32* g, const ed::pt& p) {
  FEC_UNROLL for (int i = 0; i < 8; ++i) { g[i] = p.x.w[i]; g[8 + i] = p.y.w[i]; g[16 + i] = p.z.w[i]; g[24 + i] = p.t.w[i]; }
}
}  // namespace

// table[j] = 2^j * base by the reference's own doubling chain (ed25519.rs:2089): one lane, 255
// sequential additions; 32 words per entry, dense.  Runs once per base point.
__global__ __launch_bounds__(64) void k_ed_build_table(const u32* __restrict__ base, u32* __restrict__ table) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  ed::pt a = ld_words(base);
#pragma unroll 1
  for (int j = 0; j < 256; ++j) {
    st_words(table + j * 32, a);
    a = ed::padd(a, a);
  }
}

// ed::padd (1864-1928) against an entry of the fixed-base LDS table.  What Add computes from the addend alone --
// rhs.y - rhs.x and rhs.y + rhs.x (1897-1898) -- is a function of the table entry, so it is computed once per entry
// when the table is staged (same Sub / Add, same operands): an LDS entry holds y - x, y + x, z, t and word 0 of x and
// of y (FT_STRIDE words).  The coordinates are loaded where they are used, so that the addend is not live across the
// nine products (three wavefronts per SIMD, 168 VGPRs).  The early-outs (1869-1880) sit behind one-word tests every
// lane taking one must pass -- is_identity needs x == 0, the negation test needs p.y == q.y -- and evaluate the exact
// masks on the plain entry (x, y, z, t: `g`, the table in global memory) inside the rare branch.
constexpr int FT_STRIDE = 35;   // odd: lanes reading different entries spread over the LDS banks
FEC_DEV fe ld_tab(const u32* e, int c) {
  fe a;
  FEC_UNROLL for (int i = 0; i < 8; ++i) a.w[i] = e[8 * c + i];
  return a;
}
FEC_DEV ed::pt padd_table(const ed::pt& p, const u32* e, const u32* g) {
  using namespace ed;
  const lmask maybe = lanes_where(p.x.w[0] == 0u || e[32] == 0u || p.y.w[0] == e[33]);
  fe a, b, d;
  d = mul(p.z, ld_tab(e, 2));
  a = mul(sub(p.y, p.x), ld_tab(e, 0));
  b = mul(add(p.y, p.x), ld_tab(e, 1));
  __builtin_amdgcn_sched_barrier(0);  // keep the load of q.t below the first three products (register budget)
  fe c = mul(mul(p.t, ld_tab(e, 3)), D_());
  __builtin_amdgcn_sched_barrier(0);
  const fe ee = sub(b, a), f = sub(d, c), gg = add(d, c), h = add(b, a);
  pt o;
  o.x = mul(ee, f);
  o.y = mul(gg, h);
  o.t = mul(ee, h);
  o.z = mul(f, gg);
  if (__builtin_expect(maybe != 0, 0)) {  // improbable: a lane's first addition is a copy (multiply_fixed_in_place)
    const pt q = ld_words(g);
    const lmask opposite = fe_eq(p.x, neg(q.x)) & fe_eq(p.y, q.y);  // 1878, raw coordinates
    const lmask idp = is_identity(p), idq = is_identity(q);
    o = pt_select(o, identity(), uniform_mask(opposite));
    o = pt_select(o, p, uniform_mask(idq));
    o = pt_select(o, q, uniform_mask(idp));
  }
  return o;
}

// ed::multiply_fixed with the addend read in place (see padd_table).
// `prefix` / `wbits` (fecgpu.hip: ensure_gen_prefix; null / 0 = none): the result after the scalar's first wbits bits --
// bits 0 .. wbits - 1, the ones multiply consumes first (2073-2094) -- depends on those bits alone; entry `low` of the
// prefix table is that result (32 words), computed once per ctx by this very kernel as multiply(base, low).  A lane
// whose low bits are not all zero starts from its entry and goes on with bit wbits.
FEC_DEV ed::pt multiply_fixed_in_place(const ed::pt& base, const u32* tab, const u32* gtab, const u32* kw, const u32* prefix,
                                       int wbits) {
  using namespace ed;
  u32 any = 0;
  FEC_UNROLL for (int i = 0; i < 8; ++i) any |= kw[i * KSTRIDE];
  const lmask early = is_identity(base) | lanes_where(any == 0);
  pt result = identity();
  int wi = 0;
  u32 cur = kw[0];
  const u32 low = wbits > 0 ? cur & ((1u << wbits) - 1u) : 0u;
  cur ^= low;
  {
    // The lane's FIRST addition is identity() + addend, which Add answers with the addend through its first
    // early-out (1869-1871): a copy of the table entry instead of nine products.
    while (cur == 0 && wi < 7) {
      ++wi;
      cur = kw[wi * KSTRIDE];
    }
    const bool have = cur != 0, pre = low != 0;
    const u32* src = pre ? prefix + (size_t)low * 32u : gtab + (have ? (u32)wi * 32u + (u32)__builtin_ctz(cur) : 0u) * 32u;
    const pt q = ld_words(src);
    if (!pre) cur &= cur - 1;  // (0 stays 0)
    result = pt_select(result, q, lanes_where(have || pre));
  }
#pragma unroll 1
  for (;;) {
    while (cur == 0 && wi < 7) {  // advance to this lane's next non-zero scalar word
      ++wi;
      cur = kw[wi * KSTRIDE];
    }
    const bool have = cur != 0;
    const lmask active = lanes_where(have);
    if (active == 0) break;  // every lane of the wavefront has consumed its set bits
    const u32 bpos = have ? (u32)__builtin_ctz(cur) : 0u;
    cur &= cur - 1;
    const u32 j = have ? (u32)wi * 32u + bpos : 0u;
    const pt sum = padd_table(result, tab + j * FT_STRIDE, gtab + j * 32u);
    result = pt_select(result, sum, active);
  }
  return pt_select(result, identity(), early);
}

// Ed25519 fixed-base: out[i] = multiply(base, scalars[i]) from the LDS addend table.
// A lane performs one addition per set bit of its scalar, so a wavefront runs for the largest
// popcount among its 64 lanes.  The workgroup therefore bins its 256 scalars by popcount (counting
// sort through LDS) and hands each wavefront one quartile -- rotated by workgroup so that no SIMD
// always gets the heavy quartile: the four wavefronts then run about 123 + 128 + 134 + 150
// iterations instead of 4 x 148.  Only the lane -> element assignment changes; every element sees
// exactly the additions the reference performs, in the reference's order.  Three workgroups per CU
// (three wavefronts per SIMD): 168 VGPRs without a spill (padd_table) and 43 KiB of LDS -- the results
// are staged out through the table's own LDS region once every lane is done.
__global__ __launch_bounds__(TPB, 3) void k_ed_fixed_base(const u32* __restrict__ scalars,
                                                       const u32* __restrict__ base,
                                                       const u32* __restrict__ table,
                                                       u32* __restrict__ out, size_t n,
                                                       const u32* __restrict__ prefix, int wbits) {
  __shared__ u32 lds_k[8 * TPB];
  __shared__ u32 lds_t[256 * FT_STRIDE];   // the addend table (see padd_table); reused to stage the results out once every lane is done
  __shared__ int lds_bin[260];
  __shared__ unsigned short lds_perm[TPB];
  const int valid = block_valid(n);
  const size_t first = (size_t)blockIdx.x * TPB;
  const int e = threadIdx.x;
  stage_in<8>(lds_k, scalars + first * 8, valid);
  static_assert(TPB == 256, "one thread stages one table entry");
  {  // entry e of the 32 KiB table (L2-resident): y - x, y + x, z, t, x.w[0], y.w[0]
    const ed::pt q = ld_words(table + (size_t)e * 32);
    const fe ymx = ed::sub(q.y, q.x), ypx = ed::add(q.y, q.x);
    u32* d = lds_t + e * FT_STRIDE;
    FEC_UNROLL for (int i = 0; i < 8; ++i) {
      d[i] = ymx.w[i];
      d[8 + i] = ypx.w[i];
      d[16 + i] = q.z.w[i];
      d[24 + i] = q.t.w[i];
    }
    d[32] = q.x.w[0];
    d[33] = q.y.w[0];
  }
  for (int v = e; v < 260; v += TPB) lds_bin[v] = 0;
  __syncthreads();
  // ---- counting sort of the workgroup's elements by popcount ----
  int pc = 0;   // additions of the element + 1 (the first one is a copy, or the prefix-table entry)
  if (e < valid) {
    const u32 k0 = lds_k[e], low = wbits > 0 ? k0 & ((1u << wbits) - 1u) : 0u;
    pc = __builtin_popcount(k0 ^ low) + (low != 0 ? 1 : 0);
    FEC_UNROLL for (int w = 1; w < 8; ++w) pc += __builtin_popcount(lds_k[w * TPB + e]);
  }
  atomicAdd(&lds_bin[pc + 1], 1);  // padding lanes count as popcount 0 and sort to the front
  __syncthreads();
  if (e < 64) {  // inclusive prefix over the 257 bins by one wavefront: 5 bins per lane, then a wave scan
    int loc[5], sum = 0;
    FEC_UNROLL for (int j = 0; j < 5; ++j) {
      const int idx = e * 5 + j;
      loc[j] = idx < 258 ? lds_bin[idx] : 0;
      sum += loc[j];
    }
    int run = sum;
    FEC_UNROLL for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(run, d, 64);
      if (e >= d) run += up;
    }
    int excl = run - sum;
    FEC_UNROLL for (int j = 0; j < 5; ++j) {
      const int idx = e * 5 + j;
      excl += loc[j];
      if (idx < 258) lds_bin[idx] = excl;  // lds_bin[b + 1] = number of elements with popcount <= b
    }
  }
  __syncthreads();
  const int pos = atomicAdd(&lds_bin[pc], 1);  // lds_bin[pc] = first slot of this popcount
  lds_perm[pos] = (unsigned short)e;
  __syncthreads();
  // wavefront w of workgroup b takes quartile (w + b) mod 4 of the sorted list
  const int slot = ((((e >> 6) + (int)blockIdx.x) & 3) << 6) | (e & 63);
  const int src = lds_perm[slot];
  ed::pt r = ed::identity();
  if (src < valid) {
    ed::pt b = ld_words(base);
    r = multiply_fixed_in_place(b, lds_t, table, lds_k + src, prefix, wbits);
  }
  __syncthreads();                      // every lane has read its last table entry
  static_assert(256 * FT_STRIDE >= 32 * TPB, "the table region holds the staged results");
  if (src < valid) st_lds(lds_t + src, TPB, r);
  __syncthreads();
  stage_out<32>(out + first * 32, lds_t, valid);
}

// ---- the same kernel for large batches: elements sorted by popcount over the WHOLE batch -------------------------
// With the in-kernel sort a wavefront's 64 lanes come from one quartile of 256 elements and run for that quartile's
// largest popcount: 134 iterations on average against 128 additions per element.  For batches of 2^16 elements and
// more three small kernels first sort the element INDICES of the whole batch by descending popcount (counting sort:
// histogram, 257-bin scan, scatter; workgroups of 1024 threads x 4 elements with LDS histograms, so that the global
// atomics are one per non-empty bin per workgroup), and the table kernel walks that permutation: every wavefront's
// lanes then have (almost always) the same popcount, no lane idles, and the workgroups come heaviest first.  Only
// the lane -> element assignment changes; results go straight to the elements' own output slots.
namespace {
constexpr int SORT_T = 1024, SORT_E = 4;   // threads per sorting workgroup, elements per thread
constexpr size_t ED_SORT_MIN = (size_t)1 << 16;
constexpr int ED_BINS = 257;               // popcount 0 .. 256
// Work-area header in front of the permutation: hist[ED_BINS] at int 0, cursor[ED_BINS] at int ED_CURSOR_AT.
// (Round 2 had cursor at int 260 and the permutation at byte 2048: cursor[252..256] shared their words with
// perm[0..4], so a batch with scalars of popcount >= 252 in two sort blocks scattered through clobbered cursors.)
constexpr int ED_CURSOR_AT = 264;
constexpr size_t ED_PERM_OFFSET = 4096;    // bytes
static_assert(ED_CURSOR_AT >= ED_BINS && (size_t)(ED_CURSOR_AT + ED_BINS) * sizeof(int) <= ED_PERM_OFFSET,
              "hist, cursor and the permutation must not overlap");

// additions of element g + 1: its set bits, the low wbits bits counting as one when a prefix table answers them
FEC_DEV int scalar_popcount(const u32* scalars, size_t g, int wbits) {
  const uint4* k = reinterpret_cast<const uint4*>(scalars + g * 8);
  uint4 a = k[0];
  const uint4 b = k[1];
  const u32 low = wbits > 0 ? a.x & ((1u << wbits) - 1u) : 0u;
  a.x ^= low;
  return (low != 0 ? 1 : 0) + __builtin_popcount(a.x) + __builtin_popcount(a.y) + __builtin_popcount(a.z) + __builtin_popcount(a.w) +
         __builtin_popcount(b.x) + __builtin_popcount(b.y) + __builtin_popcount(b.z) + __builtin_popcount(b.w);
}
}  // namespace

// hist[b] += number of elements with popcount b (hist zeroed by the launcher)
__global__ __launch_bounds__(SORT_T) void k_ed_pc_hist(const u32* __restrict__ scalars, size_t n, int* __restrict__ hist, int wbits) {
  __shared__ int lh[ED_BINS];
  for (int v = threadIdx.x; v < ED_BINS; v += SORT_T) lh[v] = 0;
  __syncthreads();
  const size_t first = (size_t)blockIdx.x * SORT_T * SORT_E;
  FEC_UNROLL for (int k = 0; k < SORT_E; ++k) {
    const size_t g = first + (size_t)k * SORT_T + threadIdx.x;
    if (g < n) atomicAdd(&lh[scalar_popcount(scalars, g, wbits)], 1);
  }
  __syncthreads();
  for (int v = threadIdx.x; v < ED_BINS; v += SORT_T)
    if (lh[v]) atomicAdd(&hist[v], lh[v]);
}
// cursor[b] = number of elements with popcount > b: the first position of bin b in descending order
__global__ __launch_bounds__(64) void k_ed_pc_scan(const int* __restrict__ hist, int* __restrict__ cursor) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int run = 0;
  for (int b = ED_BINS - 1; b >= 0; --b) {
    cursor[b] = run;
    run += hist[b];
  }
}
// perm[position] = element index, positions handed out bin by bin (order inside a bin is immaterial)
__global__ __launch_bounds__(SORT_T) void k_ed_pc_scatter(const u32* __restrict__ scalars, size_t n, int* __restrict__ cursor,
                                                        u32* __restrict__ perm, int wbits) {
  __shared__ int lh[ED_BINS];
  __shared__ int lbase[ED_BINS];
  for (int v = threadIdx.x; v < ED_BINS; v += SORT_T) lh[v] = 0;
  __syncthreads();
  const size_t first = (size_t)blockIdx.x * SORT_T * SORT_E;
  int pc[SORT_E], rank[SORT_E];
  FEC_UNROLL for (int k = 0; k < SORT_E; ++k) {
    const size_t g = first + (size_t)k * SORT_T + threadIdx.x;
    pc[k] = -1;
    rank[k] = 0;
    if (g < n) {
      pc[k] = scalar_popcount(scalars, g, wbits);
      rank[k] = atomicAdd(&lh[pc[k]], 1);
    }
  }
  __syncthreads();
  for (int v = threadIdx.x; v < ED_BINS; v += SORT_T) lbase[v] = lh[v] ? atomicAdd(&cursor[v], lh[v]) : 0;
  __syncthreads();
  FEC_UNROLL for (int k = 0; k < SORT_E; ++k) {
    const size_t g = first + (size_t)k * SORT_T + threadIdx.x;
    if (pc[k] >= 0) {
      const size_t pos = (size_t)lbase[pc[k]] + (size_t)rank[k];
      if (pos < n) perm[pos] = (u32)g;  // (always true: the cursors partition [0, n))
    }
  }
}

// k_ed_fixed_base over the permutation: lane e of workgroup b computes element perm[b * TPB + e]
__global__ __launch_bounds__(TPB, 3) void k_ed_fixed_sorted(const u32* __restrict__ scalars,
                                                         const u32* __restrict__ base,
                                                         const u32* __restrict__ table,
                                                         const u32* __restrict__ perm,
                                                         u32* __restrict__ out, size_t n,
                                                         const u32* __restrict__ prefix, int wbits) {
  __shared__ u32 lds_k[8 * TPB];
  __shared__ u32 lds_t[256 * FT_STRIDE];
  const int valid = block_valid(n);
  const int e = threadIdx.x;
  const size_t g = e < valid ? (size_t)perm[(size_t)blockIdx.x * TPB + e] : 0;
  if (e < valid) {
    const uint4* k = reinterpret_cast<const uint4*>(scalars + g * 8);
    const uint4 a = k[0], b = k[1];
    lds_k[0 * TPB + e] = a.x; lds_k[1 * TPB + e] = a.y; lds_k[2 * TPB + e] = a.z; lds_k[3 * TPB + e] = a.w;
    lds_k[4 * TPB + e] = b.x; lds_k[5 * TPB + e] = b.y; lds_k[6 * TPB + e] = b.z; lds_k[7 * TPB + e] = b.w;
  }
  static_assert(TPB == 256 && KSTRIDE == TPB, "one thread stages one table entry; scalar columns at stride TPB");
  {  // entry e of the table: y - x, y + x, z, t, x.w[0], y.w[0] (see padd_table)
    const ed::pt q = ld_words(table + (size_t)e * 32);
    const fe ymx = ed::sub(q.y, q.x), ypx = ed::add(q.y, q.x);
    u32* d = lds_t + e * FT_STRIDE;
    FEC_UNROLL for (int i = 0; i < 8; ++i) {
      d[i] = ymx.w[i];
      d[8 + i] = ypx.w[i];
      d[16 + i] = q.z.w[i];
      d[24 + i] = q.t.w[i];
    }
    d[32] = q.x.w[0];
    d[33] = q.y.w[0];
  }
  __syncthreads();
  if (e < valid) {
    const ed::pt b = ld_words(base);
    st_glb(out + g * 32, multiply_fixed_in_place(b, lds_t, table, lds_k + e, prefix, wbits));
  }
}

// ---- Add / double with their operands left in memory (register budget of three wavefronts per SIMD) ----
// the running result sits in the element's slot of the OUTPUT array (32 consecutive words), the addend in its LDS
// slot (word w at l[w * stride]).  Coordinates are loaded where they are used; the early-outs (identity operands,
// opposite points: improbable after an element's first addition) re-read both points inside their rare branch.
// Same products, operands and order as ed::padd / ed::pdbl.
namespace {
FEC_DEV fe ld_gcoord(const u32* g, int c) {
  const uint4* s4 = reinterpret_cast<const uint4*>(g + 8 * c);
  const uint4 lo = s4[0], hi = s4[1];
  fe a;
  a.w[0] = lo.x; a.w[1] = lo.y; a.w[2] = lo.z; a.w[3] = lo.w;
  a.w[4] = hi.x; a.w[5] = hi.y; a.w[6] = hi.z; a.w[7] = hi.w;
  return a;
}
FEC_DEV fe ld_lcoord(const u32* l, int stride, int c) {
  fe a;
  FEC_UNROLL for (int i = 0; i < 8; ++i) a.w[i] = l[(8 * c + i) * stride];
  return a;
}
// result (global) + addend (LDS)
FEC_DEV ed::pt padd_mem(const u32* gr, const u32* la, int stride) {
  using namespace ed;
  // the early-outs sit behind one-word tests every lane taking one must pass (is_identity: x == 0; negation test:
  // p.y == q.y); the exact masks are evaluated inside the rare branch
  lmask maybe;
  fe a, b, d;
  {
    const fe px = ld_gcoord(gr, 0), py = ld_gcoord(gr, 1), qx = ld_lcoord(la, stride, 0), qy = ld_lcoord(la, stride, 1);
    {
      const fe pz = ld_gcoord(gr, 2), qz = ld_lcoord(la, stride, 2);
      d = mul(pz, qz);
    }
    maybe = lanes_where(px.w[0] == 0u || qx.w[0] == 0u || py.w[0] == qy.w[0]);
    a = mul(sub(py, px), sub(qy, qx));
    b = mul(add(py, px), add(qy, qx));
  }
  __builtin_amdgcn_sched_barrier(0);
  fe c;
  {
    const fe pt_ = ld_gcoord(gr, 3), qt = ld_lcoord(la, stride, 3);
    c = mul(mul(pt_, qt), D_());
  }
  __builtin_amdgcn_sched_barrier(0);
  const fe ee = sub(b, a), f = sub(d, c), g = add(d, c), h = add(b, a);
  pt o;
  o.x = mul(ee, f);
  o.y = mul(g, h);
  o.t = mul(ee, h);
  o.z = mul(f, g);
  if (__builtin_expect(maybe != 0, 0)) {  // an element's first addition (result = identity); improbable afterwards
    pt p, q;
    p.x = ld_gcoord(gr, 0); p.y = ld_gcoord(gr, 1); p.z = ld_gcoord(gr, 2); p.t = ld_gcoord(gr, 3);
    q.x = ld_lcoord(la, stride, 0); q.y = ld_lcoord(la, stride, 1); q.z = ld_lcoord(la, stride, 2); q.t = ld_lcoord(la, stride, 3);
    const lmask opposite = fe_eq(p.x, neg(q.x)) & fe_eq(p.y, q.y);  // 1878, raw coordinates
    const lmask idp = is_identity(p), idq = is_identity(q);       // 1785-1791
    o = pt_select(o, identity(), uniform_mask(opposite));
    o = pt_select(o, p, uniform_mask(idq));
    o = pt_select(o, q, uniform_mask(idp));
  }
  return o;
}
// addend.double() = addend + addend (1828-1832), addend in LDS
FEC_DEV ed::pt pdbl_mem(const u32* la, int stride) {
  using namespace ed;
  // Add's early-outs with q = p: is_identity needs x == 0; the negation test x == -x needs word 0 of x to equal
  // word 0 of neg(x), which is 0 for x == 0 and 0xFFFFFFED - x.w[0] otherwise (547-570)
  lmask maybe;
  fe a, b, d;
  {
    const fe x = ld_lcoord(la, stride, 0), y = ld_lcoord(la, stride, 1);
    {
      const fe z = ld_lcoord(la, stride, 2);
      d = sqr_exact(z);
    }
    __builtin_amdgcn_sched_barrier(0);
    maybe = lanes_where(x.w[0] == 0u || x.w[0] == 0xFFFFFFEDu - x.w[0]);
    a = sqr_exact(sub(y, x));
    b = sqr_exact(add(y, x));
  }
  __builtin_amdgcn_sched_barrier(0);
  fe c;
  {
    const fe t = ld_lcoord(la, stride, 3);
    c = mul(sqr_exact(t), D_());
  }
  __builtin_amdgcn_sched_barrier(0);
  const fe ee = sub(b, a), f = sub(d, c), g = add(d, c), h = add(b, a);
  pt o;
  o.x = mul(ee, f);
  o.y = mul(g, h);
  o.t = mul(ee, h);
  o.z = mul(f, g);
  if (__builtin_expect(maybe != 0, 0)) {
    pt p;
    p.x = ld_lcoord(la, stride, 0); p.y = ld_lcoord(la, stride, 1); p.z = ld_lcoord(la, stride, 2); p.t = ld_lcoord(la, stride, 3);
    const lmask opposite = fe_eq(p.x, neg(p.x));                 // Add's test with q = p: x == -x
    const lmask idp = is_identity(p);
    o = pt_select(o, identity(), uniform_mask(opposite));
    o = pt_select(o, p, uniform_mask(idp));
  }
  return o;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------
// One workgroup of TWELVE wavefronts per CU (three per SIMD, 168 VGPRs) owns a contiguous RANGE of elements and
// keeps PS = 864 of them in slots -- the addend in LDS (128 B per slot, 108 KiB), the running result in the
// element's slot of the OUTPUT array (read and rewritten by ~128 additions per element; L2 / Infinity Cache
// traffic) -- refilling a slot from the range the moment its element finishes: no
// workgroup tail until the whole range is done.  Add and double read their operands from memory where they are
// used (padd_mem, pdbl_mem).  History of this round per 2^20 batch: 512 elements per workgroup, four wavefronts,
// result in the output array 22.9 ms; persistent, eight wavefronts, addend AND result in LDS (592 slots fill the
// 160 KiB, so three wavefronts per SIMD could not fit) 22.4 ms; this form, see the measurement in DESIGN.md.
// Each element still sees exactly the reference's operation sequence.
// ---------------------------------------------------------------------------------------------------
namespace {
constexpr int PT = 768;     // threads per workgroup: 12 wavefronts, three per SIMD
#ifndef FEC_ED_PS
#define FEC_ED_PS 864
#endif
constexpr int PS_MAIN = FEC_ED_PS;  // element slots per workgroup (12 x 64 in flight + 96 queued).  With the lock-free rings (profiles/sched_r04/ab_libs_r04b.txt):
                                    // 800 -> 18.09 ms, 832 -> 17.89, 864 -> 17.20, 880 -> 16.95 / 17.46 on two boxes, 896 -> 17.49.  Round 3's sweeps with the lock (profiles/slot_sweep_r03.txt), ms and
                                    // L2-side traffic per 2^20: 1024 -> 17.87 / 27.0 GB, 960 -> 19.05, 896 -> 18.43, 864 -> 18.10, 832 -> 17.75 / 18.4 GB, 800 -> 19.95, 768 -> 19.02
// The second instantiation, 1 024 slots: for launches whose workgroups get a little more than a whole number of
// PS_MAIN-element fills (2^18 elements: 1 024 per workgroup, 6.6 ms against 5.0) -- see kernels_p256.hip: wide_slots_pay.
constexpr int PS_WIDE = 1024;
}  // namespace

template <int PS>
__global__ __launch_bounds__(PT, 1) void k_ed_mul_pers(const u32* __restrict__ scalars, const u32* __restrict__ points,
                                                    u32* __restrict__ out, size_t n, unsigned per_wg,
                                                    unsigned* __restrict__ err, unsigned force_fault) {
  constexpr int RING = PS < 1024 ? 1024 : 2048;   // ring positions: more than there are slots (sched_lf.hpp)
  static_assert(PS <= 1024, "a ring entry holds a ten-bit slot number");
  __shared__ u32 lds_ad[32 * PS];              // addend of slot e: word w at lds_ad[w * PS + e]; the running result
                                               // lives in the element's slot of `out`
  __shared__ u32 lds_gid[PS];                  // element of slot e, relative to the workgroup's range
  __shared__ unsigned short lds_step[PS];      // current step i of slot e (A_i / D_i pending)
  __shared__ __attribute__((aligned(16))) int lds_lf[LF_INTS<RING>];   // control words + the rings D, A, F (sched_lf.hpp)
  const size_t lo = (size_t)blockIdx.x * per_wg;
  const int range = (n - lo) < (size_t)per_wg ? (int)(n - lo) : (int)per_wg;
  const int tid = threadIdx.x, lane = tid & 63;
  // control words through an LDS-address-space pointer in ONE opaque base register: otherwise every
  // word's (link-time constant, > 64 KiB) LDS address is hoisted into a VGPR of its own -- registers the
  // three-wavefront budget does not have
  lds_int_ptr ctl = (lds_int_ptr)lds_lf;
  asm volatile("" : "+v"(ctl));
  const unsigned ctl_addr = (unsigned)(size_t)ctl;
  // every slot starts in the free ring: the wavefronts' first pops are claims of 64 elements each
  lf_init<RING>(lds_lf, tid, PT, range < PS ? range : PS, force_fault ? (unsigned)FEC_DEVERR_FORCED : 0u);
  __syncthreads();

  // Claims the next element of the range for slot `e` (lane-private): loads its point as the addend, sets
  // result = identity, step = 0.  multiply's early-outs (2063-2066: identity point or zero scalar) are
  // answered at once and the slot takes the next element.  Returns the first pending operation
  // (LF_NXT_A = A_0 if bit 0 is set, else LF_NXT_D = D_0), or LF_NXT_DEAD when the range is used up (the slot dies).
  auto claim = [&](int e) -> int {
    for (;;) {
      const int rel = lds_fetch_add(ctl, LF_NEXT, 1);
      if (rel >= range) return LF_NXT_DEAD;
      const size_t g = lo + rel;
      const ed::pt base = ld_glb(points + g * 32);
      u32 any = 0;
      FEC_UNROLL for (int w = 0; w < 8; ++w) any |= scalars[g * 8 + w];
      if (any == 0 || lane_of(ed::is_identity(base))) {
        st_glb(out + g * 32, ed::identity());
        continue;
      }
      st_lds(lds_ad + e, PS, base);
      lds_gid[e] = (u32)rel;
      st_glb(out + g * 32, ed::identity());
      lds_step[e] = 0;
      return (scalars[g * 8] & 1u) ? LF_NXT_A : LF_NXT_D;
    }
  };

  int e = 0;
  int nxt = LF_NXT_NONE;  // the element's next step: ring D doubling only (bit clear), ring A addition then doubling (bit set)
  unsigned watchdog = 0;
  for (;;) {
    // hand on what the last batch left (sched_lf.hpp: no lock, no turn to wait for; the release fence in front of the
    // publishing add orders this batch's result stores before the entries that hand the slots on), take the next one
#ifndef FEC_ED_NO_STORE_DRAIN
    // this batch's result stores (global memory) have COMPLETED before the slots are handed on: the next holder of a slot
    // is another wavefront of this CU, and although this CU's vector-memory pipeline keeps a store and a later load of
    // the same line in order (which is what LLVM's memory model relies on when it emits no vmcnt wait for a
    // workgroup-scope release), the hand-over does not have to lean on that -- the wait costs nothing measurable
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    lf_push<RING>(ctl_addr, lane, nxt, e, (u32)FEC_DEVERR_SCHED_WATCHDOG);
    nxt = LF_NXT_NONE;
    const LfPop pop = lf_pop<RING>(ctl_addr, lane, watchdog, (u32)FEC_DEVERR_SCHED_WATCHDOG);
    if (pop.kind < 0) break;
    if (!lf_entry<RING>(ctl_addr, pop, lane, (u32)FEC_DEVERR_SCHED_WATCHDOG, e)) break;   // inactive lanes compute on slot 0: never stored
    const bool active = lane < pop.count;
    if (pop.kind == LF_Q_F) {   // free slots: each takes the next element of the range (64 claims at full width)
      if (active) nxt = claim(e);
      continue;
    }
    const int kind = pop.kind;
    int step = active ? lds_step[e] : 0;
    // Every global address below is formed from this index.  It is written by claim() and always < range; the test
    // keeps a broken queue (which the design excludes and the watchdog would report) from ever addressing memory
    // outside the workgroup's own range: such a lane works on element 0 of the range, stores nothing, and raises the error.
    u32 gid = active ? lds_gid[e] : 0u;
    const bool oob = gid >= (u32)range;
    gid = oob ? 0u : gid;
    const bool live = active && !oob;
    bool fin = false;
    // ONE task = one step i of the reference's loop (2075-2091) for 64 elements that agree on scalar bit i:
    //   ring A (bit set):   A_i: result = result + addend (2083-2086), then D_i
    //   ring D (bit clear): D_i: addend = addend.double() (2089)
    // The doubling of step 255 is never used and never done.  (Round 2 queued A_i and D_i separately: 383 visits of
    // the scheduler per element instead of 255 for the same arithmetic.)
    if (kind == LF_Q_A) {  // the result stays in its output slot
      // inactive lanes read element 0 of the range (always present) and slot 0: computed, never stored
      u32* slot = out + (lo + gid) * 32;
      FEC_MARK("task_add_begin");
      const ed::pt res = padd_mem(slot, lds_ad + e, PS);
      FEC_MARK("task_add_end");
      if (live) st_glb(slot, res);
    }
    {
      FEC_MARK("task_double_begin");
      const ed::pt d = pdbl_mem(lds_ad + e, PS);
      FEC_MARK("task_double_end");
      if (live) {
        if (step == 255) {  // (only reached through ring A: A_255 was the element's last operation)
          fin = true;
        } else {
          st_lds(lds_ad + e, PS, d);
          ++step;
          const u32 bit = scalar_bit(scalars, lo + gid, step);
          lds_step[e] = (unsigned short)step;
          fin = !bit && step == 255;
          nxt = bit ? LF_NXT_A : LF_NXT_D;
        }
      }
    }
    if (fin) nxt = LF_NXT_FREE;  // the element is done (its result is in place): the slot joins the free ring
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(oob) != 0, 0)) {
      lf_raise(ctl_addr, lane, (u32)FEC_DEVERR_SCHED_INDEX);
      nxt = LF_NXT_NONE;
    }
    // slots (LDS addend, output-array result) are lane-private between the pop and the push; every access to an
    // element's output slot comes from THIS workgroup (one CU, one vector L1)
  }
  __syncthreads();
  if (const int ec = lds_lf[LF_ERR]) {
    // Scheduler fault (watchdog, index guard, or the debug hook): the workgroup's results are not trustworthy.  They are
    // zero-filled AND the ctx's error word is set, which the host reads after its synchronisation: the call returns
    // FEC_E_LAUNCH (fecgpu.hip: sync_and_check, fec_ctx_check) instead of FEC_OK with plausible-looking points.
    ed::pt z;
    z.x = z.y = z.z = z.t = fe_zero();
    for (int el = tid; el < range; el += PT) st_glb(out + (lo + el) * 32, z);
    if (tid == 0 && err != nullptr) {  // plain store into pinned host memory (no PCIe atomic needed: any non-zero value is the signal)
      *reinterpret_cast<volatile unsigned*>(err) = (unsigned)ec;
      __threadfence_system();
    }
  }
}

void ed_build_table_launch(const u32* base, u32* table, hipStream_t s) {
  hipLaunchKernelGGL(k_ed_build_table, dim3(1), dim3(64), 0, s, base, table);
}
size_t ed_fixed_work_bytes(size_t n) { return n >= ED_SORT_MIN ? ED_PERM_OFFSET + n * sizeof(u32) : 0; }
void ed_fixed_launch(const SchedEnv& env, const u32* scalars, const u32* base, const u32* table, u32* out, size_t n, void* work,
                     hipStream_t s) {
  const unsigned grid = (unsigned)((n + TPB - 1) / TPB);
  // the generator's prefix table (multiply_fixed_in_place), when the ctx has one and `base` is the generator
  const bool tab = base == env.gen[FEC_ED25519] && env.gen_prefix[FEC_ED25519] != nullptr && env.gen_prefix_bits[FEC_ED25519] > 0;
  const u32* prefix = tab ? env.gen_prefix[FEC_ED25519] : nullptr;
  const int wbits = tab ? (int)env.gen_prefix_bits[FEC_ED25519] : 0;
  if (n < ED_SORT_MIN || work == nullptr) {  // small batch (or no work area): quartiles of each workgroup's own 256 elements
    hipLaunchKernelGGL(k_ed_fixed_base, dim3(grid), dim3(TPB), 0, s, scalars, base, table, out, n, prefix, wbits);
    return;
  }
  int* hist = static_cast<int*>(work);
  int* cursor = hist + ED_CURSOR_AT;
  u32* perm = reinterpret_cast<u32*>(static_cast<char*>(work) + ED_PERM_OFFSET);
  const unsigned sgrid = (unsigned)((n + (size_t)SORT_T * SORT_E - 1) / ((size_t)SORT_T * SORT_E));
  (void)hipMemsetAsync(hist, 0, ED_PERM_OFFSET, s);
  hipLaunchKernelGGL(k_ed_pc_hist, dim3(sgrid), dim3(SORT_T), 0, s, scalars, n, hist, wbits);
  hipLaunchKernelGGL(k_ed_pc_scan, dim3(1), dim3(64), 0, s, (const int*)hist, cursor);
  hipLaunchKernelGGL(k_ed_pc_scatter, dim3(sgrid), dim3(SORT_T), 0, s, scalars, n, cursor, perm, wbits);
  hipLaunchKernelGGL(k_ed_fixed_sorted, dim3(grid), dim3(TPB), 0, s, scalars, base, table, (const u32*)perm, out, n, prefix, wbits);
}

void ed_launch_mul(const SchedEnv& env, const u32* scalars, const u32* points, u32* out, size_t n, hipStream_t s,
                   unsigned cu_divisor) {
  // one workgroup per CU (or per cu_divisor-th CU) of the ctx's own device, each with a contiguous range of at least
  // 64 elements
  const unsigned cus = env.cus ? env.cus : 256u;
  size_t grid = (n + 63) / 64;
  const unsigned cap = cu_divisor > 1 && cus >= cu_divisor ? cus / cu_divisor : cus;
  if (grid > cap) grid = cap;
  const unsigned per_wg = (unsigned)((n + grid - 1) / grid);
  grid = (n + per_wg - 1) / per_wg;
  // PS_WIDE when the workgroups' elements are a little more than a whole number of PS_MAIN-element fills and a (near)
  // whole number of PS_WIDE-element ones, up to three fills (the rule of kernels_p256.hip: wide_slots_pay)
  bool wide = false;
  if (per_wg > (unsigned)PS_MAIN && per_wg <= 3u * PS_WIDE) {
    auto waste = [per_wg](unsigned q) { return (double)(((per_wg + q - 1) / q) * q) / (double)per_wg; };
    wide = waste(PS_WIDE) + 0.04 < waste(PS_MAIN);
  }
  if (wide)
    hipLaunchKernelGGL((k_ed_mul_pers<PS_WIDE>), dim3((unsigned)grid), dim3(PT), 0, s, scalars, points, out, n, per_wg, env.err,
                       env.force_fault);
  else
    hipLaunchKernelGGL((k_ed_mul_pers<PS_MAIN>), dim3((unsigned)grid), dim3(PT), 0, s, scalars, points, out, n, per_wg, env.err,
                       env.force_fault);
}

}  // namespace fecgpu
