// kernels_ed.hip -- Ed25519 variable-base Curve::multiply (ed25519.rs:2062-2097) as a workgroup-level
// task scheduler (the design of kernels_p256.hip).
//
//   result = identity; addend = point
//   for i in 0..256 { s = result + addend; result = bit(i) ? s : result; addend = addend.double() }
//
// The reference computes `result + addend` every step and keeps it only where the bit is set: the
// additions of clear bits are dead work, and so is the last doubling.  Round 1 executed them all
// (lock-step lanes: 2 x 9 field multiplications per step).  Here every element runs exactly the
// operations whose results are used, in the reference's order -- A_i (only if bit i is set), then D_i
// (for i < 255) -- and a workgroup's four wavefronts pull BATCHES of 64 elements that all need an
// addition or all need a doubling from two ready queues in LDS.  The doubling is the reference's
// double() = self + self (1828-1832), whose four self-products are formed with the exact squaring.
//
// State: the addend of each of the workgroup's E = 512 elements lives in LDS (32 words); the running
// `result` lives in the element's slot of the OUTPUT array (it is read and rewritten by ~128 addition
// batches per element and stays L2-resident; when the element finishes, its slot holds the answer).
#include <hip/hip_runtime.h>

#include "../../include/fecgpu.h"
#include "ed25519.hpp"
#include "staging.hpp"
#include "kernels.hpp"

namespace fecgpu {

namespace {

constexpr int EE = 512;  // elements per workgroup (384 measured: 25.4 vs 25.0 ms, fabric writes 14.1 vs 16.7 GB)
constexpr int RING = 512;  // ring capacity (power of two >= EE)
enum { C_TICKET = 0, C_HEAD_D, C_TAIL_D, C_HEAD_A, C_TAIL_A, C_INFLIGHT, C_REMAIN, C_ERR, C_SERVING, C_WORDS };

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

__global__ __launch_bounds__(TPB, 2) void k_ed_mul_sched(const u32* __restrict__ scalars,
                                                      const u32* __restrict__ points,
                                                      u32* __restrict__ out, size_t n) {
  __shared__ u32 lds_ad[32 * EE];             // addend of element e: word w at lds_ad[w * EE + e]
  __shared__ unsigned short lds_step[EE];     // current step i of element e (A_i / D_i pending)
  __shared__ unsigned short lds_q[2][RING];     // ready rings: [0] needs the doubling D_i, [1] needs the addition A_i
  __shared__ int lds_ctl[C_WORDS];
  const size_t first = (size_t)blockIdx.x * EE;
  const int valid = (n - first) < (size_t)EE ? (int)(n - first) : EE;
  const int tid = threadIdx.x, lane = tid & 63;
  volatile int* ctl = lds_ctl;

  // ---- stage in: addend = point (coalesced 16-byte loads), result = identity in the output slot,
  //      and the element's first pending operation: A_0 if bit 0 is set, else D_0 ----
  for (int v = tid; v < EE * 32 / 4; v += TPB) {
    const int e = (v * 4) / 32, w = (v * 4) % 32;
    uint4 x = make_uint4(0, 0, 0, 0);
    if (e < valid) x = *reinterpret_cast<const uint4*>(points + first * 32 + (size_t)v * 4);
    lds_ad[(w + 0) * EE + e] = x.x;
    lds_ad[(w + 1) * EE + e] = x.y;
    lds_ad[(w + 2) * EE + e] = x.z;
    lds_ad[(w + 3) * EE + e] = x.w;
  }
  if (tid == 0) {
    FEC_UNROLL for (int w = 0; w < C_WORDS; ++w) lds_ctl[w] = 0;
    lds_ctl[C_REMAIN] = valid;
  }
  __syncthreads();
  for (int base = 0; base < EE; base += TPB) {  // initial queues: ordered compaction of the two kinds
    const int e = base + tid;
    int kind0 = 3;
    if (e < valid) {
      st_glb(out + (first + e) * 32, ed::identity());
      lds_step[e] = 0;
      kind0 = scalar_bit(scalars, first + e, 0) ? 1 : 0;
    }
    const lmask m_d = __builtin_amdgcn_ballot_w64(kind0 == 0), m_a = __builtin_amdgcn_ballot_w64(kind0 == 1);
    const lmask below = (1ull << lane) - 1;
    for (int w = 0; w < TPB / 64; ++w) {  // the four wavefronts append in turn
      if ((tid >> 6) == w) {
        const int t_d = ctl[C_TAIL_D], t_a = ctl[C_TAIL_A];
        if (kind0 == 0) lds_q[0][(t_d + __builtin_popcountll(m_d & below)) & (RING - 1)] = (unsigned short)e;
        if (kind0 == 1) lds_q[1][(t_a + __builtin_popcountll(m_a & below)) & (RING - 1)] = (unsigned short)e;
        if (lane == 0) {
          ctl[C_TAIL_D] = t_d + __builtin_popcountll(m_d);
          ctl[C_TAIL_A] = t_a + __builtin_popcountll(m_a);
        }
      }
      __syncthreads();
    }
  }
  // the identity results must be visible to whichever wavefront runs the element's first addition.  Every
  // access to an element's slot comes from THIS workgroup (one CU, one vector L1, write-through), so
  // workgroup-scope ordering is enough; an agent-scope fence would write back / invalidate L2 across the
  // 8 XCDs on every batch (measured: 238 ms instead of 20 ms per 2^20 batch).
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();

  // ---- the scheduler loop (see kernels_p256.hip): push the finished batch, pop the next, compute ----
  int kind = -1, count = 0;
  int e = 0;
  int nxt = 3;  // 0 D-ready, 1 A-ready, 2 finished, 3 none
  unsigned spins = 0;
  for (;;) {
    const lmask m_d = __builtin_amdgcn_ballot_w64(nxt == 0), m_a = __builtin_amdgcn_ballot_w64(nxt == 1);
    const int n_d = __builtin_popcountll(m_d), n_a = __builtin_popcountll(m_a);
    const int n_fin = __builtin_popcountll(__builtin_amdgcn_ballot_w64(nxt == 2));
    const lmask below = (1ull << lane) - 1;
    const int rank_d = __builtin_popcountll(m_d & below), rank_a = __builtin_popcountll(m_a & below);
    if (count == 0) {  // nothing to push: wait OUTSIDE the lock on hints
      const int q_d = ctl[C_TAIL_D] - ctl[C_HEAD_D], q_a = ctl[C_TAIL_A] - ctl[C_HEAD_A];
      const int fl = ctl[C_INFLIGHT], rem = ctl[C_REMAIN];
      int th0 = rem >> 3;
      th0 = th0 < 1 ? 1 : (th0 > 64 ? 64 : th0);
      const bool go = q_d >= th0 || q_a >= th0 || (fl == 0 && (q_d | q_a) != 0) || (rem == 0 && fl == 0) || ctl[C_ERR] != 0;
      if (!go) {
        __builtin_amdgcn_s_sleep(64);
        if (++spins > (1u << 22)) {  // watchdog (~10 s): cannot happen unless the queue logic is broken
          if (lane == 0) ctl[C_ERR] = 1;
          break;
        }
        continue;
      }
    }
    // ---- critical section (FIFO ticket lock, lane 0) ----
    if (lane == 0) {
      const int my = atomicAdd(&lds_ctl[C_TICKET], 1);
      while (ctl[C_SERVING] != my) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int t_d = ctl[C_TAIL_D], t_a = ctl[C_TAIL_A];
    if (nxt == 0) lds_q[0][(t_d + rank_d) & (RING - 1)] = (unsigned short)e;
    if (nxt == 1) lds_q[1][(t_a + rank_a) & (RING - 1)] = (unsigned short)e;
    t_d += n_d;
    t_a += n_a;
    int inflight = ctl[C_INFLIGHT] - count;
    const int remain = ctl[C_REMAIN] - n_fin;
    int h_d = ctl[C_HEAD_D], h_a = ctl[C_HEAD_A];
    const int av_d = t_d - h_d, av_a = t_a - h_a;
    const int err = ctl[C_ERR];
    int th = remain >> 3;
    th = th < 1 ? 1 : (th > 64 ? 64 : th);
    int pick = -1;
    if (av_a >= th && av_a >= av_d) pick = 1;
    else if (av_d >= th) pick = 0;
    else if (av_a >= th) pick = 1;
    else if (inflight == 0 && (av_a | av_d) != 0) pick = av_a > av_d ? 1 : 0;
    int start = 0;
    count = 0;
    if (pick == 0) {
      count = av_d < 64 ? av_d : 64;
      start = h_d;
      h_d += count;
    } else if (pick == 1) {
      count = av_a < 64 ? av_a : 64;
      start = h_a;
      h_a += count;
    }
    inflight += count;
    const bool finished = (remain == 0 && inflight == 0) || err != 0;
    if (lane == 0) {
      ctl[C_TAIL_D] = t_d;
      ctl[C_TAIL_A] = t_a;
      ctl[C_HEAD_D] = h_d;
      ctl[C_HEAD_A] = h_a;
      ctl[C_INFLIGHT] = inflight;
      ctl[C_REMAIN] = remain;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) ctl[C_SERVING] = ctl[C_SERVING] + 1;
    // ---- end of critical section ----
    kind = pick;
    nxt = 3;
    if (finished) break;
    if (kind < 0) continue;
    spins = 0;
    const bool active = lane < count;
    e = active ? lds_q[kind][(start + lane) & (RING - 1)] : 0;
    ed::pt ad = ed::identity();
    if (active) ad = ld_lds(lds_ad + e, EE);
    int step = active ? lds_step[e] : 0;
    if (kind == 1) {  // A_i: result = result + addend  (2083-2086, bit i set)
      u32* slot = out + (first + e) * 32;
      ed::pt r = active ? ld_glb(slot) : ed::identity();
      ed::pt s = ed::padd(r, ad);
      if (active) {
        st_glb(slot, s);
        nxt = step == 255 ? 2 : 0;  // then D_i -- except the last doubling, whose result is never used
      }
    } else {  // D_i: addend = addend.double()  (2089), then step i + 1
      ed::pt d = ed::pdbl(ad);
      if (active) {
        st_lds(lds_ad + e, EE, d);
        ++step;
        lds_step[e] = (unsigned short)step;
        const u32 bit = scalar_bit(scalars, first + e, step);
        nxt = bit ? 1 : (step == 255 ? 2 : 0);
      }
    }
    // a result slot may be picked up by another wavefront of this workgroup next: the workgroup-scope
    // release fence inside the critical section (s_waitcnt vmcnt(0)) orders this batch's stores before
    // the queue entries that hand the elements on
  }
  __syncthreads();
  // ---- the early-outs of multiply (2063-2066): identity point or zero scalar -> identity ----
  for (int el = tid; el < valid; el += TPB) {
    const size_t g = first + el;
    u32 any = 0;
    FEC_UNROLL for (int w = 0; w < 8; ++w) any |= scalars[g * 8 + w];
    const ed::pt base = ld_glb(points + g * 32);
    const bool ident = lane_of(ed::is_identity(base));
    if (lds_ctl[C_ERR] != 0) {  // watchdog fired (cannot happen): all-zero results fail every parity check loudly
      ed::pt z;
      z.x = z.y = z.z = z.t = fe_zero();
      st_glb(out + g * 32, z);
    } else if (any == 0 || ident) {
      st_glb(out + g * 32, ed::identity());
    }
  }
}

void ed_launch_mul(const u32* scalars, const u32* points, u32* out, size_t n, hipStream_t s) {
  const unsigned grid = (unsigned)((n + EE - 1) / EE);
  hipLaunchKernelGGL(k_ed_mul_sched, dim3(grid), dim3(TPB), 0, s, scalars, points, out, n);
}

}  // namespace fecgpu
