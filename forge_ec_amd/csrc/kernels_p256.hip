// kernels_p256.hip -- P-256 Curve::multiply (p256.rs:2120-2156) as a workgroup-level task scheduler.
//
//   for i in 0..256 { result = result.double(); if bit(255 - i) == 1 { result = result + point } }
//
// Every step doubles, but only the lanes whose bit is set need the (twice as expensive) addition.
// Executing the addition for whole wavefronts under a mask wastes half of it; a step-synchronous
// compaction (round 1) packs it but leaves SIMDs idle at three barriers per step (PMC: 40 % of wave
// cycles parked, VALUBusy 82 %, profiles/pmc_r02_base_p256).  Here a workgroup owns E = 512 elements
// whose state (X, Y, Z, step) lives in LDS, and its four wavefronts pull BATCHES from two ready queues:
// 64 elements that all need a doubling, or 64 elements that all need an addition.  Elements therefore
// advance at their own pace (each one still sees exactly the reference's operation sequence, so
// results are bit-identical), every batch is full except in the tail of a workgroup, there is no
// workgroup barrier inside the ladder, and a wavefront holds no point state between batches.
//
// Queues, counters and the step table are LDS words guarded by one LDS ticket lock taken by lane 0 of
// a wavefront once per batch (push the finished batch, pop the next: ~100 cycles against a
// 8 000-15 000-cycle batch).  Forward progress: a wavefront waits only while another one has a batch
// in flight; when nothing is in flight any non-empty queue is handed out as a partial batch.  Waiting
// wavefronts poll the counters WITHOUT the lock (plain loads, long s_sleep) and the lock is FIFO.
#include <hip/hip_runtime.h>

#include "../../include/fecgpu.h"
#include "p256.hpp"
#include "staging.hpp"
#include "kernels.hpp"

namespace fecgpu {

namespace {

constexpr int PE = 512;  // elements per workgroup (384 + 3 workgroups per CU + asm blocks below v168 measured: 21-41 spilled VGPRs, 28.9 vs 28.6 ms)
constexpr int PRING = 512;  // ring capacity (power of two >= PE)
enum { C_TICKET = 0, C_HEAD_D, C_TAIL_D, C_HEAD_A, C_TAIL_A, C_INFLIGHT, C_REMAIN, C_ERR, C_SERVING, C_WORDS };

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

}  // namespace

template <bool FIXED>
__global__ __launch_bounds__(TPB, 2) void k_p256_mul_sched(const u32* __restrict__ scalars,
                                                        const u32* __restrict__ points,
                                                        u32* __restrict__ out, size_t n) {
  __shared__ u32 lds_st[24 * PE];             // X, Y, Z of element e: word w at lds_st[w * PE + e]
  __shared__ u32 lds_k[8 * PE];               // scalar words, same layout
  __shared__ unsigned short lds_step[PE];     // steps completed per element
  __shared__ unsigned short lds_q[2][PRING];     // ready rings: [0] needs a doubling, [1] needs the addition
  __shared__ int lds_ctl[C_WORDS];
  const size_t first = (size_t)blockIdx.x * PE;
  const int valid = (n - first) < (size_t)PE ? (int)(n - first) : PE;
  const int tid = threadIdx.x, lane = tid & 63;
  volatile int* ctl = lds_ctl;

  // ---- stage in: scalars (coalesced 16-byte loads), identity state, queues ----
  for (int v = tid; v < PE * 8 / 4; v += TPB) {
    const int e = (v * 4) / 8, w = (v * 4) % 8;
    uint4 x = make_uint4(0, 0, 0, 0);
    if (e < valid) x = *reinterpret_cast<const uint4*>(scalars + first * 8 + (size_t)v * 4);
    lds_k[(w + 0) * PE + e] = x.x;
    lds_k[(w + 1) * PE + e] = x.y;
    lds_k[(w + 2) * PE + e] = x.z;
    lds_k[(w + 3) * PE + e] = x.w;
  }
  for (int e = tid; e < PE; e += TPB) {
    FEC_UNROLL for (int w = 0; w < 24; ++w) lds_st[w * PE + e] = (w == 8) ? 1u : 0u;  // identity (0, 1, 0)
    lds_step[e] = 0;
    lds_q[0][e] = (unsigned short)e;
  }
  if (tid == 0) {
    lds_ctl[C_TICKET] = 0;
    lds_ctl[C_SERVING] = 0;
    lds_ctl[C_HEAD_D] = 0;
    lds_ctl[C_TAIL_D] = valid;
    lds_ctl[C_HEAD_A] = 0;
    lds_ctl[C_TAIL_A] = 0;
    lds_ctl[C_INFLIGHT] = 0;
    lds_ctl[C_REMAIN] = valid;
    lds_ctl[C_ERR] = 0;
  }
  __syncthreads();


  // ---- the scheduler loop: one iteration = (push the finished batch, pop the next) + compute ----
  int kind = -1, count = 0;   // batch in hand: 0 doubling, 1 addition; `count` active lanes
  int e = 0;                  // this lane's element
  int nxt = 3;                // where this lane's element goes next: 0 D-ready, 1 A-ready, 2 finished, 3 none
  unsigned spins = 0;
  for (;;) {
    const lmask m_d = __builtin_amdgcn_ballot_w64(nxt == 0), m_a = __builtin_amdgcn_ballot_w64(nxt == 1);
    const int n_d = __builtin_popcountll(m_d), n_a = __builtin_popcountll(m_a);
    const int n_fin = __builtin_popcountll(__builtin_amdgcn_ballot_w64(nxt == 2));
    const lmask below = (1ull << lane) - 1;
    const int rank_d = __builtin_popcountll(m_d & below), rank_a = __builtin_popcountll(m_a & below);
    // A wavefront with nothing to push stays OUT of the lock while it waits: it polls the queue
    // counters with plain LDS loads (hints only -- every decision is re-made under the lock) and
    // sleeps, so that idle wavefronts never compete for the lock with the ones doing work.
    if (count == 0) {
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
    // ---- critical section ----
    // ticket lock (FIFO): a test-and-set lock let three polling wavefronts starve the working one
    // for seconds in the tail of a workgroup
    if (lane == 0) {
      const int my = atomicAdd(&lds_ctl[C_TICKET], 1);
      while (ctl[C_SERVING] != my) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int t_d = ctl[C_TAIL_D], t_a = ctl[C_TAIL_A];
    if (nxt == 0) lds_q[0][(t_d + rank_d) & (PRING - 1)] = (unsigned short)e;
    if (nxt == 1) lds_q[1][(t_a + rank_a) & (PRING - 1)] = (unsigned short)e;
    t_d += n_d;
    t_a += n_a;
    int inflight = ctl[C_INFLIGHT] - count;
    const int remain = ctl[C_REMAIN] - n_fin;
    int h_d = ctl[C_HEAD_D], h_a = ctl[C_HEAD_A];
    const int av_d = t_d - h_d, av_a = t_a - h_a;
    const int err = ctl[C_ERR];
    // a batch is handed out when a queue holds a full wavefront's worth -- or, near the end of the
    // workgroup (or in a ragged last workgroup), an eighth of what is left; when nothing is in
    // flight, anything that is ready
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
    if (kind < 0) continue;  // lost the race for the batch the hint promised: back to polling
    spins = 0;
    const bool active = lane < count;
    e = active ? lds_q[kind][(start + lane) & (PRING - 1)] : 0;
    p256::pt p = p256::identity();
    if (active) p = ld_pt(lds_st + e, PE);
    int step = active ? lds_step[e] : 0;
    if (kind == 0) {
      p256::pt o = p256::pdouble(p);
      if (active) {
        st_pt(lds_st + e, PE, o);
        const int b = 255 - step;
        const u32 bit = (lds_k[(b >> 5) * PE + e] >> (b & 31)) & 1u;
        if (bit) {
          nxt = 1;
        } else {
          ++step;
          lds_step[e] = (unsigned short)step;
          nxt = step == 256 ? 2 : 0;
        }
      }
    } else {
      p256::pt q = FIXED ? ld_pt(points, 1) : (active ? ld_base(points, first + e) : p256::identity());
      lmask nd;
      p256::pt s = p256::padd_nodouble(p, q, nd);
      if (__builtin_expect(nd != 0, 0)) {  // Add (1951) returns self.double(): never on random inputs
        p256::pt d2 = p256::pdouble(p);
        s = p256::pt_select(s, d2, nd);
      }
      if (active) {
        st_pt(lds_st + e, PE, s);
        ++step;
        lds_step[e] = (unsigned short)step;
        nxt = step == 256 ? 2 : 0;
      }
    }
  }
  __syncthreads();
  // ---- results: the early-outs of multiply (2121-2124), then coalesced 16-byte stores ----
  for (int el = tid; el < valid; el += TPB) {
    u32 any = 0;
    FEC_UNROLL for (int w = 0; w < 8; ++w) any |= lds_k[w * PE + el];
    u32 zany = 0;
    if (FIXED) {
      FEC_UNROLL for (int w = 0; w < 8; ++w) zany |= points[16 + w];
    } else {
      const uint4* z = reinterpret_cast<const uint4*>(points + (first + el) * 24 + 16);
      const uint4 z0 = z[0], z1 = z[1];
      zany = z0.x | z0.y | z0.z | z0.w | z1.x | z1.y | z1.z | z1.w;
    }
    if (lds_ctl[C_ERR] != 0) {  // watchdog fired (cannot happen): all-zero results fail every parity check loudly
      FEC_UNROLL for (int w = 0; w < 24; ++w) lds_st[w * PE + el] = 0u;
    } else if (any == 0 || zany == 0) {
      FEC_UNROLL for (int w = 0; w < 24; ++w) lds_st[w * PE + el] = (w == 8) ? 1u : 0u;
    }
  }
  __syncthreads();
  for (int v = tid; v < PE * 24 / 4; v += TPB) {
    const int el = (v * 4) / 24, w = (v * 4) % 24;
    if (el < valid) {
      uint4 x;
      x.x = lds_st[(w + 0) * PE + el];
      x.y = lds_st[(w + 1) * PE + el];
      x.z = lds_st[(w + 2) * PE + el];
      x.w = lds_st[(w + 3) * PE + el];
      *reinterpret_cast<uint4*>(out + first * 24 + (size_t)v * 4) = x;
    }
  }
}

void p256_launch_mul(bool fixed, const u32* scalars, const u32* points, u32* out, size_t n, hipStream_t s) {
  const unsigned grid = (unsigned)((n + PE - 1) / PE);
  if (fixed) hipLaunchKernelGGL((k_p256_mul_sched<true>), dim3(grid), dim3(TPB), 0, s, scalars, points, out, n);
  else hipLaunchKernelGGL((k_p256_mul_sched<false>), dim3(grid), dim3(TPB), 0, s, scalars, points, out, n);
}

}  // namespace fecgpu
