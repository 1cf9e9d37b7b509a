// sched_lf.hpp -- the lock-free ready queues of the workgroup task schedulers (kernels_p256.hip, kernels_ed.hip).
//
// Rounds 2-3 guarded two ready rings and their counters with a FIFO ticket lock taken once per batch: measured
// (profiles/sched_stats_r03.txt) a hand-over cost ~1 020 cycles, a batch carried ~150 VALU instructions of scheduler
// (ballots, the ticket poll, seven control words read and rewritten, claim() run by a whole wavefront whenever ONE lane
// finished), and with the tasks stubbed out the kernel still took 10.5 ms of 23.8.  Here nothing is locked and nobody
// polls for a turn (stubbed: 3.1 ms, 88 instructions per batch):
//
//   * THREE rings of slot numbers -- D (next step: the doubling alone), A (doubling and addition), F (free slots: their
//     element has finished) -- each a multi-producer / multi-consumer queue driven by LDS fetch-and-add:
//       producer:  pos = RES += n            (reserve n positions; D and A together in ONE ds_add_rtn_u64)
//                  wait until ring[pos + rank] reads CONSUMED(lap - 1)         (it does, see below)
//                  ring[pos + rank] = slot | lap(pos)
//                  AV += n                   (publish; D and A together in one ds_add_u64)
//       consumer:  old = AV -= want          (one ds_add_rtn_u64; the return value carries BOTH counts)
//                  old < want: it lost a race -- AV += want, look again
//                  pos = HEAD += want
//                  wait until ring[pos + lane] reads WRITTEN(lap)   (reserved by a producer that has not written yet:
//                                                                    a few cycles), take the slot number,
//                  ring[pos + lane] = CONSUMED(lap)
//     AV counts are kept biased (LF_BIAS) so that a half of the 64-bit word never borrows from the other.
//     An entry is a slot number (ten bits), a consumed flag and the lap of its position modulo 32.  A ring holds more
//     positions than there are slots and a slot sits in at most one position, so at most that many positions are
//     unread at any time -- but the SPAN from the oldest unread position to RES is not bounded by that (the other
//     slots go round while one consumer dawdles between its HEAD += and its read), which is why the producer looks
//     before it writes: it never overwrites an entry that has not been consumed, however the wavefronts are
//     scheduled.  No cycle of waits: a producer of lap g waits for consumers of lap g - 1, those for producers of
//     lap g - 1, and so on down to lap 0, whose producers wait for nobody.  (tests/cpp/sched_lf_model.cpp runs this
//     protocol on host threads, which ARE descheduled for long stretches.)
//   * a wavefront pushes what its batch produced and pops the next one from ONE snapshot of {AV_D, AV_A, AV_F, REMAIN}
//     (one ds_read_b128): a ring is taken when it holds a full batch of 64 -- while fewer than 256 slots are live:
//     REMAIN / 4 entries (FEC_LF_TAIL_SHIFT) -- free slots first (the claim of 64 elements runs at full width instead
//     of for a whole wavefront per finished lane), then the fuller of D / A; otherwise the wavefront sleeps.
//     Progress: if every wavefront waits, every live slot is queued, so AV_D + AV_A + AV_F = REMAIN, and three counts
//     below the threshold (<= REMAIN / 4 each) cannot add up to REMAIN.
//   * REMAIN = live slots (a slot dies when claim() finds the range used up); REMAIN == 0 ends the kernel; an error
//     (watchdog, index guard, debug hook) sets LF_ERRFLAG in REMAIN so that every wavefront sees it in the same read.
//
// Ordering: the DS instructions of one wavefront execute in order, so a batch's slot stores precede its ring entries
// and those precede the publishing add; a workgroup-scope release fence (s_waitcnt) in front of the publish covers
// the results the Ed25519 kernel keeps in global memory.
#pragma once
#include "limbs.hpp"

namespace fecgpu {
namespace {

// control block (ints, 16-byte aligned).  Words 0..3 are the snapshot; {AV_D, AV_A} and {RES_D, RES_A} are the two
// 64-bit atomics' operands (8-byte aligned).
enum { LF_AV_D = 0, LF_AV_A, LF_AV_F, LF_REMAIN, LF_RES_D, LF_RES_A, LF_HEAD_D, LF_HEAD_A, LF_RES_F, LF_HEAD_F, LF_ERR, LF_NEXT, LF_WORDS };
#ifndef FEC_LF_SLEEP
#define FEC_LF_SLEEP 32   // s_sleep argument (units of 64 cycles) of a wavefront that found nothing to take
#endif
#ifndef FEC_LF_TAIL_SHIFT
#define FEC_LF_TAIL_SHIFT 2   // a ring is taken when it holds min(64, REMAIN >> this) entries (at least 2: see "Progress")
#endif
constexpr int LF_BIAS = 1 << 16;            // AV_D / AV_A are stored + LF_BIAS (they dip below zero when a consumer loses a race)
constexpr unsigned LF_ERRFLAG = 1u << 30;   // in REMAIN: a wavefront has raised LF_ERR
enum { LF_Q_D = 0, LF_Q_A = 1, LF_Q_F = 2 };
// what a lane hands to lf_push: its slot goes to ring D / A, has died (2), is not there (3), or is free (4)
enum { LF_NXT_D = 0, LF_NXT_A = 1, LF_NXT_DEAD = 2, LF_NXT_NONE = 3, LF_NXT_FREE = 4 };

typedef volatile __attribute__((address_space(3))) int* lds_int_ptr;
// per-lane atomic fetch-and-add on a control word (ds_add_rtn_u32; atomicAdd() on a generic pointer would be a FLAT atomic)
FEC_DEV int lds_fetch_add(lds_int_ptr ctl, int w, int v) {
  return __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)(ctl + w), v, __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The raw DS operations.  `addr` is an LDS byte address; the callers run the atomics on lane 0 only.
FEC_DEV u64 lf_add_rtn_u64(unsigned addr, u64 v) {
  u64 r;
  asm volatile("ds_add_rtn_u64 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr), "v"(v) : "memory");
  return r;
}
FEC_DEV void lf_add_u64(unsigned addr, u64 v) { asm volatile("ds_add_u64 %0, %1" : : "v"(addr), "v"(v) : "memory"); }
FEC_DEV u32 lf_add_rtn_u32(unsigned addr, u32 v) {
  u32 r;
  asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr), "v"(v) : "memory");
  return r;
}
FEC_DEV void lf_add_u32(unsigned addr, u32 v) { asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(v) : "memory"); }
FEC_DEV void lf_or_b32(unsigned addr, u32 v) { asm volatile("ds_or_b32 %0, %1" : : "v"(addr), "v"(v) : "memory"); }
FEC_DEV void lf_write_b32(unsigned addr, u32 v) { asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(v) : "memory"); }

FEC_DEV u32 lf_uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
FEC_DEV u32 lf_rank(lmask m) {  // number of set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}
// lap tag of ring position `pos`: (pos / RING) mod 32 in bits 11..15 of an entry; bit 10 = consumed; a slot number has
// ten bits.  (pos - RING wraps correctly at 0: 2^32 is a multiple of 32 * RING.)
constexpr u32 LF_CONSUMED = 0x400u;
template <int RING>
FEC_DEV u32 lf_lap(u32 pos) {
  static_assert(RING == 1024 || RING == 2048, "ring sizes: the tag is (pos / RING) mod 32 in bits 11..15");
  return RING == 1024 ? ((pos << 1) & 0xF800u) : (pos & 0xF800u);
}

// The control block and the three rings are ONE LDS array of LF_INTS<RING> ints (16-byte aligned): control words, then
// ring D, ring A, ring F (RING 16-bit entries each).  Everything is addressed from the block's LDS byte address in
// one register with immediate offsets (an LDS array above 64 KiB otherwise costs a VGPR per address constant, and a
// volatile access through a generic pointer would be a FLAT access).
template <int RING>
constexpr int LF_INTS = LF_WORDS + 3 * RING / 2;
template <int RING>
FEC_DEV unsigned lf_ring_addr(unsigned ctl, int kind, u32 pos) {
  return ctl + 4u * LF_WORDS + (unsigned)kind * (2u * RING) + 2u * (pos & (u32)(RING - 1));
}
FEC_DEV void lf_write_b16(unsigned addr, u32 v) { asm volatile("ds_write_b16 %0, %1" : : "v"(addr), "v"(v) : "memory"); }
FEC_DEV u32 lf_read_u16(unsigned addr) {
  u32 r;
  asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
  return r;
}

// The control block and the rings before the first push: every ring entry reads CONSUMED(lap -1), the F ring holds the
// `live` slots 0 .. live - 1 as WRITTEN(lap 0).  Call from every thread, then __syncthreads().
template <int RING>
FEC_DEV void lf_init(int* ctl_words, int tid, int nthreads, int live, unsigned forced_err) {
  unsigned short* q = reinterpret_cast<unsigned short*>(ctl_words + LF_WORDS);
  for (int i = tid; i < RING; i += nthreads) {
    q[LF_Q_D * RING + i] = 0xFC00u;                        // LF_CONSUMED | lap 31
    q[LF_Q_A * RING + i] = 0xFC00u;
    q[LF_Q_F * RING + i] = i < live ? (unsigned short)i : (unsigned short)0xFC00u;
  }
  if (tid == 0) {
    for (int w = 0; w < LF_WORDS; ++w) ctl_words[w] = 0;
    ctl_words[LF_AV_D] = LF_BIAS;
    ctl_words[LF_AV_A] = LF_BIAS;
    ctl_words[LF_AV_F] = live;
    ctl_words[LF_RES_F] = live;
    ctl_words[LF_REMAIN] = live | (forced_err ? (int)LF_ERRFLAG : 0);
    ctl_words[LF_ERR] = (int)forced_err;
  }
}

// raise an error code: every wavefront leaves at its next snapshot
FEC_DEV void lf_raise(unsigned ctl, int lane, u32 code) {
  if (lane == 0) {
    lf_write_b32(ctl + 4 * LF_ERR, code);
    lf_or_b32(ctl + 4 * LF_REMAIN, LF_ERRFLAG);
  }
}

// One ring entry per lane of `mine`: slot number `slot` at position `pos` of ring `kind`, written once the entry's
// previous occupant (lap - 1) has been consumed -- which it has, unless a consumer is being very slow (see the top of
// this file); then the producer waits for it.  Returns false -- after raising the error -- if that lasts.
template <int RING>
FEC_DEV bool lf_put(unsigned ctl, int lane, bool mine, int kind, u32 pos, u32 slot, u32 watchdog_code) {
  const unsigned addr = lf_ring_addr<RING>(ctl, kind, pos);
  const u32 expect = LF_CONSUMED | lf_lap<RING>(pos - (u32)RING);
  for (unsigned tries = 0;; ++tries) {
    const u32 v = mine ? lf_read_u16(addr) : expect;
    if (__builtin_amdgcn_ballot_w64(v != expect) == 0) break;
    if (tries > (1u << 20)) {
      lf_raise(ctl, lane, watchdog_code);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  if (mine) lf_write_b16(addr, slot | lf_lap<RING>(pos));
  return true;
}

// Hands the slots of this wavefront's lanes on: lane's slot `e` goes where `nxt` says (LF_NXT_*).
template <int RING>
FEC_DEV void lf_push(unsigned ctl, int lane, int nxt, int e, u32 watchdog_code) {
  const lmask m_d = __builtin_amdgcn_ballot_w64(nxt == LF_NXT_D), m_a = __builtin_amdgcn_ballot_w64(nxt == LF_NXT_A);
  if ((m_d | m_a) != 0) {
    const u32 n_d = (u32)__builtin_popcountll(m_d), n_a = (u32)__builtin_popcountll(m_a);
    const u64 both = (u64)n_d | ((u64)n_a << 32);
    u64 old = 0;
    if (lane == 0) old = lf_add_rtn_u64(ctl + 4 * LF_RES_D, both);
    const u32 pos_d = lf_uni((u32)old), pos_a = lf_uni((u32)(old >> 32));
    const u32 pos = nxt == LF_NXT_A ? pos_a + lf_rank(m_a) : pos_d + lf_rank(m_d);
    const bool mine = nxt == LF_NXT_D || nxt == LF_NXT_A;
    if (!lf_put<RING>(ctl, lane, mine, mine ? nxt : LF_Q_D, pos, (u32)e, watchdog_code)) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) lf_add_u64(ctl + 4 * LF_AV_D, both);
  }
  const lmask m_f = __builtin_amdgcn_ballot_w64(nxt == LF_NXT_FREE);
  if (m_f != 0) {
    const u32 n_f = (u32)__builtin_popcountll(m_f);
    u32 old = 0;
    if (lane == 0) old = lf_add_rtn_u32(ctl + 4 * LF_RES_F, n_f);
    const u32 pos = lf_uni(old) + lf_rank(m_f);
    if (!lf_put<RING>(ctl, lane, nxt == LF_NXT_FREE, LF_Q_F, pos, (u32)e, watchdog_code)) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) lf_add_u32(ctl + 4 * LF_AV_F, n_f);
  }
  const lmask m_x = __builtin_amdgcn_ballot_w64(nxt == LF_NXT_DEAD);
  if (m_x != 0) {
    if (lane == 0) lf_add_u32(ctl + 4 * LF_REMAIN, 0u - (u32)__builtin_popcountll(m_x));
  }
}

struct LfPop {
  int kind;    // LF_Q_D / LF_Q_A / LF_Q_F, or -1: the kernel is over (every slot has died, or an error was raised)
  int count;   // lanes 0 .. count - 1 hold an entry
  u32 pos;     // ring position of lane 0's entry
};
// Takes the next batch (blocks -- sleeping -- while there is none).  `watchdog` counts the sleeps and lost races of
// the wavefront's whole life (tools/microbench/sched_stats.hip prints their sum): 2^22 of them cannot happen unless the queue logic is broken, and then raise
// `watchdog_code` instead of hanging the GPU.
template <int RING>
FEC_DEV LfPop lf_pop(unsigned ctl, int lane, unsigned& watchdog, u32 watchdog_code) {
  LfPop r;
  r.kind = -1;
  r.count = 0;
  r.pos = 0;
  for (;;) {
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    v4i_t s;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(s) : "v"(ctl) : "memory");
    const int av_d = (int)lf_uni((u32)s.x) - LF_BIAS, av_a = (int)lf_uni((u32)s.y) - LF_BIAS, av_f = (int)lf_uni((u32)s.z);
    const u32 remain = lf_uni((u32)s.w);
    if (remain == 0u || (remain & LF_ERRFLAG) != 0u) return r;
    int th = (int)(remain >> FEC_LF_TAIL_SHIFT);
    th = th < 1 ? 1 : (th > 64 ? 64 : th);
    // A ring is taken when it holds a full batch -- or, while fewer than 256 slots are live, REMAIN / 4 entries: free
    // slots first (their claims refill the pool), then the fuller of A / D.  Anything less WAITS: a sleeping
    // wavefront costs its SIMD no issue slots (the other two have work), a thin batch costs them a whole task's.
    int pick = -1, want = 0;
    if (av_f >= th) {
      pick = LF_Q_F;
      want = av_f;
    } else {
      const int m = av_a >= av_d ? av_a : av_d;
      if (m >= th) {
        pick = av_a >= av_d ? LF_Q_A : LF_Q_D;
        want = m;
      }
    }
    want = want > 64 ? 64 : want;
    if (pick < 0) {   // nothing to take yet: other wavefronts hold batches and will push
      __builtin_amdgcn_s_sleep(FEC_LF_SLEEP);
      if (++watchdog > (1u << 22)) {
        lf_raise(ctl, lane, watchdog_code);
        return r;
      }
      continue;
    }
    int got;
    if (pick == LF_Q_F) {
      u32 old = 0;
      if (lane == 0) old = lf_add_rtn_u32(ctl + 4 * LF_AV_F, 0u - (u32)want);
      got = (int)lf_uni(old) >= want ? want : 0;   // all or nothing: a thinner batch than the policy chose is not worth a task
      if (got == 0 && lane == 0) lf_add_u32(ctl + 4 * LF_AV_F, (u32)want);
    } else {
      // subtract `want` from one half of {AV_D, AV_A}: the halves are biased, so neither borrows from the other
      const u64 dec = pick == LF_Q_A ? (0ull - ((u64)(u32)want << 32)) : (0ull - (u64)(u32)want);
      u64 old = 0;
      if (lane == 0) old = lf_add_rtn_u64(ctl + 4 * LF_AV_D, dec);
      const int had = (int)lf_uni(pick == LF_Q_A ? (u32)(old >> 32) : (u32)old) - LF_BIAS;
      got = had >= want ? want : 0;
      if (got == 0 && lane == 0) lf_add_u64(ctl + 4 * LF_AV_D, (u64)(u32)want << (pick == LF_Q_A ? 32 : 0));
    }
    if (got == 0) {   // another wavefront was faster: look again
      if (++watchdog > (1u << 22)) {
        lf_raise(ctl, lane, watchdog_code);
        return r;
      }
      continue;
    }
    u32 pos = 0;
    const int head_word = pick == LF_Q_D ? LF_HEAD_D : (pick == LF_Q_A ? LF_HEAD_A : LF_HEAD_F);
    if (lane == 0) pos = lf_add_rtn_u32(ctl + 4 * (unsigned)head_word, (u32)got);
    r.kind = pick;
    r.count = got;
    r.pos = lf_uni(pos);
    return r;
  }
}

// The slot number of this lane's entry of a popped batch (0 for a lane without one); the entry is marked consumed.  An
// entry that does not yet read WRITTEN(lap of its position) has been reserved and not yet written: re-read.  Returns
// false -- after raising the error -- if that lasts (a broken queue).
template <int RING>
FEC_DEV bool lf_entry(unsigned ctl, const LfPop& p, int lane, u32 watchdog_code, int& slot) {
  const u32 at = p.pos + (u32)lane;
  const bool active = lane < p.count;
  const u32 want_tag = lf_lap<RING>(at);
  const unsigned entry = lf_ring_addr<RING>(ctl, p.kind, at);
  u32 v = 0;
  for (unsigned tries = 0;; ++tries) {
    v = active ? lf_read_u16(entry) : 0u;
    if (__builtin_amdgcn_ballot_w64(active && (v & 0xFC00u) != want_tag) == 0) break;
    if (tries > (1u << 20)) {
      lf_raise(ctl, lane, watchdog_code);
      slot = 0;
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  if (active) lf_write_b16(entry, LF_CONSUMED | want_tag);   // the position may be written again (a lap later)
  slot = (int)(v & 1023u);
  return true;
}

}  // namespace
}  // namespace fecgpu
