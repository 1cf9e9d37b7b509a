// sched_lf_model.cpp -- the lock-free ready-queue protocol of forge_ec_amd/csrc/sched_lf.hpp, restated for host threads.
//
// Twelve threads play the twelve wavefronts of a scheduler workgroup; the control words are std::atomic words driven
// by exactly the operations the kernels use (64-bit fetch-add on the pairs {RES_D, RES_A} and {AV_D, AV_A} with the
// AV halves biased, 32-bit fetch-add on the F ring's words and on HEAD / REMAIN / NEXT, ring entries tagged with the
// lap of their position and a consumed flag -- the producer looks before it writes, the consumer marks what it has read --,
// all-or-nothing pops, the batch policy of lf_pop with its tail threshold).  A "task" only
// advances its elements' step counters and draws their next ring at random.  Checked: every element of the range is
// claimed exactly once and advanced exactly `steps` times, no slot is ever in two batches at once, no ring entry is read
// before it is written or overwritten unread, every thread terminates (a watchdog turns a stuck run into a failure), for
// slot counts, ranges and step counts that cover the initial fill, steady state, the tail and the degenerate sizes.
// Host threads are descheduled for long stretches at arbitrary points, which the wavefronts of a workgroup are not: the
// first form of the protocol (entries tagged with a generation only, the producer writing without looking) survived the
// whole GPU suite and soaks and was caught HERE -- a consumer that sleeps between its HEAD += and its read is lapped by the
// other slots going round; hence the consumed flag.
// It is a model (the kernels' code cannot run on the CPU: DS instructions, wavefront ballots), kept next to the GPU
// tests that run the real thing; what it buys is the protocol's logic under thousands of thread interleavings.
//
//   g++ -O2 -std=c++17 -pthread -o tests/cpp/sched_lf_model tests/cpp/sched_lf_model.cpp && tests/cpp/sched_lf_model
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

namespace {

constexpr int kWaves = 12, kLanes = 64;
constexpr int64_t kBias = 1 << 16;
constexpr uint32_t kErrFlag = 1u << 30;
enum { Q_D = 0, Q_A = 1, Q_F = 2 };
enum { NXT_D = 0, NXT_A = 1, NXT_DEAD = 2, NXT_NONE = 3, NXT_FREE = 4 };

struct Workgroup {
  int slots, ring, range, steps;
  std::atomic<uint64_t> res_da{0}, av_da{0};              // low half D, high half A
  std::atomic<uint32_t> res_f{0}, head[3], remain{0}, next{0};
  std::atomic<int32_t> av_f{0};
  std::vector<std::atomic<uint16_t>> q[3];
  // element state per slot (lane-private between a pop and the push that follows, as in the kernels)
  std::vector<int> slot_elem, slot_step;
  std::vector<std::atomic<int>> slot_owner;                // -1 free of a batch, else the wavefront that holds it
  std::vector<std::atomic<int>> elem_claims, elem_steps;
  std::atomic<int> failures{0};

  Workgroup(int slots_, int range_, int steps_)
      : slots(slots_), ring(slots_ < 1024 ? 1024 : 2048), range(range_), steps(steps_), slot_elem(slots_), slot_step(slots_),
        slot_owner(slots_), elem_claims(range_ > 0 ? range_ : 1), elem_steps(range_ > 0 ? range_ : 1) {
    for (auto& h : head) h = 0;
    for (int k = 0; k < 3; ++k) q[k] = std::vector<std::atomic<uint16_t>>(ring);
    const int live = range < slots ? range : slots;
    for (int i = 0; i < ring; ++i) {
      q[Q_D][i] = 0xFC00;   // consumed, lap -1
      q[Q_A][i] = 0xFC00;
      q[Q_F][i] = i < live ? (uint16_t)i : (uint16_t)0xFC00;
    }
    av_da = (uint64_t)kBias | ((uint64_t)kBias << 32);
    av_f = live;
    res_f = (uint32_t)live;
    remain = (uint32_t)live;
    for (auto& o : slot_owner) o = -1;
    for (auto& c : elem_claims) c = 0;
    for (auto& c : elem_steps) c = 0;
  }
  uint32_t lap(uint32_t pos) const { return ring == 1024 ? ((pos << 1) & 0xF800u) : (pos & 0xF800u); }
  // lf_put: the entry's previous occupant (lap - 1) must have been consumed
  bool put(int kind, uint32_t p, int slot) {
    const uint16_t expect = (uint16_t)(0x400u | lap(p - (uint32_t)ring));
    const auto t0 = std::chrono::steady_clock::now();
    while (q[kind][p & (ring - 1)].load(std::memory_order_acquire) != expect) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
        fail("a producer waited 20 s for an entry to be consumed");
        return false;
      }
      std::this_thread::yield();
    }
    q[kind][p & (ring - 1)].store((uint16_t)(slot | lap(p)), std::memory_order_release);
    return true;
  }
  void fail(const char* what) {
    if (failures.fetch_add(1) == 0) std::fprintf(stderr, "FAIL: %s (slots %d, range %d, steps %d)\n", what, slots, range, steps);
    remain.fetch_or(kErrFlag);
  }
};

struct Lane {
  int nxt = NXT_NONE, slot = 0;
};

void push(Workgroup& w, Lane (&lanes)[kLanes]) {
  uint32_t n[5] = {0, 0, 0, 0, 0};
  for (auto& l : lanes) n[l.nxt]++;
  if (n[NXT_D] + n[NXT_A]) {
    const uint64_t both = (uint64_t)n[NXT_D] | ((uint64_t)n[NXT_A] << 32);
    const uint64_t old = w.res_da.fetch_add(both);
    uint32_t pos[2] = {(uint32_t)old, (uint32_t)(old >> 32)};
    for (auto& l : lanes)
      if (l.nxt == NXT_D || l.nxt == NXT_A) {
        const uint32_t p = pos[l.nxt]++;
        if (!w.put(l.nxt, p, l.slot)) return;
      }
    w.av_da.fetch_add(both);
  }
  if (n[NXT_FREE]) {
    uint32_t p = w.res_f.fetch_add(n[NXT_FREE]);
    for (auto& l : lanes)
      if (l.nxt == NXT_FREE) {
        if (!w.put(Q_F, p, l.slot)) return;
        ++p;
      }
    w.av_f.fetch_add((int32_t)n[NXT_FREE]);
  }
  if (n[NXT_DEAD]) w.remain.fetch_sub(n[NXT_DEAD]);
}

// -> kind (or -1: over), count, first position
bool pop(Workgroup& w, int& kind, int& count, uint32_t& pos, std::mt19937& rng) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    const uint64_t av = w.av_da.load();
    const int av_d = (int)((int64_t)(uint32_t)av - kBias), av_a = (int)((int64_t)(uint32_t)(av >> 32) - kBias), av_f = w.av_f.load();
    const uint32_t remain = w.remain.load();
    if (remain == 0 || (remain & kErrFlag)) return false;
    int th = (int)(remain >> 2);
    th = th < 1 ? 1 : (th > 64 ? 64 : th);
    int pick = -1, want = 0;
    if (av_f >= th) {
      pick = Q_F;
      want = av_f;
    } else {
      const int m = av_a >= av_d ? av_a : av_d;
      if (m >= th) {
        pick = av_a >= av_d ? Q_A : Q_D;
        want = m;
      }
    }
    want = want > 64 ? 64 : want;
    if (pick < 0) {
      if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
        w.fail("a wavefront waited for 20 s: the queues are stuck");
        return false;
      }
      if (rng() & 1) std::this_thread::yield();
      continue;
    }
    int got;
    if (pick == Q_F) {
      const int had = w.av_f.fetch_sub(want);
      got = had >= want ? want : 0;
      if (!got) w.av_f.fetch_add(want);
    } else {
      const uint64_t dec = pick == Q_A ? (0ull - ((uint64_t)(uint32_t)want << 32)) : (0ull - (uint64_t)(uint32_t)want);
      const uint64_t old = w.av_da.fetch_add(dec);
      const int had = (int)((int64_t)(pick == Q_A ? (uint32_t)(old >> 32) : (uint32_t)old) - kBias);
      got = had >= want ? want : 0;
      if (!got) w.av_da.fetch_add((uint64_t)(uint32_t)want << (pick == Q_A ? 32 : 0));
    }
    if (!got) continue;
    pos = w.head[pick].fetch_add((uint32_t)got);
    kind = pick;
    count = got;
    return true;
  }
}

void wavefront(Workgroup& w, int id, unsigned seed) {
  std::mt19937 rng(seed);
  Lane lanes[kLanes];
  for (;;) {
    push(w, lanes);
    for (auto& l : lanes) l = Lane();
    int kind, count;
    uint32_t pos;
    if (!pop(w, kind, count, pos, rng)) return;
    if ((rng() & 63) == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 3000));   // a dawdling consumer
    for (int i = 0; i < count; ++i) {   // lf_entry: an entry of another lap has not been written yet
      const uint32_t at = pos + (uint32_t)i;
      uint16_t v;
      const auto t0 = std::chrono::steady_clock::now();
      while (((v = w.q[kind][at & (w.ring - 1)].load(std::memory_order_acquire)) & 0xFC00u) != w.lap(at)) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
          const uint64_t r = w.res_da.load(), a = w.av_da.load();
          std::fprintf(stderr, "stuck: wave %d kind %d pos %u (+%d of %d) entry %04x want gen %04x | RES_D %u RES_A %u RES_F %u HEAD %u %u %u AV %d %d %d remain %u\n",
                       id, kind, at, i, count, (unsigned)v, w.lap(at), (uint32_t)r, (uint32_t)(r >> 32), w.res_f.load(), w.head[0].load(), w.head[1].load(),
                       w.head[2].load(), (int)((int64_t)(uint32_t)a - kBias), (int)((int64_t)(uint32_t)(a >> 32) - kBias), w.av_f.load(), w.remain.load());
          w.fail("a ring entry never arrived");
          return;
        }
        std::this_thread::yield();
      }
      lanes[i].slot = v & 1023;
      w.q[kind][at & (w.ring - 1)].store((uint16_t)(0x400u | w.lap(at)), std::memory_order_release);   // consumed
      int expect = -1;
      if (!w.slot_owner[lanes[i].slot].compare_exchange_strong(expect, id)) {
        w.fail("a slot was handed to two batches at once");
        return;
      }
    }
    if ((rng() & 7) == 0) std::this_thread::yield();
    for (int i = 0; i < count; ++i) {
      Lane& l = lanes[i];
      if (kind == Q_F) {   // claim(): the next element of the range, or the slot dies
        const int rel = (int)w.next.fetch_add(1);
        if (rel >= w.range) {
          l.nxt = NXT_DEAD;
        } else {
          w.elem_claims[rel].fetch_add(1);
          w.slot_elem[l.slot] = rel;
          w.slot_step[l.slot] = 0;
          l.nxt = (rng() & 1) ? NXT_A : NXT_D;
        }
      } else {             // one ladder step; after `steps` of them the element is done and the slot is free
        const int el = w.slot_elem[l.slot];
        w.elem_steps[el].fetch_add(1);
        l.nxt = ++w.slot_step[l.slot] == w.steps ? NXT_FREE : ((rng() & 1) ? NXT_A : NXT_D);
      }
      w.slot_owner[l.slot].store(-1);
    }
  }
}

bool run(int slots, int range, int steps, unsigned seed) {
  Workgroup w(slots, range, steps);
  std::vector<std::thread> t;
  for (int i = 0; i < kWaves; ++i) t.emplace_back(wavefront, std::ref(w), i, seed * 977u + (unsigned)i);
  for (auto& th : t) th.join();
  if (w.failures) return false;
  for (int e = 0; e < range; ++e)
    if (w.elem_claims[e] != 1 || w.elem_steps[e] != steps) {
      std::fprintf(stderr, "FAIL: element %d claimed %d times, %d of %d steps (slots %d, range %d)\n", e, (int)w.elem_claims[e],
                   (int)w.elem_steps[e], steps, slots, range);
      return false;
    }
  const uint64_t av = w.av_da.load();
  if ((uint32_t)av != (uint32_t)kBias || (uint32_t)(av >> 32) != (uint32_t)kBias || w.av_f.load() != 0 || w.remain.load() != 0) {
    std::fprintf(stderr, "FAIL: counters not back at rest (slots %d, range %d)\n", slots, range);
    return false;
  }
  return true;
}

}  // namespace

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 3;
  const int slot_counts[] = {64, 128, 200, 864, 1024};
  const int ranges[] = {0, 1, 63, 64, 65, 700, 864, 865, 3000};
  int runs = 0;
  for (int r = 0; r < rounds; ++r)
    for (int slots : slot_counts)
      for (int range : ranges)
        for (int steps : {1, 2, 17}) {
          if (!run(slots, range, steps, (unsigned)(r * 7919 + slots * 31 + range * 3 + steps))) return 1;
          ++runs;
        }
  std::printf("sched_lf model: %d runs, every element claimed once and stepped to the end, all queues back at rest\n", runs);
  return 0;
}
