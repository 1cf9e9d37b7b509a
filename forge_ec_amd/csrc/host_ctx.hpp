// host_ctx.hpp -- the fec_ctx object and the host-side helpers shared by the translation units of
// libfecgpu.so (fecgpu.hip: parity path; canon.hip: canonical-math mode).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/fecgpu.h"
#include "kernels.hpp"
#include "staging.hpp"

struct fec_ctx {
  using u32 = fecgpu::u32;
  using u64 = fecgpu::u64;
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timing = false, timed = false;
  const char* last_kernel = "";
  // device staging for the host-pointer entry points: slots 0-3 serve pipeline lane 0 (and the
  // small one-shot calls), slots 4-7 pipeline lane 1
  void* d_buf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t d_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hipStream_t stream2 = nullptr;
  size_t chunk = (size_t)1 << 18;  // elements per pipeline chunk (see host::pipeline_chunk)
  u64* d_gen[3] = {nullptr, nullptr, nullptr};  // reference generator() per curve, device copy
  u32* d_ed_table = nullptr;                    // Ed25519 fixed-base addend table (256 x 32 words)
  u64 ed_table_base[16] = {0};                  // the base point the table was built for
  bool ed_table_valid = false;
  u64 h_gen_ed[16] = {0};                       // host copy of the Ed25519 generator (table cache key)
  u64 h_gen[3][16] = {{0}, {0}, {0}};           // host copies of the three generators (fec_batch_mul_fixed recognises them by value)
  // canonical-math mode: comb table of affine multiples of G, per-element window-table scratch
  u32* d_canon_comb[3] = {nullptr, nullptr, nullptr};   // per curve
  bool canon_comb_ready[3] = {false, false, false};
  u32* d_canon_comb8[3] = {nullptr, nullptr, nullptr};   // 8-bit comb (~512 KiB, L2-resident) per curve
  bool canon_comb8_ready[3] = {false, false, false};
  bool canon_use_comb8 = true;                  // FEC_CANON_COMB4=1 selects the 4-bit LDS comb instead
  void* d_win_scratch = nullptr;
  size_t win_scratch_cap = 0;
  void* d_zbuf = nullptr;  // Jacobian Z of the batch between the ladder and the batched normalisation
  size_t zbuf_cap = 0;
  void* d_verify = nullptr;  // canonical ECDSA verification work area (u1, u2, R, flags)
  size_t verify_cap = 0;
  void* d_tbuf = nullptr;  // Ed25519 double-mul: T of the comb result until the accumulate pass
  size_t tbuf_cap = 0;
  hipDeviceProp_t prop;
  // multi-device ctx (fec_ctx_create_multi): the shard workers; empty for a single-device ctx
  std::vector<fec_ctx*> children;
  // per-stream scratch of the composed launches (two ladders + one addition): a buffer is only ever
  // used by launches on the stream it belongs to, so calls on different streams cannot race on it
  struct StreamScratch {
    hipStream_t stream;
    void* buf;
    size_t cap;
  };
  std::vector<StreamScratch> stream_scratch;
  // ordering of launches that share ctx-owned scratch (Ed25519 addend table, canonical-mode work areas,
  // staging): a launch on a stream other than the previous launch's stream first waits for that one
  hipStream_t last_stream = nullptr;
  hipEvent_t ev_order = nullptr;
  // Device error word (kernels.hpp: SchedEnv): one word of pinned host memory mapped into the device.  A scheduler
  // kernel whose watchdog / index guard fires stores a FEC_DEVERR_* code here; the host reads it after synchronising
  // (sync_and_check, fec_ctx_check).  Sticky until read.
  unsigned* h_err = nullptr;       // host view
  unsigned* d_err = nullptr;       // device view of the same word
  unsigned debug_force_fault = 0;  // fec_ctx_debug_force_fault
  // Fixed-base prefix tables of the reference's generator() (kernels.hpp: SchedEnv; fecgpu.hip: ensure_gen_prefix):
  // built by the fixed-base launch that takes the ctx past prefix_after multiplications, 2^prefix_bits entries; 0 = off.
  u32* d_gen_prefix[3] = {nullptr, nullptr, nullptr};
  unsigned gen_prefix_bits[3] = {0, 0, 0};   // bits of the table that exists (0 = none yet / allocation refused)
  bool gen_prefix_tried[3] = {false, false, false};
  unsigned prefix_bits = 0;                  // wanted (FEC_FIXED_PREFIX_BITS at ctx creation, fec_ctx_set_fixed_prefix_bits)
  size_t fixed_elems[3] = {0, 0, 0};         // multiplications by the generator this ctx has been asked for, per curve
  size_t prefix_after = 0;                   // a table is built once fixed_elems reaches this (see fecgpu.hip: kPrefixAfter)
  bool in_multi_chunk_pipeline = false;  // set by host_pipeline while it runs more than one chunk (fecgpu.hip: SideStream)
  bool in_host_call = false;             // inside a host-pointer (synchronous) entry point: host::drained
  bool prefix_explicit = false;          // the caller asked for prefix tables (fec_ctx_set_fixed_prefix_bits): any launch may build one
  unsigned prefix_budget_pct = 25;       // a table and its build scratch may take this share of the device's FREE memory
  size_t side_stream_max = (size_t)-1;   // u1*G runs beside u2*Q on the second stream up to this many elements (fec_ctx_set_side_stream_max)
  hipStream_t stream_gather = nullptr;   // multi-device ctx: the peer copies of a shard's results (fec_multi_batch_*_dev)
  hipEvent_t ev_gather = nullptr;
};

// No exception may cross the C ABI: every extern "C" definition in fecgpu.hip and canon.hip is a function-try-block
// closed by one of these (tests/test_abi_library.py checks that none is missing).
#define FEC_ABI_CATCH_STATUS            \
  catch (const std::bad_alloc&) {       \
    return FEC_E_OOM;                   \
  }                                     \
  catch (...) {                         \
    return FEC_E_DEVICE;                \
  }
#define FEC_ABI_CATCH_VOID catch (...) {}
#define FEC_ABI_CATCH_NULL \
  catch (...) {            \
    return nullptr;        \
  }

// a multi-device ctx runs everything that is not sharded on its first shard worker
#define FEC_FIRST_DEVICE(ctx)                                      \
  do {                                                             \
    if ((ctx) && !(ctx)->children.empty()) (ctx) = (ctx)->children[0]; \
  } while (0)

namespace fecgpu {
namespace host {

inline bool curve_ok(int c) { return c == FEC_SECP256K1 || c == FEC_P256 || c == FEC_ED25519; }
inline int plimbs(int c) { return c == FEC_ED25519 ? 16 : 12; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int ensure(fec_ctx* ctx, int slot, size_t bytes) {
  if (ctx->d_cap[slot] >= bytes) return FEC_OK;
  if (ctx->d_buf[slot]) {
    // A staging buffer may hold a caller's keys: it is cleared before it goes back to the allocator.  Slots 4..7 belong to
    // the pipeline's second lane (stream2), so the clear is not queued on one stream behind whatever the other still has in
    // flight: the device is drained first (growth is rare: once per ctx and size), then the memory is cleared synchronously.
    const bool drained_ok = hipDeviceSynchronize() == hipSuccess;
    const bool cleared = hipMemset(ctx->d_buf[slot], 0, ctx->d_cap[slot]) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    (void)hipFree(ctx->d_buf[slot]);
    if (!drained_ok || !cleared) {   // the old contents could not be cleared: report it rather than carry on silently
      (void)hipGetLastError();
      ctx->d_buf[slot] = nullptr;
      ctx->d_cap[slot] = 0;
      return FEC_E_DEVICE;
    }
  }
  ctx->d_buf[slot] = nullptr;
  ctx->d_cap[slot] = 0;
  size_t cap = bytes + (bytes >> 2) + 4096;
  if (hipMalloc(&ctx->d_buf[slot], cap) != hipSuccess) {
    (void)hipGetLastError();
    return FEC_E_OOM;
  }
  ctx->d_cap[slot] = cap;
  return FEC_OK;
}

// Device scratch of at least `bytes` dedicated to `stream` (grown on demand; growing waits for that stream).
inline void* scratch_for(fec_ctx* ctx, hipStream_t stream, size_t bytes) {
  for (auto& e : ctx->stream_scratch) {
    if (e.stream != stream) continue;
    if (e.cap >= bytes) return e.buf;
    (void)hipMemsetAsync(e.buf, 0, e.cap, stream);  // cleared before it is released (it may hold u1/u2, shared points)
    (void)hipStreamSynchronize(stream);
    (void)hipFree(e.buf);
    e.buf = nullptr;
    e.cap = 0;
    if (hipMalloc(&e.buf, bytes + (bytes >> 2)) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    e.cap = bytes + (bytes >> 2);
    return e.buf;
  }
  fec_ctx::StreamScratch e{stream, nullptr, 0};
  if (hipMalloc(&e.buf, bytes + (bytes >> 2)) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  e.cap = bytes + (bytes >> 2);
  try {
    ctx->stream_scratch.push_back(e);
  } catch (...) {  // no exception may cross the C ABI
    (void)hipFree(e.buf);
    return nullptr;
  }
  return e.buf;
}

// Calls on one ctx are serialised by the caller on the HOST, but they may name different streams while
// sharing ctx-owned device scratch.  Every launch therefore orders itself after the previous launch of
// the ctx when that one went to a different stream (event record + stream wait: no host blocking).
inline void order_after_previous(fec_ctx* ctx, hipStream_t s) {
  if (ctx->last_stream && ctx->last_stream != s) {
    if (!ctx->ev_order) (void)hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming);
    if (ctx->ev_order) {
      (void)hipEventRecord(ctx->ev_order, ctx->last_stream);
      (void)hipStreamWaitEvent(s, ctx->ev_order, 0);
    }
  }
  ctx->last_stream = s;
}

struct Launch {
  fec_ctx* ctx;
  hipStream_t s;
  Launch(fec_ctx* c, void* stream, const char* name) : ctx(c), s(stream ? (hipStream_t)stream : c->stream) {
    order_after_previous(ctx, s);
    ctx->last_kernel = name;
    ctx->timed = false;
    if (ctx->timing) (void)hipEventRecord(ctx->ev0, s);
  }
  int done() {
    hipError_t e = hipGetLastError();
    if (ctx->timing) {
      (void)hipEventRecord(ctx->ev1, s);
      ctx->timed = true;
    }
    return e == hipSuccess ? FEC_OK : FEC_E_LAUNCH;
  }
};

inline unsigned grid_for(size_t n) { return (unsigned)((n + TPB - 1) / TPB); }

inline SchedEnv sched_env(const fec_ctx* ctx) {
  SchedEnv e;
  e.err = ctx->d_err;
  e.cus = ctx->prop.multiProcessorCount > 0 ? (unsigned)ctx->prop.multiProcessorCount : 256u;
  e.force_fault = ctx->debug_force_fault;
  for (int c = 0; c < 3; ++c) {
    e.gen[c] = reinterpret_cast<const u32*>(ctx->d_gen[c]);
    e.gen_prefix[c] = ctx->d_gen_prefix[c];
    e.gen_prefix_bits[c] = ctx->d_gen_prefix[c] ? ctx->gen_prefix_bits[c] : 0;
  }
  return e;
}

// Reads and clears the ctx's device error word.  Only meaningful once the launches in question have completed
// (the callers synchronise first).  FEC_E_LAUNCH when a kernel reported a fault: its outputs must not be used.
inline int take_device_error(fec_ctx* ctx) {
  if (!ctx->h_err) return FEC_OK;
  const unsigned code = *reinterpret_cast<volatile unsigned*>(ctx->h_err);
  if (code == 0) return FEC_OK;
  *reinterpret_cast<volatile unsigned*>(ctx->h_err) = 0;
  return FEC_E_LAUNCH;
}

// The end of every host-pointer entry point: wait for the stream(s), turn a HIP failure into FEC_E_LAUNCH, then
// look at the device error word.
inline int sync_and_check(fec_ctx* ctx, hipStream_t a, hipStream_t b = nullptr) {
  bool ok = hipStreamSynchronize(a) == hipSuccess;
  if (b) ok = (hipStreamSynchronize(b) == hipSuccess) && ok;
  if (!ok) {
    (void)hipGetLastError();
    (void)take_device_error(ctx);
    return FEC_E_LAUNCH;
  }
  return take_device_error(ctx);
}

// Elements per chunk of a host-pointer call: 2^18 unless fec_ctx_set_chunk says otherwise.  (A chunk gives a persistent
// scheduler workgroup 1 024 elements: the launchers then take the 1 024-slot instantiation of the kernel, one fill --
// kernels_p256.hip: wide_slots_pay.  With 832 slots only (round 3), such a chunk cost 2^20 P-256 multiplications through the
// host-pointer entry point 34.2 ms instead of 26.0: tools/host_chunk_probe.py, profiles/host_chunk_r03.txt.)
inline size_t pipeline_chunk(const fec_ctx* ctx) { return ctx->chunk; }

// The body of a host-pointer entry point whose work is queued on the ctx's own streams, and its way out.  A call that
// fails half-way (an allocation, a refused copy, a launch error) may still have copies from or into the caller's arrays
// queued, and the caller is free to release those arrays as soon as the call is back: the streams are drained first, and
// a device error word raised by the abandoned launches is dropped with them instead of being reported by the next,
// unrelated call.  While the body runs the ctx knows that it is inside a SYNCHRONOUS call (in_host_call): only there may
// a launch allocate and build a fixed-base prefix table by itself (fecgpu.hip: ensure_gen_prefix) -- the *_dev entry
// points only enqueue.
template <class F>
inline int drained(fec_ctx* ctx, F body) {
  struct InHostCall {
    fec_ctx* c;
    bool was;
    explicit InHostCall(fec_ctx* c_) : c(c_), was(c_->in_host_call) { c->in_host_call = true; }
    ~InHostCall() { c->in_host_call = was; }
  } guard(ctx);
  const int rc = body();
  if (rc == FEC_OK) return rc;
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
  (void)hipGetLastError();
  (void)take_device_error(ctx);
  return rc;
}

// Host-pointer call over chunks of ctx->chunk elements with up to four per-element inputs (slots 0-3)
// and two per-element outputs (slots 4-5): H2D, body(d_in[4], d_out[2], count), D2H per chunk on the
// ctx stream.  Device staging and any per-element scratch the body allocates stay bounded by one
// chunk however large n is.  A null input / output pointer is passed through as null.
template <class F>
inline int host_chunked(fec_ctx* ctx, size_t n, const void* const in[4], const size_t in_stride[4], void* const out[2],
                        const size_t out_stride[2], F body) {
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  const size_t pc = pipeline_chunk(ctx);
  const size_t chunk = pc < n ? pc : n;
  return drained(ctx, [&]() -> int {
  for (size_t lo = 0; lo < n; lo += chunk) {
    const size_t cnt = lo + chunk <= n ? chunk : n - lo;
    void* d_in[4] = {nullptr, nullptr, nullptr, nullptr};
    void* d_out[2] = {nullptr, nullptr};
    for (int i = 0; i < 4; ++i) {
      if (!in[i]) continue;
      int rc = ensure(ctx, i, chunk * in_stride[i]);
      if (rc != FEC_OK) return rc;
      d_in[i] = ctx->d_buf[i];
      if (hipMemcpyAsync(d_in[i], (const char*)in[i] + lo * in_stride[i], cnt * in_stride[i], hipMemcpyHostToDevice,
                         ctx->stream) != hipSuccess)
        return FEC_E_DEVICE;
    }
    for (int i = 0; i < 2; ++i) {
      if (!out[i]) continue;
      int rc = ensure(ctx, 4 + i, chunk * out_stride[i]);
      if (rc != FEC_OK) return rc;
      d_out[i] = ctx->d_buf[4 + i];
    }
    int rc = body(d_in, d_out, cnt);
    if (rc != FEC_OK) return rc;
    for (int i = 0; i < 2; ++i) {
      if (!out[i]) continue;
      if (hipMemcpyAsync((char*)out[i] + lo * out_stride[i], d_out[i], cnt * out_stride[i], hipMemcpyDeviceToHost,
                         ctx->stream) != hipSuccess)
        return FEC_E_DEVICE;
    }
    rc = sync_and_check(ctx, ctx->stream);
    if (rc != FEC_OK) return rc;
  }
  return FEC_OK;
  });
}

}  // namespace host
}  // namespace fecgpu
