// sched_ctl.hpp -- the control block of the workgroup task schedulers (kernels_p256.hip, kernels_ed.hip): queue heads
// and tails, counters and the ticket lock, as LDS words.
//
// They are accessed through an LDS-address-space pointer or with explicit ds_read / ds_write: a volatile access through
// a generic pointer is left as a FLAT access by the compiler (64-bit address, sc0 sc1, a VMEM round trip each), and a
// critical section is a chain of a dozen of them.
#pragma once
#include "limbs.hpp"

namespace fecgpu {
namespace {

// words 0..5 are rewritten in every critical section (one ds_write_b128 + one ds_write_b64); 0..7 are read with two
// ds_read_b128
enum { C_HEAD_D = 0, C_TAIL_D, C_HEAD_A, C_TAIL_A, C_INFLIGHT, C_REMAIN, C_ERR, C_TICKET, C_SERVING, C_WORDS };
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));
// the eight control words in two LDS reads, each broadcast from lane 0 into SGPRs (they are wave-uniform: the queue
// arithmetic that follows then runs on the scalar unit, not as 64-lane VALU instructions)
struct CtlWords {
  int head_d, tail_d, head_a, tail_a, inflight, remain, err;
};
FEC_DEV CtlWords ctl_read(unsigned lds_addr) {
  v4i_t a, b;
  asm volatile("ds_read_b128 %0, %2\n\t"
               "ds_read_b128 %1, %2 offset:16\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(a), "=&v"(b) : "v"(lds_addr) : "memory");
  CtlWords c;
  c.head_d = __builtin_amdgcn_readfirstlane(a.x);
  c.tail_d = __builtin_amdgcn_readfirstlane(a.y);
  c.head_a = __builtin_amdgcn_readfirstlane(a.z);
  c.tail_a = __builtin_amdgcn_readfirstlane(a.w);
  c.inflight = __builtin_amdgcn_readfirstlane(b.x);
  c.remain = __builtin_amdgcn_readfirstlane(b.y);
  c.err = __builtin_amdgcn_readfirstlane(b.z);
  return c;
}
// words 0..5 written back (the caller runs this on lane 0 only)
FEC_DEV void ctl_write(unsigned lds_addr, int head_d, int tail_d, int head_a, int tail_a, int inflight, int remain) {
  v4i_t a;
  a.x = head_d; a.y = tail_d; a.z = head_a; a.w = tail_a;
  v2i_t b;
  b.x = inflight; b.y = remain;
  asm volatile("ds_write_b128 %0, %1\n\t"
               "ds_write_b64 %0, %2 offset:16\n\t"
               "s_waitcnt lgkmcnt(0)"
               : : "v"(lds_addr), "v"(a), "v"(b) : "memory");
}

typedef volatile __attribute__((address_space(3))) int* lds_int_ptr;
// atomic fetch-and-add on control word `w` (ds_add_rtn_u32; atomicAdd() on a generic pointer would be a FLAT atomic)
FEC_DEV int lds_fetch_add(lds_int_ptr ctl, int w, int v) {
  return __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)(ctl + w), v, __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace
}  // namespace fecgpu
