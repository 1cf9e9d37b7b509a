// coop.hpp -- helpers for executing ONE point addition on several lanes of a wavefront (the ordered folds of
// Curve::multi_scalar_multiply, Ecdsa::batch_verify, ...: a strictly sequential chain of additions whose only
// parallelism is inside one addition).  Values travel between lanes through eight-word slots of LDS owned by
// the wavefront; a wavefront's LDS instructions execute in program order, so a read issued after another
// lane's write (same wavefront, later instruction) returns the written data and no s_waitcnt is needed between
// them, only compiler ordering.  secp256k1.hpp carries its own copy of these helpers (namespace secp::coop).
#pragma once
#include "limbs.hpp"

namespace fecgpu {
namespace coopx {

#ifdef FEC_HOST_EMUL
FEC_DEV fe ld(const u32* sh, int slot) {
  fe a;
  FEC_UNROLL for (int i = 0; i < 8; ++i) a.w[i] = sh[slot * 8 + i];
  return a;
}
FEC_DEV void st(u32* sh, int slot, const fe& a) {
  FEC_UNROLL for (int i = 0; i < 8; ++i) sh[slot * 8 + i] = a.w[i];
}
FEC_DEV void sync() {}
FEC_DEV int lane_id() { return 0; }
#else
// a slot is 32 bytes, 16-byte aligned (the caller's array is): two ds_read_b128 / ds_write_b128
FEC_DEV fe ld(const u32* sh, int slot) {
  const uint4* s4 = reinterpret_cast<const uint4*>(sh + slot * 8);
  const uint4 lo = s4[0], hi = s4[1];
  fe a;
  a.w[0] = lo.x; a.w[1] = lo.y; a.w[2] = lo.z; a.w[3] = lo.w;
  a.w[4] = hi.x; a.w[5] = hi.y; a.w[6] = hi.z; a.w[7] = hi.w;
  return a;
}
FEC_DEV void st(u32* sh, int slot, const fe& a) {
  uint4* s4 = reinterpret_cast<uint4*>(sh + slot * 8);
  s4[0] = make_uint4(a.w[0], a.w[1], a.w[2], a.w[3]);
  s4[1] = make_uint4(a.w[4], a.w[5], a.w[6], a.w[7]);
}
FEC_DEV void sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
FEC_DEV int lane_id() { return (int)(threadIdx.x & 63); }
#endif
// slot of lane 0..4, `other` for every further lane
FEC_DEV int pick(int lane, int a0, int a1, int a2, int a3, int a4, int other) {
  return lane == 0 ? a0 : (lane == 1 ? a1 : (lane == 2 ? a2 : (lane == 3 ? a3 : (lane == 4 ? a4 : other))));
}

}  // namespace coopx
}  // namespace fecgpu
