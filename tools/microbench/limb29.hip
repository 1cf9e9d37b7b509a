// 29-bit limbs against 32-bit limbs for the 256 x 256 -> 512-bit product core.
//
// With 32-bit limbs every partial product costs a v_mad_u64_u32 AND a v_addc that catches its carry (a column sum
// of eight 64-bit products needs 67 bits).  With nine limbs of 29 bits a product is below 2^58 and a column of nine
// stays below 2^62: the 81 products accumulate into plain 64-bit column registers with NO carry capture.  What that
// buys and what it costs is timed here in three forms, operands already in limb form:
//   k_core29      81 v_mad_u64_u32 into 17 64-bit columns, columns left unnormalised (the bare lower bound);
//   k_core29_norm the same plus the carry propagation that turns the columns back into 29-bit limbs (what an
//                 implementation that keeps field elements in this radix between operations pays per product);
//   k_mul_wide    the shipped 32-bit product scanning (64 mads, 64 captures, 16 moves), for reference.
// The parity path would additionally need every quirk of the reference re-derived for the radix (not timed).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../forge_ec_amd/csrc -o limb29 limb29.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "secp256k1.hpp"
using namespace fecgpu;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITERS = 2000;
constexpr u32 M29 = (1u << 29) - 1;

template <bool NORM>
__global__ __launch_bounds__(256) void k_core29(const u32* in, u64* out) {
  u32 a[9], b[9];
  for (int i = 0; i < 9; ++i) { a[i] = (in[threadIdx.x * 18 + i] ^ blockIdx.x) & M29; b[i] = in[threadIdx.x * 18 + 9 + i] & M29; }
  u64 acc = 0;
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
    u64 col[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) col[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
      for (int j = 0; j < 9; ++j) col[i + j] += (u64)a[i] * b[j];   // v_mad_u64_u32, no carry out of 64 bits
    }
    if (NORM) {
      u64 carry = 0;
      u32 limb[18];
#pragma unroll
      for (int k = 0; k < 17; ++k) {
        const u64 c = col[k] + carry;
        limb[k] = (u32)c & M29;
        carry = c >> 29;
      }
      limb[17] = (u32)carry;
#pragma unroll
      for (int i = 0; i < 9; ++i) a[i] = (limb[i] ^ limb[i + 9]) & M29;   // keep the chain dependent
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i) a[i] = ((u32)col[i] ^ (u32)(col[i + 8] >> 3)) & M29;
    }
    acc ^= col[16];
  }
  u64 s = acc;
  for (int i = 0; i < 9; ++i) s ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_mul_wide(const u32* in, u32* out) {
  fe a, b;
  for (int i = 0; i < 8; ++i) { a.w[i] = in[threadIdx.x * 18 + i] ^ blockIdx.x; b.w[i] = in[threadIdx.x * 18 + 9 + i]; }
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) { u32 t[16]; mul_wide(t, a, b); for (int i = 0; i < 8; ++i) a.w[i] = t[i] ^ t[i + 8]; }
  for (int i = 0; i < 8; ++i) out[(blockIdx.x * 256 + threadIdx.x) * 8 + i] = a.w[i];
}
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); int cus = prop.multiProcessorCount;
  static u32 hu[256 * 18]; for (int i = 0; i < 256 * 18; ++i) hu[i] = (u32)rand() * 2654435761u;
  u32* uin; void* dout;
  CK(hipMalloc(&uin, sizeof(hu))); CK(hipMalloc(&dout, (size_t)cus * 3 * 256 * 64));
  CK(hipMemcpy(uin, hu, sizeof(hu), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wps : {1, 2, 3}) {
    float ms[3];
    for (int which = 0; which < 3; ++which) for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (which == 0) k_core29<false><<<cus * wps, 256>>>(uin, (u64*)dout);
      else if (which == 1) k_core29<true><<<cus * wps, 256>>>(uin, (u64*)dout);
      else k_mul_wide<<<cus * wps, 256>>>(uin, (u32*)dout);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[which], e0, e1));
    }
    printf("%d waves/SIMD: 9 x 9 limbs of 29 bits, 81 mads, columns unnormalised %.1f SIMD-ns per product | with the carry "
           "propagation back to 29-bit limbs %.1f | 8 x 8 words of 32 bits, mul_wide (64 mads + 64 captures) %.1f\n",
           wps, ms[0] * 1e6 / ((double)ITERS * wps), ms[1] * 1e6 / ((double)ITERS * wps), ms[2] * 1e6 / ((double)ITERS * wps));
  }
  return 0;
}
