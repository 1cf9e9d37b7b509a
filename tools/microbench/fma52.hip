// 52-bit-limb v_fma_f64 product core against the 32-bit-limb v_mad_u64_u32 product scanning (VERDICT r1,
// item 2 iv).  256 x 256 -> 512 bits as 5 x 5 limbs of 52 bits held in doubles: each partial product
// needs TWO v_fma_f64 (high part, then the exact low part) and two 64-bit integer accumulations of their
// bit patterns into column sums (Emmart / Luo / Weems "Faster modular exponentiation using double
// precision floating point arithmetic on the GPU").  This kernel times ONLY that core -- 50 fma + 50
// 64-bit adds, operands already in double form, column sums left unnormalised -- i.e. a LOWER bound
// for the method: the parity path would also have to convert both operands from 8 x 32-bit words
// (about 60 instructions) and normalise/repack ten 64-bit column sums into sixteen 32-bit words
// (about 60 more) around every product, because the reference's reductions are defined on 64-bit limbs.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../forge_ec_amd/csrc -o fma52 fma52.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "secp256k1.hpp"
using namespace fecgpu;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITERS = 2000;

__global__ __launch_bounds__(256) void k_fma52(const double* in, long long* out) {
  double a[5], b[5];
  for (int i = 0; i < 5; ++i) { a[i] = in[threadIdx.x * 10 + i] + blockIdx.x; b[i] = in[threadIdx.x * 10 + 5 + i]; }
  long long col[10];
  for (int k = 0; k < 10; ++k) col[k] = 0;
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;   // the method's magic constants (round-toward-zero mode)
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        double hi = __builtin_fma(a[i], b[j], C1);
        double lo = __builtin_fma(a[i], b[j], C2 - hi);
        col[i + j + 1] += __builtin_bit_cast(long long, hi);
        col[i + j] += __builtin_bit_cast(long long, lo);
      }
    }
    a[0] = __builtin_bit_cast(double, (col[0] & 0x000FFFFFFFFFFFFFll) | 0x4330000000000000ll);  // keep the chain dependent
  }
  long long s = 0;
  for (int k = 0; k < 10; ++k) s ^= col[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_mul_wide(const u32* in, u32* out) {
  fe a, b;
  for (int i = 0; i < 8; ++i) { a.w[i] = in[threadIdx.x * 16 + i] ^ blockIdx.x; b.w[i] = in[threadIdx.x * 16 + 8 + i]; }
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) { u32 t[16]; mul_wide(t, a, b); for (int i = 0; i < 8; ++i) a.w[i] = t[i] ^ t[i + 8]; }
  for (int i = 0; i < 8; ++i) out[(blockIdx.x * 256 + threadIdx.x) * 8 + i] = a.w[i];
}
__global__ __launch_bounds__(256) void k_secp_mul(const u32* in, u32* out) {
  fe a, b;
  for (int i = 0; i < 8; ++i) { a.w[i] = in[threadIdx.x * 16 + i] ^ blockIdx.x; b.w[i] = in[threadIdx.x * 16 + 8 + i]; }
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) a = secp::mul(a, b);
  for (int i = 0; i < 8; ++i) out[(blockIdx.x * 256 + threadIdx.x) * 8 + i] = a.w[i];
}
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); int cus = prop.multiProcessorCount;
  double hd[2560]; for (int i = 0; i < 2560; ++i) hd[i] = (double)(((unsigned long long)rand() << 21) ^ rand());
  u32 hu[4096]; for (int i = 0; i < 4096; ++i) hu[i] = (u32)rand() * 2654435761u;
  double* din; u32* uin; void* dout;
  CK(hipMalloc(&din, sizeof(hd))); CK(hipMalloc(&uin, sizeof(hu))); CK(hipMalloc(&dout, (size_t)cus * 3 * 256 * 64));
  CK(hipMemcpy(din, hd, sizeof(hd), hipMemcpyHostToDevice)); CK(hipMemcpy(uin, hu, sizeof(hu), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wps : {1, 2, 3}) {
    float ms[3];
    for (int which = 0; which < 3; ++which) for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (which == 0) k_fma52<<<cus * wps, 256>>>(din, (long long*)dout);
      else if (which == 1) k_mul_wide<<<cus * wps, 256>>>(uin, (u32*)dout);
      else k_secp_mul<<<cus * wps, 256>>>(uin, (u32*)dout);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[which], e0, e1));
    }
    printf("%d waves/SIMD: fma52 core (50 v_fma_f64 + 50 64-bit adds, no conversions) %.1f SIMD-ns per product | "
           "mul_wide, compiler-scheduled v_mad_u64_u32 form %.1f | whole secp256k1 Mul, asm (product + Montgomery) %.1f\n",
           wps, ms[0] * 1e6 / ((double)ITERS * wps), ms[1] * 1e6 / ((double)ITERS * wps), ms[2] * 1e6 / ((double)ITERS * wps));
  }
  return 0;
}
