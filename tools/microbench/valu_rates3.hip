// Round-3 microbenchmark: (1) statistical stress of nop-less VCC carry chains (the compiler pads
// every VCC write->read with s_nop 1 on gfx950; is that needed for v_addc/v_subb chains?),
// (2) v_cndmask encodings, (3) candidate select idioms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef uint32_t u32; typedef uint64_t u64;

__device__ __forceinline__ u32 xs(u32& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// 8-word add and sub through VCC, no s_nop, one asm statement each
__device__ __forceinline__ u32 asm_add256(u32 r[8], const u32 a[8], const u32 b[8]) {
  u32 c;
  asm("v_add_co_u32 %0, vcc, %9, %17\n v_addc_co_u32 %1, vcc, %10, %18, vcc\n v_addc_co_u32 %2, vcc, %11, %19, vcc\n"
      "v_addc_co_u32 %3, vcc, %12, %20, vcc\n v_addc_co_u32 %4, vcc, %13, %21, vcc\n v_addc_co_u32 %5, vcc, %14, %22, vcc\n"
      "v_addc_co_u32 %6, vcc, %15, %23, vcc\n v_addc_co_u32 %7, vcc, %16, %24, vcc\n v_addc_co_u32 %8, vcc, 0, 0, vcc\n"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(c)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
        "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
  return c;
}
__device__ __forceinline__ u32 asm_sub256(u32 r[8], const u32 a[8], const u32 b[8]) {
  u32 c;
  asm("v_sub_co_u32 %0, vcc, %9, %17\n v_subb_co_u32 %1, vcc, %10, %18, vcc\n v_subb_co_u32 %2, vcc, %11, %19, vcc\n"
      "v_subb_co_u32 %3, vcc, %12, %20, vcc\n v_subb_co_u32 %4, vcc, %13, %21, vcc\n v_subb_co_u32 %5, vcc, %14, %22, vcc\n"
      "v_subb_co_u32 %6, vcc, %15, %23, vcc\n v_subb_co_u32 %7, vcc, %16, %24, vcc\n v_addc_co_u32 %8, vcc, 0, 0, vcc\n"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(c)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
        "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
  return c;
}
// reference with 64-bit arithmetic the compiler lowers itself (no shared code with the asm)
__device__ __forceinline__ u32 ref_add256(u32 r[8], const u32 a[8], const u32 b[8]) {
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { u64 s = (u64)a[i] + b[i] + c; r[i] = (u32)s; c = s >> 32; }
  return (u32)c;
}
__device__ __forceinline__ u32 ref_sub256(u32 r[8], const u32 a[8], const u32 b[8]) {
  u64 bo = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { u64 d = (u64)a[i] - b[i] - bo; r[i] = (u32)d; bo = (d >> 32) & 1; }
  return (u32)bo;
}
__global__ __launch_bounds__(256) void k_stress(unsigned long long* mism, int iters, u32 seed) {
  u32 s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + seed; if (!s) s = 1;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    u32 a[8], b[8], r1[8], r2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      u32 x = xs(s), y = xs(s), m = xs(s);
      a[i] = (m & 3) == 0 ? 0xFFFFFFFFu : ((m & 12) == 0 ? 0u : x);   // many all-ones / zero words: long ripples
      b[i] = (m & 48) == 0 ? 0u : ((m & 192) == 0 ? 0xFFFFFFFFu : ((m & 768) == 0 ? 1u : y));
    }
    u32 c1 = asm_add256(r1, a, b), c2 = ref_add256(r2, a, b);
    u32 d = c1 ^ c2;
#pragma unroll
    for (int i = 0; i < 8; ++i) d |= r1[i] ^ r2[i];
    c1 = asm_sub256(r1, a, b); c2 = ref_sub256(r2, a, b);
    d |= c1 ^ c2;
#pragma unroll
    for (int i = 0; i < 8; ++i) d |= r1[i] ^ r2[i];
    bad += d != 0;
  }
  if (bad) atomicAdd(mism, (unsigned long long)bad);
}

#define REP8(s) s s s s s s s s
#define ITERS 2048
#define SCLOB "vcc","s10","s11","s12","s13","s14","s15","s16","s17","s18","s19","s20","s21","s22","s23","s24","s25"
#define K32(name, BODY) \
__global__ __launch_bounds__(256) void name(u32* out, u32 seed) { \
  u32 a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u; \
  u32 c0 = a, c1 = b, c2 = a + 1, c3 = b + 2, c4 = a + 3, c5 = b + 4, c6 = a + 5, c7 = b + 6; \
  for (int it = 0; it < ITERS; ++it) { \
    asm volatile(REP8(BODY) \
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) \
      : "v"(a), "v"(b) : SCLOB); } \
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7; }

K32(k_cnd_e32_vcc, "v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n"
                   "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n")
K32(k_cnd_e64_vcc, "v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n"
                   "v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc\n")
K32(k_cnd_e64_s, "v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n"
                 "v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]\n")
// e32 cndmask where src differs from dst (no RAW on the destination chain)
K32(k_cnd_e32_vcc_nodep, "v_cndmask_b32_e32 %0, %8, %9, vcc\n v_cndmask_b32_e32 %1, %8, %9, vcc\n v_cndmask_b32_e32 %2, %8, %9, vcc\n v_cndmask_b32_e32 %3, %8, %9, vcc\n"
                   "v_cndmask_b32_e32 %4, %8, %9, vcc\n v_cndmask_b32_e32 %5, %8, %9, vcc\n v_cndmask_b32_e32 %6, %8, %9, vcc\n v_cndmask_b32_e32 %7, %8, %9, vcc\n")
K32(k_cnd_const01_vcc, "v_cndmask_b32_e64 %0, 0, 1, vcc\n v_cndmask_b32_e64 %1, 0, 1, vcc\n v_cndmask_b32_e64 %2, 0, 1, vcc\n v_cndmask_b32_e64 %3, 0, 1, vcc\n"
                   "v_cndmask_b32_e64 %4, 0, 1, vcc\n v_cndmask_b32_e64 %5, 0, 1, vcc\n v_cndmask_b32_e64 %6, 0, 1, vcc\n v_cndmask_b32_e64 %7, 0, 1, vcc\n")
K32(k_and_or, "v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n"
              "v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9\n")
K32(k_subb, "v_sub_co_u32 %0, vcc, %0, %8\n v_subb_co_u32 %1, vcc, %1, %9, vcc\n v_subb_co_u32 %2, vcc, %2, %8, vcc\n v_subb_co_u32 %3, vcc, %3, %9, vcc\n"
            "v_subb_co_u32 %4, vcc, %4, %8, vcc\n v_subb_co_u32 %5, vcc, %5, %9, vcc\n v_subb_co_u32 %6, vcc, %6, %8, vcc\n v_subb_co_u32 %7, vcc, %7, %9, vcc\n")
// mad chain as the schoolbook row would issue it: mad (sdst ignored) then addc on vcc
K32(k_mul_hi_lo_pair, "v_mul_lo_u32 %0, %8, %9\n v_mul_hi_u32 %1, %8, %9\n v_mul_lo_u32 %2, %8, %9\n v_mul_hi_u32 %3, %8, %9\n"
                      "v_mul_lo_u32 %4, %8, %9\n v_mul_hi_u32 %5, %8, %9\n v_mul_lo_u32 %6, %8, %9\n v_mul_hi_u32 %7, %8, %9\n")

typedef void (*kern_t)(u32*, u32);
struct Case { const char* name; kern_t k; int n; };
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate / 1e6;
  unsigned long long* d_m; CK(hipMalloc(&d_m, 8));
  for (int wps : {1, 2, 4, 8}) {
    CK(hipMemset(d_m, 0, 8));
    int iters = 20000;
    k_stress<<<cus * wps, 256>>>(d_m, iters, 12345u + wps); CK(hipDeviceSynchronize());
    unsigned long long m; CK(hipMemcpy(&m, d_m, 8, hipMemcpyDeviceToHost));
    printf("carry-chain stress, %d waves/SIMD: %llu mismatches in %.3e add+sub chain pairs\n", wps, m, (double)cus * wps * 256 * iters);
  }
  u32* out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  std::vector<Case> cases = {{"v_cndmask_e32 vcc (dst=src0)", k_cnd_e32_vcc, 64}, {"v_cndmask_e32 vcc (no dep)", k_cnd_e32_vcc_nodep, 64},
    {"v_cndmask_e64 vcc", k_cnd_e64_vcc, 64}, {"v_cndmask_e64 s[10:11]", k_cnd_e64_s, 64}, {"v_cndmask_e64 0,1,vcc", k_cnd_const01_vcc, 64},
    {"v_and_or_b32", k_and_or, 64}, {"v_sub_co/v_subb chain (no nop)", k_subb, 64}, {"v_mul_lo + v_mul_hi", k_mul_hi_lo_pair, 64}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-34s %6s %10s %14s\n", "instruction", "w/SIMD", "ms", "cyc/wave-inst");
  for (auto& c : cases) for (int wps : {1, 2, 4}) {
    int blocks = cus * wps;
    c.k<<<blocks, 256>>>(out, 1); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); c.k<<<blocks, 256>>>(out, 2); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %6d %10.4f %14.3f\n", c.name, wps, ms, ms * 1e-3 * clk * 1e9 / ((double)wps * ITERS * c.n));
  }
  return 0;
}
