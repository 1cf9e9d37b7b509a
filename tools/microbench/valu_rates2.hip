// Second-round integer-VALU microbenchmark for gfx950: separates true issue cost
// from VCC/SGPR carry dependencies (round-1 chains all shared VCC), and checks
// whether the VALU-writes-SGPR -> VALU-reads-SGPR hazard is interlocked.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates2 valu_rates2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
#define REP8(s) s s s s s s s s
#define ITERS 2048
#define SCLOB "vcc","s10","s11","s12","s13","s14","s15","s16","s17","s18","s19","s20","s21","s22","s23","s24","s25"

#define K64(name, BODY) \
__global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t seed) { \
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u; \
  uint64_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 2, c4 = a + 3, c5 = b + 4, c6 = a + 5, c7 = b + 6; \
  for (int it = 0; it < ITERS; ++it) { \
    asm volatile(REP8(BODY) \
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) \
      : "v"(a), "v"(b) : SCLOB); } \
  uint64_t cs = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7; \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)cs ^ (uint32_t)(cs >> 32); }

#define K32(name, BODY) \
__global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t seed) { \
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u; \
  uint32_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 2, c4 = a + 3, c5 = b + 4, c6 = a + 5, c7 = b + 6; \
  for (int it = 0; it < ITERS; ++it) { \
    asm volatile(REP8(BODY) \
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) \
      : "v"(a), "v"(b) : SCLOB); } \
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7; }

// mad_u64_u32 with 8 distinct carry-out SGPR pairs (no WAW on VCC)
K64(k_mad64_sdst8,
  "v_mad_u64_u32 %0, s[10:11], %8, %9, %0\n" "v_mad_u64_u32 %1, s[12:13], %8, %9, %1\n"
  "v_mad_u64_u32 %2, s[14:15], %8, %9, %2\n" "v_mad_u64_u32 %3, s[16:17], %8, %9, %3\n"
  "v_mad_u64_u32 %4, s[18:19], %8, %9, %4\n" "v_mad_u64_u32 %5, s[20:21], %8, %9, %5\n"
  "v_mad_u64_u32 %6, s[22:23], %8, %9, %6\n" "v_mad_u64_u32 %7, s[24:25], %8, %9, %7\n")
// mad with inline-constant 0 addend (pure 32x32->64 multiply)
K64(k_mad64_c0,
  "v_mad_u64_u32 %0, s[10:11], %8, %9, 0\n" "v_mad_u64_u32 %1, s[12:13], %8, %9, 0\n"
  "v_mad_u64_u32 %2, s[14:15], %8, %9, 0\n" "v_mad_u64_u32 %3, s[16:17], %8, %9, 0\n"
  "v_mad_u64_u32 %4, s[18:19], %8, %9, 0\n" "v_mad_u64_u32 %5, s[20:21], %8, %9, 0\n"
  "v_mad_u64_u32 %6, s[22:23], %8, %9, 0\n" "v_mad_u64_u32 %7, s[24:25], %8, %9, 0\n")
K64(k_lshl_add_u64,
  "v_lshl_add_u64 %0, %0, 0, %1\n" "v_lshl_add_u64 %1, %1, 0, %2\n" "v_lshl_add_u64 %2, %2, 0, %3\n"
  "v_lshl_add_u64 %3, %3, 0, %4\n" "v_lshl_add_u64 %4, %4, 0, %5\n" "v_lshl_add_u64 %5, %5, 0, %6\n"
  "v_lshl_add_u64 %6, %6, 0, %7\n" "v_lshl_add_u64 %7, %7, 0, %0\n")
// add_co with 8 distinct carry-out pairs
K32(k_add_co_sdst8,
  "v_add_co_u32 %0, s[10:11], %0, %8\n" "v_add_co_u32 %1, s[12:13], %1, %8\n"
  "v_add_co_u32 %2, s[14:15], %2, %8\n" "v_add_co_u32 %3, s[16:17], %3, %8\n"
  "v_add_co_u32 %4, s[18:19], %4, %8\n" "v_add_co_u32 %5, s[20:21], %5, %8\n"
  "v_add_co_u32 %6, s[22:23], %6, %8\n" "v_add_co_u32 %7, s[24:25], %7, %8\n")
// a real 8-limb carry chain through VCC with the compiler's s_nop 1 spacing
K32(k_addc_chain_nop,
  "v_add_co_u32 %0, vcc, %0, %8\n s_nop 1\n" "v_addc_co_u32 %1, vcc, %1, %9, vcc\n s_nop 1\n"
  "v_addc_co_u32 %2, vcc, %2, %8, vcc\n s_nop 1\n" "v_addc_co_u32 %3, vcc, %3, %9, vcc\n s_nop 1\n"
  "v_addc_co_u32 %4, vcc, %4, %8, vcc\n s_nop 1\n" "v_addc_co_u32 %5, vcc, %5, %9, vcc\n s_nop 1\n"
  "v_addc_co_u32 %6, vcc, %6, %8, vcc\n s_nop 1\n" "v_addc_co_u32 %7, vcc, %7, %9, vcc\n s_nop 1\n")
K32(k_addc_chain_nonop,
  "v_add_co_u32 %0, vcc, %0, %8\n" "v_addc_co_u32 %1, vcc, %1, %9, vcc\n"
  "v_addc_co_u32 %2, vcc, %2, %8, vcc\n" "v_addc_co_u32 %3, vcc, %3, %9, vcc\n"
  "v_addc_co_u32 %4, vcc, %4, %8, vcc\n" "v_addc_co_u32 %5, vcc, %5, %9, vcc\n"
  "v_addc_co_u32 %6, vcc, %6, %8, vcc\n" "v_addc_co_u32 %7, vcc, %7, %9, vcc\n")
// two interleaved carry chains (vcc and s[10:11]) -- does interleaving hide the hazard?
K32(k_addc_2chains,
  "v_add_co_u32 %0, vcc, %0, %8\n" "v_add_co_u32 %4, s[10:11], %4, %8\n"
  "v_addc_co_u32 %1, vcc, %1, %9, vcc\n" "v_addc_co_u32 %5, s[10:11], %5, %9, s[10:11]\n"
  "v_addc_co_u32 %2, vcc, %2, %8, vcc\n" "v_addc_co_u32 %6, s[10:11], %6, %8, s[10:11]\n"
  "v_addc_co_u32 %3, vcc, %3, %9, vcc\n" "v_addc_co_u32 %7, s[10:11], %7, %9, s[10:11]\n")
K32(k_cndmask_vcc, "v_cndmask_b32 %0, %0, %8, vcc\n" "v_cndmask_b32 %1, %1, %8, vcc\n" "v_cndmask_b32 %2, %2, %8, vcc\n"
  "v_cndmask_b32 %3, %3, %8, vcc\n" "v_cndmask_b32 %4, %4, %8, vcc\n" "v_cndmask_b32 %5, %5, %8, vcc\n"
  "v_cndmask_b32 %6, %6, %8, vcc\n" "v_cndmask_b32 %7, %7, %8, vcc\n")
K32(k_cndmask_sgpr, "v_cndmask_b32 %0, %0, %8, s[10:11]\n" "v_cndmask_b32 %1, %1, %8, s[10:11]\n" "v_cndmask_b32 %2, %2, %8, s[10:11]\n"
  "v_cndmask_b32 %3, %3, %8, s[10:11]\n" "v_cndmask_b32 %4, %4, %8, s[10:11]\n" "v_cndmask_b32 %5, %5, %8, s[10:11]\n"
  "v_cndmask_b32 %6, %6, %8, s[10:11]\n" "v_cndmask_b32 %7, %7, %8, s[10:11]\n")
// cndmask fed by a v_cmp each time (the realistic select pattern)
K32(k_cmp_cndmask, "v_cmp_lt_u32 vcc, %0, %8\n" "s_nop 1\n" "v_cndmask_b32 %1, %1, %8, vcc\n" "v_cndmask_b32 %2, %2, %9, vcc\n"
  "v_cndmask_b32 %3, %3, %8, vcc\n" "v_cndmask_b32 %4, %4, %9, vcc\n" "v_cndmask_b32 %5, %5, %8, vcc\n"
  "v_cndmask_b32 %6, %6, %9, vcc\n" "v_cndmask_b32 %7, %7, %8, vcc\n")
K32(k_bfi, "v_bfi_b32 %0, %8, %0, %9\n" "v_bfi_b32 %1, %8, %1, %9\n" "v_bfi_b32 %2, %8, %2, %9\n" "v_bfi_b32 %3, %8, %3, %9\n"
  "v_bfi_b32 %4, %8, %4, %9\n" "v_bfi_b32 %5, %8, %5, %9\n" "v_bfi_b32 %6, %8, %6, %9\n" "v_bfi_b32 %7, %8, %7, %9\n")
K32(k_and, "v_and_b32 %0, %0, %8\n" "v_and_b32 %1, %1, %8\n" "v_and_b32 %2, %2, %8\n" "v_and_b32 %3, %3, %8\n"
  "v_and_b32 %4, %4, %8\n" "v_and_b32 %5, %5, %8\n" "v_and_b32 %6, %6, %8\n" "v_and_b32 %7, %7, %8\n")
K32(k_sub_u32, "v_sub_u32 %0, %0, %8\n" "v_sub_u32 %1, %1, %8\n" "v_sub_u32 %2, %2, %8\n" "v_sub_u32 %3, %3, %8\n"
  "v_sub_u32 %4, %4, %8\n" "v_sub_u32 %5, %5, %8\n" "v_sub_u32 %6, %6, %8\n" "v_sub_u32 %7, %7, %8\n")
K32(k_mov, "v_mov_b32 %0, %8\n" "v_mov_b32 %1, %9\n" "v_mov_b32 %2, %8\n" "v_mov_b32 %3, %9\n"
  "v_mov_b32 %4, %8\n" "v_mov_b32 %5, %9\n" "v_mov_b32 %6, %8\n" "v_mov_b32 %7, %9\n")
K32(k_fma_f32, "v_fma_f32 %0, %0, %8, %9\n" "v_fma_f32 %1, %1, %8, %9\n" "v_fma_f32 %2, %2, %8, %9\n" "v_fma_f32 %3, %3, %8, %9\n"
  "v_fma_f32 %4, %4, %8, %9\n" "v_fma_f32 %5, %5, %8, %9\n" "v_fma_f32 %6, %6, %8, %9\n" "v_fma_f32 %7, %7, %8, %9\n")
K32(k_add3, "v_add3_u32 %0, %0, %8, %9\n" "v_add3_u32 %1, %1, %8, %9\n" "v_add3_u32 %2, %2, %8, %9\n" "v_add3_u32 %3, %3, %8, %9\n"
  "v_add3_u32 %4, %4, %8, %9\n" "v_add3_u32 %5, %5, %8, %9\n" "v_add3_u32 %6, %6, %8, %9\n" "v_add3_u32 %7, %7, %8, %9\n")
K32(k_mul_lo, "v_mul_lo_u32 %0, %0, %8\n" "v_mul_lo_u32 %1, %1, %8\n" "v_mul_lo_u32 %2, %2, %8\n" "v_mul_lo_u32 %3, %3, %8\n"
  "v_mul_lo_u32 %4, %4, %8\n" "v_mul_lo_u32 %5, %5, %8\n" "v_mul_lo_u32 %6, %6, %8\n" "v_mul_lo_u32 %7, %7, %8\n")
K32(k_cmp_only, "v_cmp_lt_u32 s[10:11], %0, %8\n" "v_cmp_lt_u32 s[12:13], %1, %8\n" "v_cmp_lt_u32 s[14:15], %2, %8\n" "v_cmp_lt_u32 s[16:17], %3, %8\n"
  "v_cmp_lt_u32 s[18:19], %4, %8\n" "v_cmp_lt_u32 s[20:21], %5, %8\n" "v_cmp_lt_u32 s[22:23], %6, %8\n" "v_cmp_lt_u32 s[24:25], %7, %8\n")

// hazard correctness probe: 4-limb add of all-ones + 1 must ripple a carry through every limb.
__global__ void k_hazard_probe(uint32_t* out) {
  uint32_t x0 = 0xFFFFFFFFu, x1 = 0xFFFFFFFFu, x2 = 0xFFFFFFFFu, x3 = 0xFFFFFFFFu, one = 1, zero = 0, co;
  asm volatile("v_add_co_u32 %0, vcc, %0, %5\n" "v_addc_co_u32 %1, vcc, %1, %6, vcc\n"
               "v_addc_co_u32 %2, vcc, %2, %6, vcc\n" "v_addc_co_u32 %3, vcc, %3, %6, vcc\n"
               "v_addc_co_u32 %4, vcc, %6, %6, vcc\n"
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=v"(co) : "v"(one), "v"(zero) : "vcc");
  uint32_t ok = (x0 == 0 && x1 == 0 && x2 == 0 && x3 == 0 && co == 1);
  out[blockIdx.x * blockDim.x + threadIdx.x] = ok;
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Case { const char* name; kern_t k; int inst_per_iter; };

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk_ghz = prop.clockRate / 1e6;
  printf("device: %s CUs=%d clockRate=%.3f GHz\n", prop.gcnArchName, cus, clk_ghz);
  const int threads = 256;
  uint32_t* out; CK(hipMalloc(&out, (size_t)cus * 8 * threads * sizeof(uint32_t)));
  {
    k_hazard_probe<<<cus * 4, threads>>>(out); CK(hipDeviceSynchronize());
    std::vector<uint32_t> h((size_t)cus * 4 * threads); CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    size_t good = 0; for (auto v : h) good += v;
    printf("hazard probe (addc chain through VCC with NO s_nop): %zu / %zu lanes correct\n", good, h.size());
  }
  std::vector<Case> cases = {
    {"v_mov_b32", k_mov, 64}, {"v_and_b32", k_and, 64}, {"v_sub_u32", k_sub_u32, 64}, {"v_fma_f32", k_fma_f32, 64},
    {"v_add3_u32", k_add3, 64}, {"v_bfi_b32", k_bfi, 64}, {"v_mul_lo_u32", k_mul_lo, 64},
    {"v_lshl_add_u64 (dep ring)", k_lshl_add_u64, 64},
    {"v_mad_u64_u32 8 sdst pairs", k_mad64_sdst8, 64}, {"v_mad_u64_u32 +0, 8 sdst", k_mad64_c0, 64},
    {"v_add_co_u32 8 sdst pairs", k_add_co_sdst8, 64}, {"v_cmp_lt_u32 8 sdst pairs", k_cmp_only, 64},
    {"addc chain vcc + s_nop 1", k_addc_chain_nop, 64}, {"addc chain vcc no nop", k_addc_chain_nonop, 64},
    {"addc 2 interleaved chains", k_addc_2chains, 64},
    {"v_cndmask vcc", k_cndmask_vcc, 64}, {"v_cndmask s[10:11]", k_cndmask_sgpr, 64},
    {"v_cmp + nop + 7 cndmask (per 8)", k_cmp_cndmask, 64},
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-34s %6s %10s %14s %14s\n", "instruction", "w/SIMD", "ms", "cyc/wave-inst", "Tinst-lane/s");
  for (auto& c : cases) {
    for (int wps : {1, 2, 4}) {
      int blocks = cus * wps;
      c.k<<<blocks, threads>>>(out, 1); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); c.k<<<blocks, threads>>>(out, 2); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double wi = (double)wps * ITERS * c.inst_per_iter;
      printf("%-34s %6d %10.4f %14.3f %14.3f\n", c.name, wps, ms, ms * 1e-3 * clk_ghz * 1e9 / wi,
             (double)blocks * threads * ITERS * c.inst_per_iter / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
