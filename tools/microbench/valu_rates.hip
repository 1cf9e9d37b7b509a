// Integer-VALU issue-rate microbenchmark for gfx950 (MI355X).
// Measures wave-instruction throughput of the candidate limb-arithmetic
// instructions so the limb representation and the roofline peak are chosen
// from measurement, not assumption ("peak MAD32/s must be
// measured on the box").
//
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

#define REP8(s) s s s s s s s s
#define ITERS 2048

// Each kernel: 8 independent dependency chains, 8x8 = 64 instructions per loop
// body (8 chains x 8 repeats), ITERS iterations.
#define KERNEL_BEGIN(name) \
__global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t seed) { \
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u; \
  uint64_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 2, c4 = a + 3, c5 = b + 4, c6 = a + 5, c7 = b + 6; \
  uint32_t d0 = a, d1 = b, d2 = a + 1, d3 = b + 2, d4 = a + 3, d5 = b + 4, d6 = a + 5, d7 = b + 6; \
  double f0 = a, f1 = b, f2 = 1.5, f3 = 2.5, f4 = 3.5, f5 = 4.5, f6 = 5.5, f7 = 6.5; \
  double fa = 1.0000001, fb = 0.5; \
  (void)c0;(void)c1;(void)c2;(void)c3;(void)c4;(void)c5;(void)c6;(void)c7; \
  (void)d0;(void)d1;(void)d2;(void)d3;(void)d4;(void)d5;(void)d6;(void)d7; \
  (void)f0;(void)f1;(void)f2;(void)f3;(void)f4;(void)f5;(void)f6;(void)f7;(void)fa;(void)fb; \
  for (int it = 0; it < ITERS; ++it) {

#define KERNEL_END \
  } \
  uint64_t cs = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7; \
  uint32_t ds = d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7; \
  double fs = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7; \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)cs ^ (uint32_t)(cs >> 32) ^ ds ^ (uint32_t)fs; \
}

#define CHAIN8_64(INS) \
  asm volatile(REP8( \
    INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7")) \
    : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) \
    : "v"(a), "v"(b) : "vcc");

#define CHAIN8_32(INS) \
  asm volatile(REP8( \
    INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7")) \
    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) \
    : "v"(a), "v"(b) : "vcc");

#define CHAIN8_F64(INS) \
  asm volatile(REP8( \
    INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7")) \
    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) \
    : "v"(fa), "v"(fb) : "vcc");

#define I_MAD64(r)    "v_mad_u64_u32 " r ", vcc, %8, %9, " r "\n"
#define I_MAD64S(r)   "v_mad_u64_u32 " r ", s[10:11], %8, %9, " r "\n"
#define I_MULLO(r)    "v_mul_lo_u32 " r ", " r ", %8\n"
#define I_MULHI(r)    "v_mul_hi_u32 " r ", " r ", %8\n"
#define I_MAD24(r)    "v_mad_u32_u24 " r ", " r ", %8, %9\n"
#define I_MULHI24(r)  "v_mul_hi_u32_u24 " r ", " r ", %8\n"
#define I_ADD(r)      "v_add_u32 " r ", " r ", %8\n"
#define I_ADDCO(r)    "v_add_co_u32 " r ", vcc, " r ", %8\n"
#define I_ADDC(r)     "v_addc_co_u32 " r ", vcc, " r ", %8, vcc\n"
#define I_ADD3(r)     "v_add3_u32 " r ", " r ", %8, %9\n"
#define I_LSHLADD(r)  "v_lshl_add_u32 " r ", " r ", 3, %8\n"
#define I_ALIGNBIT(r) "v_alignbit_b32 " r ", " r ", %8, 31\n"
#define I_CNDMASK(r)  "v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define I_XOR(r)      "v_xor_b32 " r ", " r ", %8\n"
#define I_MADU16(r)   "v_mad_u32_u16 " r ", " r ", %8, %9\n"
#define I_DOT4(r)     "v_dot4_u32_u8 " r ", " r ", %8, %9\n"
#define I_DOT2(r)     "v_dot2_u32_u16 " r ", " r ", %8, %9\n"
#define I_FMA64(r)    "v_fma_f64 " r ", " r ", %8, %9\n"
#define I_FMA32(r)    "v_fma_f32 " r ", " r ", %8, %9\n"
#define I_LSHL64(r)   "v_lshlrev_b64 " r ", 1, " r "\n"
#define I_ADDCO_S(r)  "v_add_co_u32 " r ", s[10:11], " r ", %8\n"
#define I_ADDC_S(r)   "v_addc_co_u32 " r ", s[10:11], " r ", %8, s[10:11]\n"

KERNEL_BEGIN(k_mad_u64_u32)   CHAIN8_64(I_MAD64)    KERNEL_END
KERNEL_BEGIN(k_mul_lo_u32)    CHAIN8_32(I_MULLO)    KERNEL_END
KERNEL_BEGIN(k_mul_hi_u32)    CHAIN8_32(I_MULHI)    KERNEL_END
KERNEL_BEGIN(k_mad_u32_u24)   CHAIN8_32(I_MAD24)    KERNEL_END
KERNEL_BEGIN(k_mul_hi_u32_u24) CHAIN8_32(I_MULHI24) KERNEL_END
KERNEL_BEGIN(k_add_u32)       CHAIN8_32(I_ADD)      KERNEL_END
KERNEL_BEGIN(k_add_co_u32)    CHAIN8_32(I_ADDCO)    KERNEL_END
KERNEL_BEGIN(k_addc_co_u32)   CHAIN8_32(I_ADDC)     KERNEL_END
KERNEL_BEGIN(k_add3_u32)      CHAIN8_32(I_ADD3)     KERNEL_END
KERNEL_BEGIN(k_lshl_add_u32)  CHAIN8_32(I_LSHLADD)  KERNEL_END
KERNEL_BEGIN(k_alignbit_b32)  CHAIN8_32(I_ALIGNBIT) KERNEL_END
KERNEL_BEGIN(k_cndmask_b32)   CHAIN8_32(I_CNDMASK)  KERNEL_END
KERNEL_BEGIN(k_xor_b32)       CHAIN8_32(I_XOR)      KERNEL_END
KERNEL_BEGIN(k_mad_u32_u16)   CHAIN8_32(I_MADU16)   KERNEL_END
KERNEL_BEGIN(k_dot4_u32_u8)   CHAIN8_32(I_DOT4)     KERNEL_END
KERNEL_BEGIN(k_dot2_u32_u16)  CHAIN8_32(I_DOT2)     KERNEL_END
KERNEL_BEGIN(k_fma_f64)       CHAIN8_F64(I_FMA64)   KERNEL_END
KERNEL_BEGIN(k_lshlrev_b64)   CHAIN8_64(I_LSHL64)   KERNEL_END

// mixed: 1 mad_u64_u32 followed by 2 add/addc -- the realistic limb-mul inner pattern
#define I_MIX(r) "v_mad_u64_u32 " r ", vcc, %8, %9, " r "\n" "v_add_co_u32 %10, vcc, %10, %8\n" "v_addc_co_u32 %11, vcc, %11, %9, vcc\n"
__global__ __launch_bounds__(256) void k_mix_mad_2add(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 2, c4 = a + 3, c5 = b + 4, c6 = a + 5, c7 = b + 6;
  uint32_t e0 = a, e1 = b;
  for (int it = 0; it < ITERS; ++it) {
    asm volatile(REP8(
      I_MIX("%0") I_MIX("%1") I_MIX("%2") I_MIX("%3") I_MIX("%4") I_MIX("%5") I_MIX("%6") I_MIX("%7"))
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(a), "v"(b), "v"(e0), "v"(e1) : "vcc");
  }
  uint64_t cs = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)cs ^ (uint32_t)(cs >> 32) ^ e0 ^ e1;
}

// one dependent chain only (latency): 64 dependent mad_u64_u32
#define I_DEP(r) "v_mad_u64_u32 %0, vcc, %8, %9, %0\n"
KERNEL_BEGIN(k_mad_u64_u32_dep) CHAIN8_64(I_DEP) KERNEL_END
#define I_DEPADD(r) "v_add_u32 %0, %0, %8\n"
KERNEL_BEGIN(k_add_u32_dep) CHAIN8_32(I_DEPADD) KERNEL_END

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Case { const char* name; kern_t k; int inst_per_iter; };

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  double clk_ghz = prop.clockRate / 1e6;
  printf("device: %s  CUs=%d  clockRate=%.3f GHz  arch=%s\n", prop.name, cus, clk_ghz, prop.gcnArchName);
  std::vector<Case> cases = {
    {"v_fma_f64", k_fma_f64, 64}, {"v_add_u32", k_add_u32, 64}, {"v_add_co_u32", k_add_co_u32, 64},
    {"v_addc_co_u32", k_addc_co_u32, 64}, {"v_add3_u32", k_add3_u32, 64}, {"v_lshl_add_u32", k_lshl_add_u32, 64},
    {"v_alignbit_b32", k_alignbit_b32, 64}, {"v_cndmask_b32", k_cndmask_b32, 64}, {"v_xor_b32", k_xor_b32, 64},
    {"v_lshlrev_b64", k_lshlrev_b64, 64},
    {"v_mad_u64_u32", k_mad_u64_u32, 64}, {"v_mul_lo_u32", k_mul_lo_u32, 64}, {"v_mul_hi_u32", k_mul_hi_u32, 64},
    {"v_mad_u32_u24", k_mad_u32_u24, 64}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 64},
    {"v_mad_u32_u16", k_mad_u32_u16, 64}, {"v_dot4_u32_u8", k_dot4_u32_u8, 64}, {"v_dot2_u32_u16", k_dot2_u32_u16, 64},
    {"mix(1 mad64 + add_co + addc)", k_mix_mad_2add, 192},
    {"v_mad_u64_u32 (1 dep chain)", k_mad_u64_u32_dep, 64}, {"v_add_u32 (1 dep chain)", k_add_u32_dep, 64},
  };
  const int threads = 256;
  uint32_t* out; CK(hipMalloc(&out, (size_t)cus * 8 * threads * sizeof(uint32_t) * 2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-34s %6s %10s %14s %14s\n", "instruction", "w/SIMD", "ms", "cyc/wave-inst", "Tinst-lane/s");
  for (auto& c : cases) {
    for (int waves_per_simd : {1, 2, 4, 8}) {
      int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD per block
      c.k<<<blocks, threads>>>(out, 1);   // warm
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      c.k<<<blocks, threads>>>(out, 2);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double wave_insts_per_simd = (double)waves_per_simd * ITERS * c.inst_per_iter;
      double cyc = ms * 1e-3 * clk_ghz * 1e9 / wave_insts_per_simd;   // SIMD cycles per wave-instruction (at nominal clock)
      double lane_rate = (double)blocks * threads * ITERS * c.inst_per_iter / (ms * 1e-3) / 1e12;
      printf("%-34s %6d %10.4f %14.3f %14.3f\n", c.name, waves_per_simd, ms, cyc, lane_rate);
    }
  }
  CK(hipFree(out));
  return 0;
}
