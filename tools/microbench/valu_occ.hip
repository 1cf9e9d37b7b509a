// Issue cost of the limb-arithmetic instructions on gfx950 against RESIDENT wavefronts per SIMD (1..8): the same dependent
// chains as valu_latency.hip, but with the temporaries in low registers so that the kernels stay under 64 VGPRs and
// all of the requested wavefronts are resident at once (valu_latency.hip clobbers v250.., i.e. two per SIMD at most --
// its "3" rows are two resident wavefronts and a tail).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_occ valu_occ.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
#define REP4(s) s s s s
#define REP64(s) REP4(REP4(REP4(s)))
#define ITERS 1024
#define KERN(name, INS) __global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t seed) { \
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u; uint64_t c = a; uint32_t d = b, e2 = a + 7; \
  for (int it = 0; it < ITERS; ++it) { asm volatile(REP64(INS) : "+v"(c), "+v"(d), "+v"(e2) : "v"(a), "v"(b) : "vcc", "s10", "s11", "s12", "s13", "v40", "v41", "v42"); } \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)c ^ (uint32_t)(c >> 32) ^ d ^ e2; }
KERN(k_mad_dep, "v_mad_u64_u32 %0, s[10:11], %3, %4, %0\n")
KERN(k_mad_addc, "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n")
KERN(k_mad_mov_mad, "v_mad_u64_u32 v[40:41], s[10:11], %3, %4, v[40:41]\n v_mov_b32_e32 v42, v41\n v_mad_u64_u32 v[40:41], s[10:11], v42, %4, v[40:41]\n")
KERN(k_addc_dep, "v_addc_co_u32_e32 %1, vcc, %1, %3, vcc\n")
KERN(k_add_dep, "v_add_u32_e32 %1, %1, %3\n")
KERN(k_mov_dep, "v_mov_b32_e32 %1, %2\n v_mov_b32_e32 %2, %1\n")
KERN(k_mullo_dep, "v_mul_lo_u32 %1, %1, %3\n")
KERN(k_alignbit_dep, "v_alignbit_b32 %1, %1, %3, 31\n")
KERN(k_mad_indep2, "v_mad_u64_u32 %0, s[10:11], %3, %4, %0\n v_mul_lo_u32 %1, %1, %3\n")
KERN(k_mad_then_add, "v_mad_u64_u32 %0, s[10:11], %3, %4, %0\n v_add_u32_e32 %1, %1, %3\n v_add_u32_e32 %2, %2, %4\n")
// where a cheap instruction sits relative to a mad and the addc that takes its carry
KERN(k_mad_mov_addc, "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_mov_b32_e32 v42, %2\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n")
KERN(k_mad_addc_mov, "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n v_mov_b32_e32 v42, %2\n")
KERN(k_mad_sub_addc, "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_sub_u32_e32 %2, %2, %3\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n")
KERN(k_mad_addc_sub, "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n v_sub_u32_e32 %2, %2, %3\n")
KERN(k_mad_2mov_addc, "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_mov_b32_e32 v42, %2\n v_mov_b32_e32 v41, %2\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n")
KERN(k_mad_addc_e64, "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n")
KERN(k_mad_s_addc_s, "v_mad_u64_u32 %0, s[10:11], %3, %4, %0\n v_addc_co_u32_e64 %1, s[10:11], 0, %1, s[10:11]\n")
KERN(k_mad_mad_addc_addc, "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_mad_u64_u32 v[40:41], s[10:11], %3, %4, v[40:41]\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n v_addc_co_u32_e64 %2, s[10:11], 0, %2, s[10:11]\n")
// which carry register, which encoding of the addc
KERN(k_mad_vcc_addc_e64, "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_addc_co_u32_e64 %1, vcc, 0, %1, vcc\n")
KERN(k_mad_s_addc_alt, "v_mad_u64_u32 %0, s[10:11], %3, %4, %0\n v_addc_co_u32_e64 %1, s[10:11], 0, %1, s[10:11]\n v_mad_u64_u32 %0, s[12:13], %3, %4, %0\n v_addc_co_u32_e64 %1, s[12:13], 0, %1, s[12:13]\n")
KERN(k_mad_s_addc_keep, "v_mad_u64_u32 %0, s[10:11], %3, %4, %0\n v_addc_co_u32_e64 %1, s[12:13], 0, %1, s[10:11]\n")
KERN(k_mad_vcc_addc_keep, "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_addc_co_u32_e64 %1, s[12:13], 0, %1, vcc\n")
// carry chains: VOP2 (e32, implicit vcc) against VOP3 (e64) encodings
KERN(k_addc_dep_e64, "v_addc_co_u32_e64 %1, vcc, %1, %3, vcc\n")
KERN(k_addc_dep_e64s, "v_addc_co_u32_e64 %1, s[10:11], %1, %3, s[10:11]\n")
KERN(k_chain4_e32, "v_add_co_u32_e32 %1, vcc, %1, %3\n v_addc_co_u32_e32 %2, vcc, %2, %4, vcc\n v_addc_co_u32_e32 v40, vcc, v40, %3, vcc\n v_addc_co_u32_e32 v41, vcc, v41, %4, vcc\n")
KERN(k_chain4_e64, "v_add_co_u32_e64 %1, vcc, %1, %3\n v_addc_co_u32_e64 %2, vcc, %2, %4, vcc\n v_addc_co_u32_e64 v40, vcc, v40, %3, vcc\n v_addc_co_u32_e64 v41, vcc, v41, %4, vcc\n")
KERN(k_subchain4_e32, "v_sub_co_u32_e32 %1, vcc, %1, %3\n v_subb_co_u32_e32 %2, vcc, %2, %4, vcc\n v_subb_co_u32_e32 v40, vcc, v40, %3, vcc\n v_subb_co_u32_e32 v41, vcc, v41, %4, vcc\n")
KERN(k_subchain4_e64, "v_sub_co_u32_e64 %1, vcc, %1, %3\n v_subb_co_u32_e64 %2, vcc, %2, %4, vcc\n v_subb_co_u32_e64 v40, vcc, v40, %3, vcc\n v_subb_co_u32_e64 v41, vcc, v41, %4, vcc\n")
typedef void (*kern_t)(uint32_t*, uint32_t);
struct Case { const char* name; kern_t k; int per; };
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); int cus = prop.multiProcessorCount;
  double ghz = prop.clockRate / 1e6;
  uint32_t* out; CK(hipMalloc(&out, (size_t)cus * 16 * 256 * 4));
  Case cases[] = {{"v_mad_u64_u32 dependent", k_mad_dep, 1}, {"mad + addc(vcc) pair, dependent mads", k_mad_addc, 2},
    {"mad -> mov(lo) -> mad", k_mad_mov_mad, 3}, {"v_addc chain", k_addc_dep, 1}, {"v_add_u32 dependent", k_add_dep, 1},
    {"v_mov ping-pong", k_mov_dep, 2}, {"v_mul_lo_u32 dependent", k_mullo_dep, 1}, {"v_alignbit dependent", k_alignbit_dep, 1},
    {"mad dep + independent mul_lo", k_mad_indep2, 2}, {"mad dep + 2 independent adds", k_mad_then_add, 3},
    {"mad, MOV, addc", k_mad_mov_addc, 3}, {"mad, addc, MOV", k_mad_addc_mov, 3}, {"mad, SUB, addc", k_mad_sub_addc, 3},
    {"mad, addc, SUB", k_mad_addc_sub, 3}, {"mad, MOV, MOV, addc", k_mad_2mov_addc, 4}, {"mad, addc_e64(0,0)", k_mad_addc_e64, 2},
    {"mad(s[10:11]), addc_e64(s[10:11])", k_mad_s_addc_s, 2}, {"mad, mad', addc, addc' (two chains interleaved)", k_mad_mad_addc_addc, 4},
    {"mad(vcc), addc_e64(vcc)", k_mad_vcc_addc_e64, 2}, {"mad(sA), addc_e64(sA); mad(sB), addc_e64(sB)", k_mad_s_addc_alt, 4},
    {"v_addc chain e64 (vcc)", k_addc_dep_e64, 1}, {"v_addc chain e64 (sgpr pair)", k_addc_dep_e64s, 1},
    {"add_co + 3 addc, e32", k_chain4_e32, 4}, {"add_co + 3 addc, e64", k_chain4_e64, 4},
    {"sub_co + 3 subb, e32", k_subchain4_e32, 4}, {"sub_co + 3 subb, e64", k_subchain4_e64, 4},
    {"mad(sA), addc_e64 carry-in sA, carry-out sB", k_mad_s_addc_keep, 2}, {"mad(vcc), addc_e64 carry-in vcc, carry-out sB", k_mad_vcc_addc_keep, 2}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-40s %6s %10s %14s\n", "chain", "w/SIMD", "ms", "cyc/instr");
  for (auto& c : cases) for (int wps : {1, 2, 3, 4, 5, 6, 8}) {
    int blocks = cus * wps; float ms;
    for (int r = 0; r < 2; ++r) { CK(hipEventRecord(e0)); c.k<<<blocks, 256>>>(out, r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    double inst = (double)ITERS * 64 * c.per;
    printf("%-40s %6d %10.4f %14.3f   (per SIMD: %.3f cyc/instr)\n", c.name, wps, ms, ms * 1e-3 * ghz * 1e9 / inst, ms * 1e-3 * ghz * 1e9 / inst / wps);
  }
  return 0;
}
