// salu_carry.hip -- can the carry captures of the 256 x 256-bit product move to the scalar unit?
//
// Every partial product of the shipped product scanning costs two VALU issue slots: the v_mad_u64_u32 and the v_addc that
// counts its carry.  All hot kernels are VALU-issue bound (one VALU instruction per 3.8-3.95 cycles per SIMD) while their
// scalar unit is nearly idle, and the scalar unit issues beside the VALU from another wavefront of the same SIMD.  Variant B
// (gen_salu_carry.py) sends the carries of the wide columns to SGPR pairs, sums the lane masks with scalar 3:2 compressors
// and rebuilds the overflow word with one v_addc per bit plane: 24 VALU instructions fewer per product, 142 scalar ones more.
// Timed at three wavefronts per SIMD like the kernels; both variants must produce identical results.
//
//   python tools/microbench/gen_salu_carry.py [MIN_N]
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench/salu_carry tools/microbench/salu_carry.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "salu_carry.inc"

typedef unsigned u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

#define LOAD_FIXED                                                                                            \
  "v_mov_b32 v32, %0\n\tv_mov_b32 v33, %1\n\tv_mov_b32 v34, %2\n\tv_mov_b32 v35, %3\n\t"                         \
  "v_mov_b32 v36, %4\n\tv_mov_b32 v37, %5\n\tv_mov_b32 v38, %6\n\tv_mov_b32 v39, %7\n\t"                         \
  "v_mov_b32 v40, %8\n\tv_mov_b32 v41, %9\n\tv_mov_b32 v42, %10\n\tv_mov_b32 v43, %11\n\t"                       \
  "v_mov_b32 v44, %12\n\tv_mov_b32 v45, %13\n\tv_mov_b32 v46, %14\n\tv_mov_b32 v47, %15\n\t"                     \
  "s_mov_b32 s19, %16\n\t"
#define STORE_FIXED                                                                                           \
  "v_mov_b32 %0, v32\n\tv_mov_b32 %1, v33\n\tv_mov_b32 %2, v34\n\tv_mov_b32 %3, v35\n\t"                         \
  "v_mov_b32 %4, v36\n\tv_mov_b32 %5, v37\n\tv_mov_b32 %6, v38\n\tv_mov_b32 %7, v39\n\t"
#define LOOP_TAIL "s_sub_u32 s19, s19, 1\n\ts_cmp_lg_u32 s19, 0\n\ts_cbranch_scc1 .Lsc_loop%=\n\t"

template <int VARIANT>
__global__ __launch_bounds__(256, 3) void k_product(const u32* __restrict__ in, u32* __restrict__ out, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  u32 a0 = in[t * 16 + 0], a1 = in[t * 16 + 1], a2 = in[t * 16 + 2], a3 = in[t * 16 + 3], a4 = in[t * 16 + 4],
      a5 = in[t * 16 + 5], a6 = in[t * 16 + 6], a7 = in[t * 16 + 7];
  u32 b0 = in[t * 16 + 8], b1 = in[t * 16 + 9], b2 = in[t * 16 + 10], b3 = in[t * 16 + 11], b4 = in[t * 16 + 12],
      b5 = in[t * 16 + 13], b6 = in[t * 16 + 14], b7 = in[t * 16 + 15];
  if (VARIANT == 0) {
    asm volatile(LOAD_FIXED ".Lsc_loop%=:\n\t" SALU_CARRY_BODY_A LOOP_TAIL STORE_FIXED
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7), "s"(iters)
                 : SALU_CARRY_CLOBBERS);
  } else {
    asm volatile(LOAD_FIXED ".Lsc_loop%=:\n\t" SALU_CARRY_BODY_B LOOP_TAIL STORE_FIXED
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7), "s"(iters)
                 : SALU_CARRY_CLOBBERS);
  }
  out[t * 8 + 0] = a0; out[t * 8 + 1] = a1; out[t * 8 + 2] = a2; out[t * 8 + 3] = a3;
  out[t * 8 + 4] = a4; out[t * 8 + 5] = a5; out[t * 8 + 6] = a6; out[t * 8 + 7] = a7;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, iters = 4000;
  std::vector<u32> h;
  u32 *din, *dout[2];
  unsigned long long s = 99;
  for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {   // 1, 2, 3 wavefronts per SIMD (a workgroup is four)
    const size_t threads = (size_t)cus * wgs_per_cu * 256;
    h.resize(threads * 16);
    for (auto& v : h) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; v = (u32)(s >> 32); }
    CK(hipMalloc(&din, threads * 64));
    CK(hipMemcpy(din, h.data(), threads * 64, hipMemcpyHostToDevice));
    float ms[2] = {0, 0};
    for (int v = 0; v < 2; ++v) {
      CK(hipMalloc(&dout[v], threads * 32));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (v == 0) hipLaunchKernelGGL(k_product<0>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, din, dout[v], iters);
        else hipLaunchKernelGGL(k_product<1>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, din, dout[v], iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[v], e0, e1));
      }
    }
    std::vector<u32> r0(threads * 8), r1(threads * 8);
    CK(hipMemcpy(r0.data(), dout[0], threads * 32, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r1.data(), dout[1], threads * 32, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < r0.size(); ++i) bad += r0[i] != r1[i];
    // SIMD-ns per product: the kernel runs `iters` products per lane on cus * 4 SIMDs holding wgs_per_cu wavefronts each
    const double per[2] = {ms[0] * 1e6 / iters / wgs_per_cu, ms[1] * 1e6 / iters / wgs_per_cu};
    printf("%d wavefront(s) per SIMD: A (v_addc per carry) %.3f ms = %.1f SIMD-ns per product; B (carries on the scalar unit) %.3f ms = %.1f; "
           "B / A = %.3f; results %s\n", wgs_per_cu, ms[0], per[0], ms[1], per[1], ms[1] / ms[0], bad ? "DIFFER" : "identical");
    CK(hipFree(din)); CK(hipFree(dout[0])); CK(hipFree(dout[1]));
  }
  return 0;
}
