// A FAIRER prototype of north_star's thread mapping than coop_mul.hip: eight lanes of a wavefront share one
// 256 x 256 -> 512-bit product (eight products per wavefront), lane j holding word j of both operands.  Operand
// words travel with __shfl (ds_bpermute), lane j accumulates columns j and j + 8 in 96-bit accumulators, and the
// carries ripple from lane to lane with __shfl_up -- the "__shfl-based carry propagation inside the Montgomery-mul
// inner loop" of the sketch.  Compared with the shipped mapping (one product per LANE, mul_wide) on the same
// operands, results checked against each other.  Microbenchmark only, not a product path.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../forge_ec_amd/csrc -o coop_mul_shfl coop_mul_shfl.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "limbs.hpp"
using namespace fecgpu;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITERS = 256;

// lane j of each 8-lane group: in a_j, b_j; out word j (lo) and word j + 8 (hi) of the product
__device__ __forceinline__ void coop8_mul_wide(u32 a, u32 b, u32& lo_out, u32& hi_out) {
  const int lane = threadIdx.x & 63, j = lane & 7, base = lane & ~7;
  u64 acc_lo = 0, acc_hi = 0;   // columns j and j + 8, with overflow counts
  u32 ov_lo = 0, ov_hi = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u32 ai = __shfl(a, base + i);
    const u32 bk = __shfl(b, base + ((j - i) & 7));
    const u64 p = (u64)ai * bk;
    if (i <= j) { const u64 s = acc_lo + p; ov_lo += s < p; acc_lo = s; }
    else        { const u64 s = acc_hi + p; ov_hi += s < p; acc_hi = s; }
  }
  // column k contributes (lo32, hi32, ov) to words k, k + 1, k + 2: gather the three contributions per word
  // word j     = lo(col j)   + hi(col j-1)  + ov(col j-2)        (cols < 0: none)
  // word j + 8 = lo(col j+8) + hi(col j+7)  + ov(col j+6)        (col 15 does not exist: lane 7's hi column is empty)
  u32 h1 = __shfl_up((u32)(acc_lo >> 32), 1), o2 = __shfl_up(ov_lo, 2);
  if (j < 1) h1 = 0;
  if (j < 2) o2 = 0;
  u64 w_lo = (u64)(u32)acc_lo + h1 + o2;
  // for the high half the neighbours are: hi(col j+7) = lane j-1's hi column for j >= 1, lane 7's LO column for j == 0
  u32 hh = __shfl((u32)(acc_hi >> 32), base + ((j - 1) & 7)), hl7 = __shfl((u32)(acc_lo >> 32), base + 7);
  u32 oh = __shfl(ov_hi, base + ((j - 2) & 7)), ol6 = __shfl(ov_lo, base + 6), ol7 = __shfl(ov_lo, base + 7);
  const u32 n1 = j == 0 ? hl7 : hh;
  const u32 n2 = j == 0 ? ol6 : (j == 1 ? ol7 : oh);
  u64 w_hi = (u64)(u32)acc_hi + n1 + n2;
  // ripple the carries (each < 3) up the sixteen words: lo half lane 0 -> 7, then into the hi half lane 0 -> 7
  u32 c = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const u32 cin = __shfl(c, base + ((k - 1) & 7));
    if (j == k) { w_lo += (k == 0 ? 0 : cin); c = (u32)(w_lo >> 32); }
  }
  u32 c_top = __shfl(c, base + 7);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const u32 cin = k == 0 ? c_top : __shfl(c, base + k - 1);
    if (j == k) { w_hi += cin; c = (u32)(w_hi >> 32); }
  }
  lo_out = (u32)w_lo;
  hi_out = (u32)w_hi;
}

// eight products per wavefront, ITERS dependent rounds (the product's low half feeds the next round's operand)
__global__ __launch_bounds__(256) void k_coop8(const u32* in, u32* outg) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // lane g holds word (g & 7) of product g >> 3
  u32 a = in[2 * g], b = in[2 * g + 1], lo = 0, hi = 0;
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
    coop8_mul_wide(a, b, lo, hi);
    a = lo ^ hi;
    b ^= hi;
  }
  outg[2 * g] = lo;
  outg[2 * g + 1] = hi;
}
// one product per lane with the shipped mul_wide, same dependency pattern; in/out word-interleaved the same way
__global__ __launch_bounds__(256) void k_lane(const u32* in, u32* outg, size_t nprod) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nprod) return;
  fe a, b;
  for (int w = 0; w < 8; ++w) { a.w[w] = in[2 * (p * 8 + w)]; b.w[w] = in[2 * (p * 8 + w) + 1]; }
  u32 t[16];
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
    mul_wide(t, a, b);
    for (int w = 0; w < 8; ++w) { a.w[w] = t[w] ^ t[8 + w]; b.w[w] ^= t[8 + w]; }
  }
  for (int w = 0; w < 8; ++w) { outg[2 * (p * 8 + w)] = t[w]; outg[2 * (p * 8 + w) + 1] = t[8 + w]; }
}

int main() {
  const size_t nprod = 1 << 20, nlane = nprod * 8;
  std::vector<u32> h(2 * nlane);
  u64 x = 0x243F6A8885A308D3ULL;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (u32)(x >> 16); }
  u32 *din, *d1, *d2;
  CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&d1, h.size() * 4)); CK(hipMalloc(&d2, h.size() * 4));
  CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms_coop = 0, ms_lane = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_coop8, dim3(nlane / 256), dim3(256), 0, 0, din, d1); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_coop, e0, e1));
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lane, dim3(nprod / 256), dim3(256), 0, 0, din, d2, nprod); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_lane, e0, e1));
  }
  std::vector<u32> r1(h.size()), r2(h.size());
  CK(hipMemcpy(r1.data(), d1, h.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), d2, h.size() * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < h.size(); ++i) bad += r1[i] != r2[i];
  const double prods = (double)nprod * ITERS;
  printf("eight lanes per product (__shfl operands + __shfl carry ripple): %.3f ms  %.2f G products/s\n", ms_coop, prods / ms_coop / 1e6);
  printf("one lane per product (mul_wide):                                 %.3f ms  %.2f G products/s\n", ms_lane, prods / ms_lane / 1e6);
  printf("ratio %.2fx, mismatching words %zu of %zu\n", ms_coop / ms_lane, bad, h.size());
  return bad ? 1 : 0;
}
