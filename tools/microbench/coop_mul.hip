// Prototype of the OTHER thread mapping (north_star: one wavefront per scalar-mul, 256-bit limbs
// staged in LDS, cross-lane carry propagation): the 256x256 -> 512-bit product of ONE pair of
// operands computed cooperatively by the 64 lanes of a wavefront -- lane (i,j) forms a_i*b_j, the
// 15 column sums are reduced through LDS, carries are resolved across lanes with a ballot-based
// carry-lookahead.  Compared with the shipped mapping (one product per LANE, mul_wide) on the same
// operands; both are checked against each other.  Microbenchmark only, not a product path.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../forge_ec_amd/csrc -o coop_mul coop_mul.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "limbs.hpp"
using namespace fecgpu;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITERS = 512;

// one wavefront = one multiplication.  a,b: 8 words each in LDS; out: 16 words in LDS.
__device__ __forceinline__ void coop_mul_wide(const u32* a, const u32* b, u32* out, u64* col /*15*8 u64*/) {
  const int lane = threadIdx.x & 63;
  const int i = lane >> 3, j = lane & 7;
  u64 p = (u64)a[i] * b[j];
  col[(i + j) * 8 + i] = p;
  __builtin_amdgcn_wave_barrier();
  // lanes 0..14: 96-bit column sums
  u32 lo = 0, hi = 0, ov = 0;
  if (lane < 15) {
    int i0 = lane > 7 ? lane - 7 : 0, i1 = lane < 7 ? lane : 7;
    u64 acc = 0;
    for (int q = i0; q <= i1; ++q) {
      u64 v = col[lane * 8 + q];
      u64 s = acc + v;
      ov += s < v;
      acc = s;
    }
    lo = (u32)acc; hi = (u32)(acc >> 32);
  }
  // word k = lo_k + hi_{k-1} + ov_{k-2} (+ carries): three-operand sum per lane, then lookahead
  u32 hp = __shfl_up(hi, 1), op = __shfl_up(ov, 2);
  if (lane == 0) hp = 0;
  if (lane < 2) op = 0;
  u64 s = (u64)lo + hp + op;          // < 3 * 2^32
  u32 w = (u32)s, c = (u32)(s >> 32);  // carry 0..2 into lane+1
  // two rounds resolve the multi-valued carries into single-bit generate/propagate form
  u32 cin = __shfl_up(c, 1); if (lane == 0) cin = 0;
  u64 s2 = (u64)w + cin;
  w = (u32)s2;
  u32 g = (u32)(s2 >> 32);             // 0/1
  // carry-lookahead across lanes: G = generate mask, P = propagate mask (word == 0xFFFFFFFF)
  unsigned long long G = __ballot(g != 0), P = __ballot(w == 0xFFFFFFFFu);
  // carry-in(k) = G(k-1) | (P(k-1) & carry-in(k-1)), evaluated on the scalar unit (wave-uniform)
  unsigned long long cm = 0;
  for (int k = 1; k < 16; ++k) {
    unsigned long long prev = 1ull << (k - 1);
    if ((G & prev) || ((P & prev) && (cm & prev))) cm |= 1ull << k;
  }
  w += (u32)((cm >> lane) & 1);
  if (lane < 16) out[lane] = w;
}

__global__ __launch_bounds__(256) void k_coop(const u32* in, u32* outg) {
  __shared__ u32 sa[4][8], sb[4][8], so[4][16];
  __shared__ u64 col[4][15 * 8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane < 8) { sa[wave][lane] = in[(blockIdx.x * 4 + wave) % 1024 * 16 + lane]; sb[wave][lane] = in[(blockIdx.x * 4 + wave) % 1024 * 16 + 8 + lane]; }
  __syncthreads();
  for (int it = 0; it < ITERS; ++it) {
    coop_mul_wide(sa[wave], sb[wave], so[wave], col[wave]);
    __builtin_amdgcn_wave_barrier();
    if (lane < 8) sa[wave][lane] = so[wave][lane] ^ so[wave][lane + 8];   // dependent chain like the lane version
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < 16) outg[(blockIdx.x * 4 + wave) * 16 + lane] = so[wave][lane];
}
__global__ __launch_bounds__(256) void k_lane(const u32* in, u32* outg) {
  fe a, b; int e = (blockIdx.x * 256 + threadIdx.x) % 1024;
  for (int i = 0; i < 8; ++i) { a.w[i] = in[e * 16 + i]; b.w[i] = in[e * 16 + 8 + i]; }
  u32 t[16];
  for (int it = 0; it < ITERS; ++it) { mul_wide(t, a, b); for (int i = 0; i < 8; ++i) a.w[i] = t[i] ^ t[i + 8]; }
  for (int i = 0; i < 16; ++i) outg[(size_t)(blockIdx.x * 256 + threadIdx.x) * 16 + i] = t[i];
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount, blocks = cus * 2;
  std::vector<u32> h(1024 * 16); for (auto& v : h) v = (u32)rand() * 2654435761u + (u32)rand();
  u32 *in, *o1, *o2; CK(hipMalloc(&in, h.size() * 4)); CK(hipMalloc(&o1, (size_t)blocks * 4 * 16 * 4)); CK(hipMalloc(&o2, (size_t)blocks * 256 * 16 * 4));
  CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
  k_coop<<<blocks, 256>>>(in, o1); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); k_coop<<<blocks, 256>>>(in, o1); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  double coop_rate = (double)blocks * 4 * ITERS / (ms * 1e-3);
  printf("wave-per-multiplication (cooperative): %.3f ms, %.3e 512-bit products/s\n", ms, coop_rate);
  k_lane<<<blocks, 256>>>(in, o2); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); k_lane<<<blocks, 256>>>(in, o2); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  double lane_rate = (double)blocks * 256 * ITERS / (ms * 1e-3);
  printf("lane-per-multiplication (shipped mul_wide): %.3f ms, %.3e 512-bit products/s\n", ms, lane_rate);
  printf("ratio lane/coop = %.1fx\n", lane_rate / coop_rate);
  // cross-check: wave w of block 0 (element w) against lane w of block 0 for ITERS dependent steps
  std::vector<u32> r1(4 * 16), r2(256 * 16);
  CK(hipMemcpy(r1.data(), o1, r1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), o2, r2.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0; for (int w = 0; w < 4; ++w) for (int i = 0; i < 16; ++i) bad += r1[w * 16 + i] != r2[w * 16 + i];
  printf("cross-check of the two mappings on 4 operand pairs x %d dependent products: %s\n", ITERS, bad ? "MISMATCH" : "identical");
  return bad != 0;
}
