// Cross-check + timing of the hand-allocated asm field multiplications (field_asm.inc) against the
// compiler-scheduled forms they replace, on the GPU.  Operands: random 256-bit values, an edge x edge
// grid, and operands built to force the rare continuations (borrow out of word 1, result >= p).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../forge_ec_amd/csrc -o asm_check asm_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "secp256k1.hpp"
#include "p256.hpp"
#include "ed25519.hpp"
using namespace fecgpu;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int OP>
__device__ fe run_new(const fe& a, const fe& b) {
  if (OP == 0) return secp::mul(a, b);
#ifdef HAVE_SECP_SQR
  if (OP == 1) return secp::sqr(a);
#endif
#ifdef HAVE_P256
  if (OP == 2) return p256::mul(a, b);
  if (OP == 3) return p256::sqr(a);
#endif
#ifdef HAVE_ED
  if (OP == 4) return ed::mul(a, b);
  if (OP == 5) return ed::sqr_exact(a);
#endif
  if (OP == 6) return secp::mul_small(a, 3);
  if (OP == 7) return secp::mul_small(a, 8);
  if (OP == 8) return p256::mul_small(a, 3);
  if (OP == 9) return p256::mul_small(a, 8);
  return a;
}
template <int OP>
__device__ fe run_old(const fe& a, const fe& b) {
  if (OP == 0) return secp::mul_cxx(a, b);
#ifdef HAVE_SECP_SQR
  if (OP == 1) return secp::sqr_cxx(a);
#endif
#ifdef HAVE_P256
  if (OP == 2) return p256::mul_cxx(a, b);
  if (OP == 3) return p256::sqr_cxx(a);
#endif
#ifdef HAVE_ED
  if (OP == 4) return ed::mul_cxx(a, b);
  if (OP == 5) return ed::sqr_cxx(a);
#endif
  if (OP == 6) return secp::mul_small_cxx(a, 3);
  if (OP == 7) return secp::mul_small_cxx(a, 8);
  if (OP == 8) return p256::mul_small_cxx(a, 3);
  if (OP == 9) return p256::mul_small_cxx(a, 8);
  return a;
}

template <int OP>
__global__ __launch_bounds__(256) void k_check(const u32* a, const u32* b, size_t n, unsigned* bad, u32* first_bad) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  fe x, y;
  for (int w = 0; w < 8; ++w) { x.w[w] = a[i * 8 + w]; y.w[w] = b[i * 8 + w]; }
  fe r1 = run_new<OP>(x, y), r0 = run_old<OP>(x, y);
  bool same = true;
  for (int w = 0; w < 8; ++w) same = same && r1.w[w] == r0.w[w];
  if (!same && atomicAdd(bad, 1u) == 0) {
    for (int w = 0; w < 8; ++w) { first_bad[w] = x.w[w]; first_bad[8 + w] = y.w[w]; first_bad[16 + w] = r0.w[w]; first_bad[24 + w] = r1.w[w]; }
  }
}

constexpr int ITERS = 2000;
template <int OP, bool NEW>
__global__ __launch_bounds__(256) void k_time(const u32* in, u32* out) {
  fe a, b;
  for (int i = 0; i < 8; ++i) { a.w[i] = in[threadIdx.x * 16 + i] ^ blockIdx.x; b.w[i] = in[threadIdx.x * 16 + 8 + i]; }
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) a = NEW ? run_new<OP>(a, b) : run_old<OP>(a, b);
  for (int i = 0; i < 8; ++i) out[(blockIdx.x * 256 + threadIdx.x) * 8 + i] = a.w[i];
}

static void mul_c_mod2_256(const u32 m[8], u32 out[8]) {  // out = m * (2^32 + 977) mod 2^256
  unsigned long long carry = 0;
  u32 t[8];
  for (int i = 0; i < 8; ++i) { unsigned long long p = (unsigned long long)m[i] * 977u + carry; t[i] = (u32)p; carry = p >> 32; }
  carry = 0;
  for (int i = 0; i < 8; ++i) { unsigned long long s = (unsigned long long)t[i] + (i ? m[i - 1] : 0) + carry; out[i] = (u32)s; carry = s >> 32; }
}

template <int OP>
static int check(const char* name, const std::vector<u32>& ha, const std::vector<u32>& hb) {
  size_t n = ha.size() / 8;
  u32 *da, *db, *dfb; unsigned* dbad;
  CK(hipMalloc(&da, ha.size() * 4)); CK(hipMalloc(&db, hb.size() * 4)); CK(hipMalloc(&dbad, 4)); CK(hipMalloc(&dfb, 32 * 4));
  CK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dbad, 0, 4));
  k_check<OP><<<(unsigned)((n + 255) / 256), 256>>>(da, db, n, dbad, dfb);
  CK(hipDeviceSynchronize());
  unsigned bad; u32 fb[32];
  CK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(fb, dfb, sizeof(fb), hipMemcpyDeviceToHost));
  printf("%-12s %zu operand pairs: %u mismatches\n", name, n, bad);
  if (bad) {
    const char* lab[4] = {"a", "b", "old", "new"};
    for (int k = 0; k < 4; ++k) { printf("  %s =", lab[k]); for (int w = 7; w >= 0; --w) printf(" %08x", fb[k * 8 + w]); printf("\n"); }
  }
  CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dbad)); CK(hipFree(dfb));
  return bad != 0;
}

template <int OP>
static void timeit(const char* name, int cus) {
  u32 *in, *out; CK(hipMalloc(&in, 256 * 16 * 4)); CK(hipMalloc(&out, (size_t)cus * 2 * 256 * 8 * 4));
  u32 h[256 * 16]; for (int i = 0; i < 256 * 16; ++i) h[i] = (u32)rand() * 2654435761u + i;
  CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nw = 0; nw < 2; ++nw) for (int wps : {1, 2}) {
    int blocks = cus * wps; float ms;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (nw) k_time<OP, true><<<blocks, 256>>>(in, out); else k_time<OP, false><<<blocks, 256>>>(in, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%-12s %s %d w/SIMD: %8.4f ms  SIMD-ns per op %.2f\n", name, nw ? "asm" : "c++", wps, ms, ms * 1e6 / ((double)ITERS * wps));
  }
  CK(hipFree(in)); CK(hipFree(out));
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  std::vector<u32> a, b;
  auto push = [&](const u32* x, const u32* y) { a.insert(a.end(), x, x + 8); b.insert(b.end(), y, y + 8); };
  // edge x edge grid
  std::vector<std::vector<u32>> edges;
  const u32 P_SECP[8] = {0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  const u32 P_P256[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 1, 0xFFFFFFFFu};
  const u32 P_ED[8] = {0xFFFFFFEDu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu};
  for (const u32* p : {P_SECP, P_P256, P_ED}) for (int d = -2; d <= 2; ++d) {
    std::vector<u32> e(p, p + 8);
    long long v = (long long)e[0] + d; e[0] = (u32)v;  // only word 0 (no ripple needed for these constants +-2 except ED/P256 +1,+2)
    if (v > 0xFFFFFFFFll) for (int w = 1; w < 8 && ++e[w] == 0; ++w) {}
    edges.push_back(e);
  }
  for (u32 k : {0u, 1u, 2u, 3u, 8u, 977u, 0xFFFFFFFFu}) { std::vector<u32> e(8, 0); e[0] = k; edges.push_back(e); }
  edges.push_back(std::vector<u32>(8, 0xFFFFFFFFu));
  for (int w = 0; w < 8; ++w) { std::vector<u32> e(8, 0); e[w] = 0xFFFFFFFFu; edges.push_back(e); e.assign(8, 0xFFFFFFFFu); e[w] = 0; edges.push_back(e); e.assign(8, 0); e[w] = 0x80000000u; edges.push_back(e); e[w] = 1; edges.push_back(e); }
  { std::vector<u32> e(8, 0); e[0] = 0x3D1; e[1] = 1; edges.push_back(e); }
  srand(12345);
  for (int r = 0; r < 24; ++r) { std::vector<u32> e(8); for (auto& x : e) x = (u32)rand() * 2654435761u ^ (u32)rand(); edges.push_back(e); }
  for (auto& x : edges) for (auto& y : edges) push(x.data(), y.data());
  // forced borrow out of word 1 (secp Mul): a = M*c mod 2^256 with the low words of M tiny and the top large, b = 1
  for (int r = 0; r < 4096; ++r) {
    u32 m[8], aa[8], one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    for (auto& x : m) x = (u32)rand() * 2654435761u ^ (u32)rand();
    m[0] = r & 7; m[1] = (r >> 3) & 1; m[7] |= 0xC0000000u;
    if (r & 16) { m[2] = m[3] = m[4] = m[5] = m[6] = 0; }
    mul_c_mod2_256(m, aa);
    push(aa, one);
    push(one, aa);
  }
  // random
  for (int r = 0; r < (1 << 18); ++r) {
    u32 x[8], y[8];
    for (auto& t : x) t = (u32)rand() * 2654435761u ^ ((u32)rand() << 7);
    for (auto& t : y) t = (u32)rand() * 2654435761u ^ ((u32)rand() << 7);
    if ((r & 15) == 0) x[7] = 0xFFFFFFFFu;
    if ((r & 31) == 0) y[7] = 0xFFFFFFFFu;
    if ((r & 63) == 0) for (int w = 2; w < 8; ++w) x[w] = 0xFFFFFFFFu;
    push(x, y);
  }
  int fail = 0;
  fail |= check<0>("secp mul", a, b);
#ifdef HAVE_SECP_SQR
  fail |= check<1>("secp sqr", a, b);
#endif
#ifdef HAVE_P256
  fail |= check<2>("p256 mul", a, b);
  fail |= check<3>("p256 sqr", a, b);
#endif
#ifdef HAVE_ED
  fail |= check<4>("ed mul", a, b);
  fail |= check<5>("ed sqr", a, b);
#endif
  fail |= check<6>("secp mul3", a, b);
  fail |= check<7>("secp mul8", a, b);
  fail |= check<8>("p256 mul3", a, b);
  fail |= check<9>("p256 mul8", a, b);
  timeit<0>("secp mul", cus);
  timeit<6>("secp mul3", cus);
  timeit<8>("p256 mul3", cus);
#ifdef HAVE_SECP_SQR
  timeit<1>("secp sqr", cus);
#endif
#ifdef HAVE_P256
  timeit<2>("p256 mul", cus);
  timeit<3>("p256 sqr", cus);
#endif
#ifdef HAVE_ED
  timeit<4>("ed mul", cus);
  timeit<5>("ed sqr", cus);
#endif
  printf(fail ? "FAIL\n" : "ALL OK\n");
  return fail;
}
