// Cycles per field operation of the shipped device headers, in isolation: each lane runs a
// dependent chain of one operation; 1, 2 waves per SIMD.  Guides kernel-level optimisation.
// build: hipcc --offload-arch=gfx950 -O3 -I../../forge_ec_amd/csrc -o field_ops field_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "secp256k1.hpp"
#include "p256.hpp"
#include "ed25519.hpp"
#include "canon_curves.hpp"
using namespace fecgpu;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITERS = 2000;

__device__ fe load(const u32* in, int k) { fe a; for (int i = 0; i < 8; ++i) a.w[i] = in[(threadIdx.x * 4 + k) * 8 + i] ^ blockIdx.x; return a; }
__device__ void store(u32* out, const fe& a) { for (int i = 0; i < 8; ++i) out[(blockIdx.x * 256 + threadIdx.x) * 8 + i] = a.w[i]; }

#define KERN(name, BODY) __global__ __launch_bounds__(256) void name(const u32* in, u32* out) { \
  fe a = load(in, 0), b = load(in, 1); \
  _Pragma("unroll 1") for (int it = 0; it < ITERS; ++it) { BODY } store(out, a); }

KERN(k_secp_mul, a = secp::mul(a, b);)
KERN(k_secp_sqr, a = secp::sqr(a);)
KERN(k_secp_add, a = secp::add(a, b);)
KERN(k_secp_sub, a = secp::sub(a, b);)
KERN(k_secp_mul_small, a = secp::mul_small(a, 3);)
KERN(k_secp_mul2, a = secp::mul(a, b); b = secp::mul(b, a);)
KERN(k_p256_mul, a = p256::mul(a, b);)
KERN(k_p256_add, a = p256::add(a, b);)
KERN(k_p256_sub, a = p256::sub(a, b);)
KERN(k_ed_mul, a = ed::mul(a, b);)
KERN(k_ed_add, a = ed::add(a, b);)
KERN(k_ed_sub, a = ed::sub(a, b);)
KERN(k_csecp_mul, a = csecp::mul(a, b);)
KERN(k_csecp_sqr, a = csecp::sqr(a);)
KERN(k_cp256_mul, a = cp256::mul(a, b);)
KERN(k_cp256_sqr, a = cp256::sqr(a);)
KERN(k_ced_mul, a = ced::mul(a, b);)
KERN(k_ced_sqr, a = ced::sqr(a);)
KERN(k_nsecp_mmul, a = canon::Fn<canon::NSecp>::mmul(a, b);)
KERN(k_select, a = fe_select(a, b, lanes_where((a.w[0] & 1) != 0)); b.w[0] += a.w[1];)
__global__ __launch_bounds__(256) void k_mulwide(const u32* in, u32* out) {
  fe a = load(in, 0), b = load(in, 1);
  _Pragma("unroll 1") for (int it = 0; it < ITERS; ++it) { u32 t[16]; mul_wide(t, a, b); for (int i = 0; i < 8; ++i) a.w[i] = t[i] ^ t[i + 8]; }
  store(out, a); }
__global__ __launch_bounds__(256) void k_secp_padd(const u32* in, u32* out) {
  secp::pt p, q; p.x = load(in, 0); p.y = load(in, 1); p.z = load(in, 2); q.x = load(in, 3); q.y = load(in, 0); q.z = load(in, 1);
  _Pragma("unroll 1") for (int it = 0; it < ITERS / 10; ++it) { lmask nd; p = secp::padd_nodouble(p, q, nd); }
  store(out, p.x); }
__global__ __launch_bounds__(256) void k_secp_pdouble(const u32* in, u32* out) {
  secp::pt p; p.x = load(in, 0); p.y = load(in, 1); p.z = load(in, 2);
  _Pragma("unroll 1") for (int it = 0; it < ITERS / 10; ++it) { p = secp::pdouble(p); }
  store(out, p.x); }

typedef void (*kern_t)(const u32*, u32*);
struct Case { const char* name; kern_t k; double ops; };
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  u32 *in, *out; CK(hipMalloc(&in, 256 * 4 * 8 * 4)); CK(hipMalloc(&out, (size_t)cus * 2 * 256 * 8 * 4));
  u32 h[256 * 32]; for (int i = 0; i < 256 * 32; ++i) h[i] = (u32)rand() * 2654435761u + i; CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  Case cases[] = {{"mul_wide (512-bit product)", k_mulwide, ITERS}, {"secp mul", k_secp_mul, ITERS}, {"secp mul x2 interleaved", k_secp_mul2, 2.0 * ITERS},
    {"secp sqr", k_secp_sqr, ITERS}, {"secp mul_small", k_secp_mul_small, ITERS}, {"secp add", k_secp_add, ITERS}, {"secp sub", k_secp_sub, ITERS},
    {"p256 mul", k_p256_mul, ITERS}, {"p256 add", k_p256_add, ITERS}, {"p256 sub", k_p256_sub, ITERS},
    {"ed mul", k_ed_mul, ITERS}, {"ed add", k_ed_add, ITERS}, {"ed sub", k_ed_sub, ITERS}, {"fe_select", k_select, ITERS}, {"canon secp mul", k_csecp_mul, ITERS}, {"canon secp sqr", k_csecp_sqr, ITERS}, {"canon p256 mul", k_cp256_mul, ITERS}, {"canon p256 sqr", k_cp256_sqr, ITERS}, {"canon ed25519 mul", k_ced_mul, ITERS}, {"canon ed25519 sqr", k_ced_sqr, ITERS}, {"mont mul mod n (secp256k1)", k_nsecp_mmul, ITERS},
    {"secp padd_nodouble", k_secp_padd, ITERS / 10}, {"secp pdouble", k_secp_pdouble, ITERS / 10}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-30s %6s %10s %16s\n", "op", "w/SIMD", "ms", "ns/op per wave");
  for (auto& c : cases) for (int wps : {1, 2}) {
    int blocks = cus * wps;
    c.k<<<blocks, 256>>>(in, out); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); c.k<<<blocks, 256>>>(in, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: wps waves each doing c.ops ops in ms -> SIMD time per op = ms / (wps * ops)
    printf("%-30s %6d %10.4f %16.2f  (SIMD-ns per op %.2f)\n", c.name, wps, ms, ms * 1e6 / c.ops, ms * 1e6 / (c.ops * wps));
  }
  return 0;
}
