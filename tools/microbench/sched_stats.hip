// sched_stats.hip -- where do the P-256 scheduler's instructions go?  Runs k_p256_mul_sched (the shipped source,
// compiled here with -DFEC_SCHED_STATS) on 2^20 random elements and prints how many batches of each kind ran, how full
// they were, how many took a rare leg, and how often waiting wavefronts polled.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFEC_SCHED_STATS -o tools/microbench/sched_stats tools/microbench/sched_stats.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../forge_ec_amd/csrc/kernels_p256.hip"

static unsigned long long sm(unsigned long long& s) {
  unsigned long long z = (s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)1 << atoi(argv[1]) : (size_t)1 << 20;
  std::vector<unsigned long long> k(n * 4), p(n * 12);
  unsigned long long s = 12345;
  for (auto& v : k) v = sm(s);
  // argv[2]: 1 = sparse scalars (2^255 + 1: 254 doublings, one addition), 2 = dense scalars (all ones: every step adds):
  // with rocprofv3 --pmc SQ_INSTS_VALU these give the DYNAMIC size of a doubling task and of an addition task
  const int pattern = argc > 2 ? atoi(argv[2]) : 0;
  if (pattern == 1) for (size_t i = 0; i < n; ++i) { k[4 * i] = 1; k[4 * i + 1] = k[4 * i + 2] = 0; k[4 * i + 3] = 1ull << 63; }
  if (pattern == 2) for (auto& v : k) v = ~0ull;
  for (auto& v : p) v = sm(s) >> 1;  // (any 256-bit coordinates do: multiply never validates its point)
  unsigned *dk, *dp, *dout;
  hipMalloc(&dk, n * 32);
  hipMalloc(&dp, n * 96);
  hipMalloc(&dout, n * 96);
  hipMemcpy(dk, k.data(), n * 32, hipMemcpyHostToDevice);
  hipMemcpy(dp, p.data(), n * 96, hipMemcpyHostToDevice);
  fecgpu::SchedEnv env;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  env.cus = prop.multiProcessorCount;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    unsigned long long zero[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(fecgpu::g_sched_stats), zero, sizeof zero);
    hipEventRecord(e0, 0);
    fecgpu::p256_launch_mul(env, false, dk, dp, dout, n, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(fecgpu::g_sched_stats), sizeof st);
    printf("n=%zu  %.3f ms (wave-local statistics counters)\n", n, ms);
    printf("doubling batches %llu, lanes %llu (%.2f per batch; %.2f doublings per element)\n", st[0], st[1],
           (double)st[1] / st[0], (double)st[1] / n);
    printf("addition batches %llu, lanes %llu (%.2f per batch; %.2f additions per element)\n", st[2], st[3],
           (double)st[3] / st[2], (double)st[3] / n);
    printf("claim tasks %llu (%.3f per 64 elements); sleeps + lost races of waiting wavefronts %llu (%.3f per batch)\n", st[7],
           (double)st[7] * 64.0 / n, st[6], (double)st[6] / (st[0] + st[2]));
  }
  return 0;
}
