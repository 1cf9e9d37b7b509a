// l2_write.hip -- does the XCD's L2 keep rewritten lines, or does every store reach the fabric?
//
// The Ed25519 scheduler keeps each in-flight element's running result in its slot of the output array and rewrites it
// ~128 times; the PMC passes show every one of those stores leaving the L2 (15.4 GB of WRITE_SIZE per 2^20 batch,
// profiles/pmc_r03/ed25519-var) although the working set (3.4 MB per XCD) fits the 4 MB L2 and its re-READS do hit
// (2.0 GB fetched).  This rewrites a small region `passes` times with one store flavour per launch; run it under
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...      (and again with --pmc FETCH_SIZE)
// and compare the bytes per launch with region * passes (write-through) and region (write-back).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench/l2_write tools/microbench/l2_write.hip
//   tools/microbench/l2_write [region KiB per workgroup, default 8] [passes, default 100]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

// MODE 0 plain store, 1 nontemporal store, 2 sc1 (agent-scope, write-through by definition) store,
//      3 plain store after a plain load of the same 16 bytes (read-modify-write, as the scheduler does),
//      4 plain stores while the workgroup also streams through a large read-only buffer (do streaming loads evict the
//        dirty lines?), 5 the same with nontemporal streaming loads,
//      6 / 7 / 8: read-modify-write where a LANE owns a 128-byte line and one store instruction writes 16 bytes of it
//        (the scheduler's shape: eight instructions cover the line), or two / eight adjacent lanes share a line (32 bytes /
//        the whole line per instruction) -- at which store granularity does the L2 keep the dirty line?
template <int MODE>
__global__ __launch_bounds__(256) void k_rewrite(v4u* __restrict__ region, const v4u* __restrict__ stream, size_t stream_v4,
                                                 int per_wg_v4, int passes) {
  v4u* mine = region + (size_t)blockIdx.x * per_wg_v4;
  v4u acc = {blockIdx.x, threadIdx.x, 1u, 2u};
  if (MODE >= 6) {
    // per_wg_v4 / 8 lines per workgroup; lanes-per-line L = 1, 2 or 8; a wave instruction covers 64 / L lines
    constexpr int L = MODE == 6 ? 1 : (MODE == 7 ? 2 : 8);
    const int lines = per_wg_v4 / 8;
    for (int p = 0; p < passes; ++p) {
      for (int first = 0; first < lines; first += 256 / L) {
        const int line = first + threadIdx.x / L;
        if (line < lines) {
          for (int piece = threadIdx.x % L; piece < 8; piece += L) {   // 16-byte pieces of the line this lane writes
            v4u* q = mine + (size_t)line * 8 + piece;
            acc += *q;
            acc.x += (unsigned)p;
            *q = acc;
          }
        }
      }
      __syncthreads();
    }
    if (acc.x == 0xFFFFFFFFu && acc.y == 0xFFFFFFFFu) mine[0] = acc;
    return;
  }
  for (int p = 0; p < passes; ++p) {
    for (int i = threadIdx.x; i < per_wg_v4; i += 256) {
      if (MODE == 3) acc += mine[i];
      acc.x += (unsigned)p;
      if (MODE == 1) __builtin_nontemporal_store(acc, mine + i);
      else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(mine + i), "v"(acc) : "memory");
      else mine[i] = acc;
    }
    if (MODE == 4 || MODE == 5) {   // 64 KiB of streaming reads per pass and workgroup
      const size_t base = ((size_t)blockIdx.x * passes + p) * 4096 % (stream_v4 - 4096);
      for (int i = threadIdx.x; i < 4096; i += 256) acc += MODE == 5 ? __builtin_nontemporal_load(stream + base + i) : stream[base + i];
    }
    __syncthreads();   // (a workgroup barrier: s_waitcnt vmcnt(0) first, so a pass's stores have been issued to the L2)
  }
  if (acc.x == 0xFFFFFFFFu && acc.y == 0xFFFFFFFFu) mine[0] = acc;
}

int main(int argc, char** argv) {
  const int kib = argc > 1 ? atoi(argv[1]) : 8, passes = argc > 2 ? atoi(argv[2]) : 100;
  const int wgs = 256, per_wg_v4 = kib * 1024 / 16;
  const size_t region_bytes = (size_t)wgs * kib * 1024, stream_v4 = (size_t)256 << 20 >> 4;
  v4u *region, *stream;
  hipMalloc(&region, region_bytes);
  hipMalloc(&stream, stream_v4 * 16);
  hipMemset(region, 0, region_bytes);
  hipMemset(stream, 1, stream_v4 * 16);
  printf("region %.2f MB in all (%d KiB per workgroup, 256 workgroups), %d passes: write-through would be %.1f MB per launch, "
         "write-back %.2f MB\n", region_bytes / 1e6, kib, passes, region_bytes * (double)passes / 1e6, region_bytes / 1e6);
  const char* names[9] = {"plain", "nontemporal", "sc1", "plain after a load of the same bytes", "plain + plain streaming reads",
                          "plain + nontemporal streaming reads", "read-modify-write, a lane per line (16 B of a line per instruction)",
                          "read-modify-write, two lanes per line (32 B per instruction)", "read-modify-write, eight lanes per line (the whole line per instruction)"};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 9; ++mode) {
    hipEventRecord(e0, 0);
    switch (mode) {
      case 0: hipLaunchKernelGGL(k_rewrite<0>, dim3(wgs), dim3(256), 0, 0, region, stream, stream_v4, per_wg_v4, passes); break;
      case 1: hipLaunchKernelGGL(k_rewrite<1>, dim3(wgs), dim3(256), 0, 0, region, stream, stream_v4, per_wg_v4, passes); break;
      case 2: hipLaunchKernelGGL(k_rewrite<2>, dim3(wgs), dim3(256), 0, 0, region, stream, stream_v4, per_wg_v4, passes); break;
      case 3: hipLaunchKernelGGL(k_rewrite<3>, dim3(wgs), dim3(256), 0, 0, region, stream, stream_v4, per_wg_v4, passes); break;
      case 4: hipLaunchKernelGGL(k_rewrite<4>, dim3(wgs), dim3(256), 0, 0, region, stream, stream_v4, per_wg_v4, passes); break;
      case 5: hipLaunchKernelGGL(k_rewrite<5>, dim3(wgs), dim3(256), 0, 0, region, stream, stream_v4, per_wg_v4, passes); break;
      case 6: hipLaunchKernelGGL(k_rewrite<6>, dim3(wgs), dim3(256), 0, 0, region, stream, stream_v4, per_wg_v4, passes); break;
      case 7: hipLaunchKernelGGL(k_rewrite<7>, dim3(wgs), dim3(256), 0, 0, region, stream, stream_v4, per_wg_v4, passes); break;
      default: hipLaunchKernelGGL(k_rewrite<8>, dim3(wgs), dim3(256), 0, 0, region, stream, stream_v4, per_wg_v4, passes); break;
    }
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d (%s): %.3f ms\n", mode, names[mode], ms);
  }
  return 0;
}
