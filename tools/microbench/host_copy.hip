// host_copy.hip -- what a host-pointer batch pays on the PCIe side, per way of moving caller memory:
//   pageable  : hipMemcpyAsync straight from / to malloc'ed memory (the runtime stages it)
//   register  : hipHostRegister the caller's range in place, DMA, hipHostUnregister
//   bounce    : CPU memcpy through a pinned buffer owned by the library, DMA
// for fresh (never touched) and warm destination buffers.  One GPU, one host thread unless stated.
//   hipcc --offload-arch=gfx950 -O2 -o host_copy host_copy.hip && ./host_copy
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

static void* fresh(size_t bytes) {  // untouched anonymous pages, like a large malloc / Vec::with_capacity
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  return p == MAP_FAILED ? nullptr : p;
}

static void par_memcpy(char* dst, const char* src, size_t bytes, int threads) {
  if (threads <= 1) {
    std::memcpy(dst, src, bytes);
    return;
  }
  std::vector<std::thread> w;
  const size_t per = (bytes / threads + 4095) & ~(size_t)4095;
  for (int t = 0; t < threads; ++t) {
    const size_t lo = per * t, hi = lo + per < bytes ? lo + per : bytes;
    if (lo >= hi) break;
    w.emplace_back([=] { std::memcpy(dst + lo, src + lo, hi - lo); });
  }
  for (auto& x : w) x.join();
}

int main() {
  hipStream_t s;
  CK(hipSetDevice(0));
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t sizes[] = {(size_t)8 << 20, (size_t)32 << 20, (size_t)128 << 20};
  void* dev = nullptr;
  CK(hipMalloc(&dev, sizes[2]));
  void* pinned = nullptr;
  CK(hipHostMalloc(&pinned, sizes[2], hipHostMallocDefault));
  std::memset(pinned, 1, sizes[2]);
  for (size_t bytes : sizes) {
    const double mb = bytes / 1e6;
    // ---- H2D from warm pageable memory (the caller filled its inputs)
    char* src = (char*)fresh(bytes);
    std::memset(src, 3, bytes);
    for (int rep = 0; rep < 2; ++rep) {
      double t0 = now();
      CK(hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, s));
      double t1 = now();
      CK(hipStreamSynchronize(s));
      double t2 = now();
      std::printf("H2D %6.0f MB pageable           : call %.2f ms, done %.2f ms, %.1f GB/s\n", mb, (t1 - t0) * 1e3, (t2 - t0) * 1e3, mb / (t2 - t0) / 1e3);
    }
    {
      double t0 = now();
      CK(hipHostRegister(src, bytes, hipHostRegisterDefault));
      double t1 = now();
      CK(hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, s));
      CK(hipStreamSynchronize(s));
      double t2 = now();
      CK(hipHostUnregister(src));
      double t3 = now();
      std::printf("H2D %6.0f MB register in place  : register %.2f ms, copy %.2f ms (%.1f GB/s), unregister %.2f ms, total %.2f ms\n", mb,
                  (t1 - t0) * 1e3, (t2 - t1) * 1e3, mb / (t2 - t1) / 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3);
    }
    for (int th : {1, 4}) {
      double t0 = now();
      par_memcpy((char*)pinned, src, bytes, th);
      double t1 = now();
      CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, s));
      CK(hipStreamSynchronize(s));
      double t2 = now();
      std::printf("H2D %6.0f MB bounce, %d thread(s) : memcpy %.2f ms (%.1f GB/s), DMA %.2f ms (%.1f GB/s)\n", mb, th, (t1 - t0) * 1e3,
                  mb / (t1 - t0) / 1e3, (t2 - t1) * 1e3, mb / (t2 - t1) / 1e3);
    }
    munmap(src, bytes);
    // ---- D2H into FRESH pageable memory (the caller allocated its output and never touched it)
    for (int mode = 0; mode < 4; ++mode) {
      char* dst = (char*)fresh(bytes);
      double t0 = now();
      if (mode == 0) {
        CK(hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t1 = now();
        std::printf("D2H %6.0f MB pageable, fresh    : %.2f ms, %.1f GB/s\n", mb, (t1 - t0) * 1e3, mb / (t1 - t0) / 1e3);
        t0 = now();
        CK(hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        t1 = now();
        std::printf("D2H %6.0f MB pageable, warm     : %.2f ms, %.1f GB/s\n", mb, (t1 - t0) * 1e3, mb / (t1 - t0) / 1e3);
      } else if (mode == 1) {
        CK(hipHostRegister(dst, bytes, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t2 = now();
        CK(hipHostUnregister(dst));
        double t3 = now();
        std::printf("D2H %6.0f MB register, fresh    : register %.2f ms, copy %.2f ms (%.1f GB/s), unregister %.2f ms, total %.2f ms\n", mb,
                    (t1 - t0) * 1e3, (t2 - t1) * 1e3, mb / (t2 - t1) / 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3);
      } else {
        const int th = mode == 2 ? 1 : 4;
        CK(hipMemcpyAsync(pinned, dev, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t1 = now();
        par_memcpy(dst, (const char*)pinned, bytes, th);
        double t2 = now();
        std::printf("D2H %6.0f MB bounce, %d thread(s), fresh : DMA %.2f ms (%.1f GB/s), memcpy %.2f ms (%.1f GB/s)\n", mb, th, (t1 - t0) * 1e3,
                    mb / (t1 - t0) / 1e3, (t2 - t1) * 1e3, mb / (t2 - t1) / 1e3);
      }
      munmap(dst, bytes);
    }
  }
  CK(hipFree(dev));
  CK(hipHostFree(pinned));
  return 0;
}
