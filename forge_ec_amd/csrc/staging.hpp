// staging.hpp -- workgroup shape and the HBM <-> LDS staging helpers shared by all kernels.
#pragma once
#include "limbs.hpp"

namespace fecgpu {

constexpr int TPB = 256;  // threads per workgroup = 4 wavefronts, one per SIMD

// ------------------------------------------------------------------------------------------
// HBM <-> LDS staging: coalesced 16-byte accesses, word-major (transposed) LDS image
//   word w of the workgroup's element e lives at lds[w * TPB + e]
// ------------------------------------------------------------------------------------------
template <int W>
FEC_DEV void stage_in(u32* lds, const u32* g, int valid) {
  for (int v = threadIdx.x; v < TPB * W / 4; v += TPB) {
    int e = (v * 4) / W, w = (v * 4) % W;
    if (e < valid) {
      uint4 x = *reinterpret_cast<const uint4*>(g + (size_t)v * 4);
      lds[(w + 0) * TPB + e] = x.x;
      lds[(w + 1) * TPB + e] = x.y;
      lds[(w + 2) * TPB + e] = x.z;
      lds[(w + 3) * TPB + e] = x.w;
    }
  }
}
template <int W>
FEC_DEV void stage_out(u32* g, const u32* lds, int valid) {
  for (int v = threadIdx.x; v < TPB * W / 4; v += TPB) {
    int e = (v * 4) / W, w = (v * 4) % W;
    if (e < valid) {
      uint4 x;
      x.x = lds[(w + 0) * TPB + e];
      x.y = lds[(w + 1) * TPB + e];
      x.z = lds[(w + 2) * TPB + e];
      x.w = lds[(w + 3) * TPB + e];
      *reinterpret_cast<uint4*>(g + (size_t)v * 4) = x;
    }
  }
}
FEC_DEV fe load_fe(const u32* l, int stride) {
  fe a;
  FEC_UNROLL for (int i = 0; i < 8; ++i) a.w[i] = l[i * stride];
  return a;
}
FEC_DEV void store_fe(u32* l, int stride, const fe& a) {
  FEC_UNROLL for (int i = 0; i < 8; ++i) l[i * stride] = a.w[i];
}
FEC_DEV int block_valid(size_t n) {
  size_t first = (size_t)blockIdx.x * TPB;
  size_t left = n - first;
  return left < (size_t)TPB ? (int)left : TPB;
}

}  // namespace fecgpu
