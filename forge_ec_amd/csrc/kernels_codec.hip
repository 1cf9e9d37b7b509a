// kernels_codec.hip -- point DECODING and the uncompressed encoding, exactly as
// the reference computes them:
//   k_decompress           PointAffine::from_bytes(&[u8; 33])   secp256k1.rs:896-976, p256.rs:1580-1639, ed25519.rs:1526-1582
//   k_encode_uncompressed  UncompressedPoint::from_affine        forge-ec-encoding/src/point.rs:186-211
//   k_decode_uncompressed  UncompressedPoint::to_affine          point.rs:214-281 (+ PointAffine::new of the curve)
// One element per lane.  The work is the reference's exponentiations (its own pow loops with its own,
// partly wrong, exponents): 256 squarings + the multiplications of the set exponent bits per sqrt.
#include <hip/hip_runtime.h>

#include "../../include/fecgpu.h"
#include "ed25519.hpp"
#include "p256.hpp"
#include "secp256k1.hpp"
#include "staging.hpp"
#include "kernels.hpp"

namespace fecgpu {

namespace {

struct CSecp {
  static constexpr bool BIG_ENDIAN_BYTES = true;
  FEC_DEV static lmask decompress(const fe& xv, lmask odd, fe& x, fe& y) { return secp::decompress(xv, odd, x, y); }
  FEC_DEV static lmask decode_uncompressed(const fe& xv, const fe& yv, fe& x, fe& y) { return secp::decode_uncompressed(xv, yv, x, y); }
  FEC_DEV static fe bytes_value(const fe& a) { return secp::mul(a, fe_small(1)); }  // to_bytes (138-178): mont_reduce
};
struct CP256 {
  static constexpr bool BIG_ENDIAN_BYTES = true;
  FEC_DEV static lmask decompress(const fe& xv, lmask odd, fe& x, fe& y) { return p256::decompress(xv, odd, x, y); }
  FEC_DEV static lmask decode_uncompressed(const fe& xv, const fe& yv, fe& x, fe& y) { return p256::decode_uncompressed(xv, yv, x, y); }
  FEC_DEV static fe bytes_value(const fe& a) { return a; }                           // to_bytes (288-300): raw limbs
};
struct CEd {
  static constexpr bool BIG_ENDIAN_BYTES = false;
  FEC_DEV static lmask decompress(const fe& xv, lmask odd, fe& x, fe& y) { return ed::decompress(xv, odd, x, y); }
  FEC_DEV static lmask decode_uncompressed(const fe& xv, const fe& yv, fe& x, fe& y) { return ed::decode_uncompressed(xv, yv, x, y); }
  FEC_DEV static fe bytes_value(const fe& a) { return ed::reduce(a); }                // to_bytes (295-310): reduce()
};

// the 32 bytes at `b` as a 256-bit value: big-endian (secp256k1, P-256) or little-endian (Ed25519)
template <bool BE>
FEC_DEV fe value_of(const unsigned char* b) {
  fe v;
  FEC_UNROLL for (int w = 0; w < 8; ++w) {
    u32 x = 0;
    FEC_UNROLL for (int j = 0; j < 4; ++j) {
      const int k = 4 * w + j;  // byte k of the value, little-endian index
      x |= (u32)b[BE ? 31 - k : k] << (8 * j);
    }
    v.w[w] = x;
  }
  return v;
}
template <bool BE>
FEC_DEV void bytes_of(unsigned char* b, const fe& v) {
  FEC_UNROLL for (int k = 0; k < 32; ++k) b[BE ? 31 - k : k] = (unsigned char)(v.w[k >> 2] >> (8 * (k & 3)));
}
FEC_DEV void put_xy(u32* xy, size_t i, const fe& x, const fe& y, bool ok) {
  FEC_UNROLL for (int w = 0; w < 8; ++w) {
    xy[i * 16 + w] = ok ? x.w[w] : 0u;
    xy[i * 16 + 8 + w] = ok ? y.w[w] : 0u;
  }
}

}  // namespace

template <class C>
__global__ __launch_bounds__(TPB) void k_decompress(const unsigned char* __restrict__ in, u32* __restrict__ xy,
                                                    unsigned char* __restrict__ inf, unsigned char* __restrict__ ok,
                                                    size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const unsigned char* b = in + i * 33;
  const unsigned char prefix = b[0];
  fe x, y;
  const lmask some = C::decompress(value_of<C::BIG_ENDIAN_BYTES>(b + 1), lanes_where(prefix == 0x03), x, y);
  const bool ident = prefix == 0x00;                       // Some(identity): x = y = 0, infinity
  const bool good = ident || ((prefix == 0x02 || prefix == 0x03) && lane_of(some));
  put_xy(xy, i, x, y, good && !ident);
  inf[i] = ident ? 1 : 0;
  ok[i] = good ? 1 : 0;
}

template <class C>
__global__ __launch_bounds__(TPB) void k_decode_uncompressed(const unsigned char* __restrict__ in, u32* __restrict__ xy,
                                                             unsigned char* __restrict__ inf, unsigned char* __restrict__ ok,
                                                             size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const unsigned char* b = in + i * 65;
  const unsigned char prefix = b[0];
  fe x, y;
  const lmask some = C::decode_uncompressed(value_of<C::BIG_ENDIAN_BYTES>(b + 1), value_of<C::BIG_ENDIAN_BYTES>(b + 33), x, y);
  const bool ident = prefix == 0x00;                       // 221-226: C::to_affine(&C::identity())
  const bool good = ident || (prefix == 0x04 && lane_of(some));
  put_xy(xy, i, x, y, good && !ident);
  inf[i] = ident ? 1 : 0;
  ok[i] = good ? 1 : 0;
}

template <class C>
__global__ __launch_bounds__(TPB) void k_encode_uncompressed(const u32* __restrict__ xy, const unsigned char* __restrict__ inf,
                                                             unsigned char* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  fe x, y;
  FEC_UNROLL for (int w = 0; w < 8; ++w) {
    x.w[w] = xy[i * 16 + w];
    y.w[w] = xy[i * 16 + 8 + w];
  }
  x = C::bytes_value(x);
  y = C::bytes_value(y);
  unsigned char* o = out + i * 65;
  const bool ident = inf != nullptr && inf[i] != 0;
  if (ident) {
    for (int k = 0; k < 65; ++k) o[k] = 0;
    return;
  }
  o[0] = 0x04;
  bytes_of<C::BIG_ENDIAN_BYTES>(o + 1, x);
  bytes_of<C::BIG_ENDIAN_BYTES>(o + 33, y);
}

void codec_launch(int op, int curve, const void* in, const void* in2, void* out, void* out2, void* out3, size_t n,
                  hipStream_t s) {
  const dim3 g((unsigned)((n + TPB - 1) / TPB)), b(TPB);
  const unsigned char* bi = static_cast<const unsigned char*>(in);
  if (op == 0 || op == 1) {
    u32* xy = static_cast<u32*>(out);
    unsigned char* inf = static_cast<unsigned char*>(out2);
    unsigned char* ok = static_cast<unsigned char*>(out3);
    if (op == 0) {
      if (curve == FEC_SECP256K1) hipLaunchKernelGGL((k_decompress<CSecp>), g, b, 0, s, bi, xy, inf, ok, n);
      else if (curve == FEC_P256) hipLaunchKernelGGL((k_decompress<CP256>), g, b, 0, s, bi, xy, inf, ok, n);
      else hipLaunchKernelGGL((k_decompress<CEd>), g, b, 0, s, bi, xy, inf, ok, n);
    } else {
      if (curve == FEC_SECP256K1) hipLaunchKernelGGL((k_decode_uncompressed<CSecp>), g, b, 0, s, bi, xy, inf, ok, n);
      else if (curve == FEC_P256) hipLaunchKernelGGL((k_decode_uncompressed<CP256>), g, b, 0, s, bi, xy, inf, ok, n);
      else hipLaunchKernelGGL((k_decode_uncompressed<CEd>), g, b, 0, s, bi, xy, inf, ok, n);
    }
  } else {
    const u32* xy = static_cast<const u32*>(in);
    const unsigned char* inf = static_cast<const unsigned char*>(in2);
    unsigned char* o = static_cast<unsigned char*>(out);
    if (curve == FEC_SECP256K1) hipLaunchKernelGGL((k_encode_uncompressed<CSecp>), g, b, 0, s, xy, inf, o, n);
    else if (curve == FEC_P256) hipLaunchKernelGGL((k_encode_uncompressed<CP256>), g, b, 0, s, xy, inf, o, n);
    else hipLaunchKernelGGL((k_encode_uncompressed<CEd>), g, b, 0, s, xy, inf, o, n);
  }
}

}  // namespace fecgpu
