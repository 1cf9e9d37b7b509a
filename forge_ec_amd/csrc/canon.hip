// canon.hip -- the CANONICAL-MATH MODE translation unit of libfecgpu.so (include/fecgpu_canon.h):
// kernels (canon_kernels.hpp), launch helpers and extern "C" entry points.  NOT reference parity.
#include <hip/hip_runtime.h>

#include "../../include/fecgpu.h"
#include "../../include/fecgpu_canon.h"
#include "host_ctx.hpp"
#include "canon_kernels.hpp"

using namespace fecgpu;
using namespace fecgpu::host;

namespace {

// ---- canonical-math mode -------------------------------------------------------------------
inline bool canon_curve_ok(int c) { return curve_ok(c); }

int ensure_canon_comb(fec_ctx* ctx, int curve, hipStream_t s) {
  if (ctx->canon_comb_ready[curve]) return FEC_OK;
  if (!ctx->d_canon_comb[curve] &&
      hipMalloc(&ctx->d_canon_comb[curve],
                (size_t)(curve == FEC_ED25519 ? canon::ED_COMB_WORDS : canon::COMB_WORDS) * sizeof(u32)) != hipSuccess) {
    (void)hipGetLastError();
    return FEC_E_OOM;
  }
  if (curve == FEC_SECP256K1) hipLaunchKernelGGL((k_canon_build_comb<csecp>), dim3(1), dim3(64), 0, s, ctx->d_canon_comb[curve]);
  else if (curve == FEC_P256) hipLaunchKernelGGL((k_canon_build_comb<cp256>), dim3(1), dim3(64), 0, s, ctx->d_canon_comb[curve]);
  else hipLaunchKernelGGL(k_ced_build_comb, dim3(1), dim3(64), 0, s, ctx->d_canon_comb[curve]);
  if (hipGetLastError() != hipSuccess) return FEC_E_LAUNCH;
  // the table is read by kernels on either pipeline stream: finish it before anyone can race
  if (hipStreamSynchronize(s) != hipSuccess) {
    (void)hipGetLastError();
    return FEC_E_LAUNCH;
  }
  ctx->canon_comb_ready[curve] = true;
  return FEC_OK;
}

int ensure_canon_comb8(fec_ctx* ctx, int curve, hipStream_t s) {
  if (ctx->canon_comb8_ready[curve]) return FEC_OK;
  if (!ctx->d_canon_comb8[curve] &&
      hipMalloc(&ctx->d_canon_comb8[curve],
                (curve == FEC_ED25519 ? canon::ED_COMB8_WORDS : canon::COMB8_WORDS) * sizeof(u32)) != hipSuccess) {
    (void)hipGetLastError();
    return FEC_E_OOM;
  }
  if (curve == FEC_SECP256K1) hipLaunchKernelGGL((k_canon_build_comb8<csecp>), dim3(canon::COMB8_WINDOWS), dim3(TPB), 0, s, ctx->d_canon_comb8[curve]);
  else if (curve == FEC_P256) hipLaunchKernelGGL((k_canon_build_comb8<cp256>), dim3(canon::COMB8_WINDOWS), dim3(TPB), 0, s, ctx->d_canon_comb8[curve]);
  else hipLaunchKernelGGL(k_ced_build_comb8, dim3(canon::COMB8_WINDOWS + 1), dim3(TPB), 0, s, ctx->d_canon_comb8[curve]);
  if (hipGetLastError() != hipSuccess) return FEC_E_LAUNCH;
  if (hipStreamSynchronize(s) != hipSuccess) {
    (void)hipGetLastError();
    return FEC_E_LAUNCH;
  }
  ctx->canon_comb8_ready[curve] = true;
  return FEC_OK;
}

// grow-only device buffer owned by the ctx (kernels of earlier calls may still use the old one)
int ensure_owned(void** buf, size_t* cap, size_t need) {
  if (*cap >= need) return FEC_OK;
  if (hipDeviceSynchronize() != hipSuccess) return FEC_E_LAUNCH;
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr;
  *cap = 0;
  if (hipMalloc(buf, need) != hipSuccess) {
    (void)hipGetLastError();
    return FEC_E_OOM;
  }
  *cap = need;
  return FEC_OK;
}

int launch_canon_normalize(fec_ctx* ctx, int curve, u64* dxy, unsigned char* dst, size_t n, hipStream_t s) {
  const size_t lanes = (n + canon::NORM_GROUP - 1) / canon::NORM_GROUP;
  const size_t stride = (lanes + 63) / 64 * 64;
  u32* xy = reinterpret_cast<u32*>(dxy);
  const u32* z = reinterpret_cast<const u32*>(ctx->d_zbuf);
  if (curve == FEC_SECP256K1) hipLaunchKernelGGL((k_canon_normalize<csecp>), dim3(grid_for(stride)), dim3(TPB), 0, s, xy, z, dst, n, stride);
  else if (curve == FEC_P256) hipLaunchKernelGGL((k_canon_normalize<cp256>), dim3(grid_for(stride)), dim3(TPB), 0, s, xy, z, dst, n, stride);
  else hipLaunchKernelGGL((k_canon_normalize<ced>), dim3(grid_for(stride)), dim3(TPB), 0, s, xy, z, dst, n, stride);
  return hipGetLastError() == hipSuccess ? FEC_OK : FEC_E_LAUNCH;
}

// phase 1 of k*G (comb).  finish = false leaves the projective result in dxy / zbuf (/ tbuf for
// Ed25519) for an accumulate pass; finish = true normalises to affine.
int launch_canon_mul_base(fec_ctx* ctx, int curve, const u64* ds, u64* dxy, unsigned char* dst, size_t n,
                          void* stream, bool finish = true) {
  if (n == 0) return FEC_OK;
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  const bool comb8 = ctx->canon_use_comb8;
  int rc = comb8 ? ensure_canon_comb8(ctx, curve, s) : ensure_canon_comb(ctx, curve, s);
  if (rc == FEC_OK) rc = ensure_owned(&ctx->d_zbuf, &ctx->zbuf_cap, n * 32);
  if (rc == FEC_OK && !finish && curve == FEC_ED25519) rc = ensure_owned(&ctx->d_tbuf, &ctx->tbuf_cap, n * 32);
  if (rc != FEC_OK) return rc;
  Launch L(ctx, stream, finish ? "k_canon_mul_base+k_canon_normalize" : "k_canon_mul_base");
  u32* tb = !finish && curve == FEC_ED25519 ? reinterpret_cast<u32*>(ctx->d_tbuf) : nullptr;
  const u32* k = reinterpret_cast<const u32*>(ds);
  u32* xy = reinterpret_cast<u32*>(dxy);
  u32* z = reinterpret_cast<u32*>(ctx->d_zbuf);
  if (comb8 && curve == FEC_SECP256K1) hipLaunchKernelGGL((k_canon_mul_base8<csecp>), dim3(grid_for(n)), dim3(TPB), 0, L.s, k, ctx->d_canon_comb8[curve], xy, z, dst, n);
  else if (comb8 && curve == FEC_P256) hipLaunchKernelGGL((k_canon_mul_base8<cp256>), dim3(grid_for(n)), dim3(TPB), 0, L.s, k, ctx->d_canon_comb8[curve], xy, z, dst, n);
  else if (comb8) hipLaunchKernelGGL(k_ced_mul_base8, dim3(grid_for(n)), dim3(TPB), 0, L.s, k, ctx->d_canon_comb8[curve], xy, z, tb, dst, n);
  else if (curve == FEC_SECP256K1) hipLaunchKernelGGL((k_canon_mul_base<csecp>), dim3(grid_for(n)), dim3(TPB), 0, L.s, k, ctx->d_canon_comb[curve], xy, z, dst, n);
  else if (curve == FEC_P256) hipLaunchKernelGGL((k_canon_mul_base<cp256>), dim3(grid_for(n)), dim3(TPB), 0, L.s, k, ctx->d_canon_comb[curve], xy, z, dst, n);
  else hipLaunchKernelGGL(k_ced_mul_base, dim3(grid_for(n)), dim3(TPB), 0, L.s, k, ctx->d_canon_comb[curve], xy, z, tb, dst, n);
  rc = finish ? launch_canon_normalize(ctx, curve, dxy, dst, n, L.s) : FEC_OK;
  int rc2 = L.done();
  return rc != FEC_OK ? rc : rc2;
}

// k*P by the windowed ladder; accum = true adds it onto the projective point already in dxy / zbuf
int launch_canon_mul(fec_ctx* ctx, int curve, const u64* ds, const u64* dp, u64* dxy, unsigned char* dst, size_t n,
                     void* stream, bool accum = false) {
  if (n == 0) return FEC_OK;
  int rc = ensure_owned(&ctx->d_win_scratch, &ctx->win_scratch_cap,
                        n * (size_t)(canon::WIN_ENTRIES * canon::WIN_ENTRY_WORDS) * sizeof(u32));
  if (rc == FEC_OK) rc = ensure_owned(&ctx->d_zbuf, &ctx->zbuf_cap, n * 32);
  if (rc != FEC_OK) return rc;
  Launch L(ctx, stream, accum ? "k_canon_mul<accum>+k_canon_normalize" : "k_canon_mul+k_canon_normalize");
  const u32* tb = accum && curve == FEC_ED25519 ? reinterpret_cast<const u32*>(ctx->d_tbuf) : nullptr;
  const u32* k = reinterpret_cast<const u32*>(ds);
  const u32* p = reinterpret_cast<const u32*>(dp);
  u32* scratch = reinterpret_cast<u32*>(ctx->d_win_scratch);
  u32* xy = reinterpret_cast<u32*>(dxy);
  u32* z = reinterpret_cast<u32*>(ctx->d_zbuf);
  dim3 g(grid_for(n)), b(TPB);
  if (curve == FEC_SECP256K1) {
    if (accum) hipLaunchKernelGGL((k_canon_mul<csecp, true>), g, b, 0, L.s, k, p, scratch, xy, z, dst, n);
    else hipLaunchKernelGGL((k_canon_mul<csecp, false>), g, b, 0, L.s, k, p, scratch, xy, z, dst, n);
  } else if (curve == FEC_P256) {
    if (accum) hipLaunchKernelGGL((k_canon_mul<cp256, true>), g, b, 0, L.s, k, p, scratch, xy, z, dst, n);
    else hipLaunchKernelGGL((k_canon_mul<cp256, false>), g, b, 0, L.s, k, p, scratch, xy, z, dst, n);
  } else {
    hipLaunchKernelGGL(k_ced_mul, g, b, 0, L.s, k, p, scratch, xy, z, tb, dst, n);
  }
  rc = launch_canon_normalize(ctx, curve, dxy, dst, n, L.s);
  int rc2 = L.done();
  return rc != FEC_OK ? rc : rc2;
}

// ECDSA verification: scalars -> u1*G + u2*Q -> compare.  d_work: u1, u2 (n*32 each), xy (n*64),
// point status (n), range flags (n).
int launch_canon_ecdsa_verify(fec_ctx* ctx, int curve, const u64* dz, const u64* dr, const u64* ds, const u64* dpk,
                              unsigned char* dres, size_t n, void* stream) {
  if (n == 0) return FEC_OK;
  const size_t need = n * (32 + 32 + 64 + 1 + 1) + 64;
  int rc = ensure_owned(&ctx->d_verify, &ctx->verify_cap, need);
  if (rc != FEC_OK) return rc;
  char* base = (char*)ctx->d_verify;
  u64* u1 = (u64*)base;
  u64* u2 = (u64*)(base + n * 32);
  u64* xy = (u64*)(base + n * 64);
  unsigned char* pst = (unsigned char*)(base + n * 128);
  unsigned char* ok = pst + n;
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  dim3 g(grid_for(n)), b(TPB);
  const size_t lanes = (n + canon::NORM_GROUP - 1) / canon::NORM_GROUP;
  const size_t stride = (lanes + 63) / 64 * 64;
  dim3 gs(grid_for(stride));
  if (curve == FEC_SECP256K1)
    hipLaunchKernelGGL((k_canon_ecdsa_scalars<canon::NSecp>), gs, b, 0, s, (const u32*)dz, (const u32*)dr, (const u32*)ds,
                       (u32*)u1, (u32*)u2, ok, n, stride);
  else
    hipLaunchKernelGGL((k_canon_ecdsa_scalars<canon::NP256>), gs, b, 0, s, (const u32*)dz, (const u32*)dr, (const u32*)ds,
                       (u32*)u1, (u32*)u2, ok, n, stride);
  if (hipGetLastError() != hipSuccess) return FEC_E_LAUNCH;
  rc = launch_canon_mul_base(ctx, curve, u1, xy, pst, n, stream, false);
  if (rc == FEC_OK) rc = launch_canon_mul(ctx, curve, u2, dpk, xy, pst, n, stream, true);
  if (rc != FEC_OK) return rc;
  if (curve == FEC_SECP256K1)
    hipLaunchKernelGGL((k_canon_ecdsa_finish<canon::NSecp>), g, b, 0, s, (const u32*)xy, (const u32*)dr, ok, pst, dres, n);
  else
    hipLaunchKernelGGL((k_canon_ecdsa_finish<canon::NP256>), g, b, 0, s, (const u32*)xy, (const u32*)dr, ok, pst, dres, n);
  return hipGetLastError() == hipSuccess ? FEC_OK : FEC_E_LAUNCH;
}

// BIP-340 / EdDSA: prepare -> u1*G + u2*P -> final test.  Work area: P xy (n*64), expected R xy (n*64,
// EdDSA only), u2 (n*32), result xy (n*64), point status (n), flags (n).
int launch_canon_sig_verify(fec_ctx* ctx, int curve, const u64* d_key, const u64* d_r, const u64* d_s, const u64* d_e,
                            unsigned char* dres, size_t n, void* stream) {
  if (n == 0) return FEC_OK;
  const size_t need = n * (64 + 64 + 32 + 64 + 1 + 1) + 64;
  int rc = ensure_owned(&ctx->d_verify, &ctx->verify_cap, need);
  if (rc != FEC_OK) return rc;
  char* base = (char*)ctx->d_verify;
  u64* pxy = (u64*)base;
  u64* rxy = (u64*)(base + n * 64);
  u64* u2 = (u64*)(base + n * 128);
  u64* xy = (u64*)(base + n * 160);
  unsigned char* pst = (unsigned char*)(base + n * 224);
  unsigned char* ok = pst + n;
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  dim3 g(grid_for(n)), b(TPB);
  if (curve == FEC_SECP256K1)
    hipLaunchKernelGGL(k_canon_bip340_prepare, g, b, 0, s, (const u32*)d_key, (const u32*)d_r, (const u32*)d_s,
                       (const u32*)d_e, (u32*)pxy, (u32*)u2, ok, n);
  else
    hipLaunchKernelGGL(k_ced_eddsa_prepare, g, b, 0, s, (const u32*)d_key, (const u32*)d_r, (const u32*)d_s,
                       (const u32*)d_e, (u32*)pxy, (u32*)rxy, (u32*)u2, ok, n);
  if (hipGetLastError() != hipSuccess) return FEC_E_LAUNCH;
  rc = launch_canon_mul_base(ctx, curve, d_s, xy, pst, n, stream, false);   // u1 = s
  if (rc == FEC_OK) rc = launch_canon_mul(ctx, curve, u2, pxy, xy, pst, n, stream, true);
  if (rc != FEC_OK) return rc;
  if (curve == FEC_SECP256K1)
    hipLaunchKernelGGL(k_canon_bip340_finish, g, b, 0, s, (const u32*)xy, (const u32*)d_r, ok, pst, dres, n);
  else
    hipLaunchKernelGGL(k_ced_eddsa_finish, g, b, 0, s, (const u32*)xy, (const u32*)rxy, ok, pst, dres, n);
  return hipGetLastError() == hipSuccess ? FEC_OK : FEC_E_LAUNCH;
}

}  // namespace

extern "C" {

// ---- canonical-math mode (include/fecgpu_canon.h): NOT reference parity ----------------------
int fec_canon_mul_base_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_scalars, uint64_t* d_out_xy,
                           uint8_t* d_status, size_t n, void* stream) try {
  FEC_FIRST_DEVICE(ctx);  // canonical mode is not sharded: a multi-device ctx runs it on devices[0]
  if (!ctx || (n && (!d_scalars || !d_out_xy || !d_status))) return FEC_E_ARG;
  if (!canon_curve_ok(curve)) return curve_ok(curve) ? FEC_E_UNSUPPORTED : FEC_E_ARG;
  if (!aligned16(d_scalars) || !aligned16(d_out_xy)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return launch_canon_mul_base(ctx, curve, d_scalars, d_out_xy, d_status, n, stream);
} FEC_ABI_CATCH_STATUS

int fec_canon_mul_base(fec_ctx* ctx, fec_curve curve, const uint64_t* scalars, uint64_t* out_xy, uint8_t* status,
                       size_t n) try {
  FEC_FIRST_DEVICE(ctx);  // canonical mode is not sharded: a multi-device ctx runs it on devices[0]
  if (!ctx || (n && (!scalars || !out_xy || !status))) return FEC_E_ARG;
  if (!canon_curve_ok(curve)) return curve_ok(curve) ? FEC_E_UNSUPPORTED : FEC_E_ARG;
  if (n == 0) return FEC_OK;
  const void* const in[4] = {scalars, nullptr, nullptr, nullptr};
  const size_t in_stride[4] = {32, 0, 0, 0};
  void* const out[2] = {out_xy, status};
  const size_t out_stride[2] = {64, 1};
  return host_chunked(ctx, n, in, in_stride, out, out_stride, [&](void* const d[4], void* const o[2], size_t cnt) {
    return launch_canon_mul_base(ctx, curve, (const u64*)d[0], (u64*)o[0], (unsigned char*)o[1], cnt, nullptr);
  });
} FEC_ABI_CATCH_STATUS

int fec_canon_mul_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_scalars, const uint64_t* d_points_xy,
                      uint64_t* d_out_xy, uint8_t* d_status, size_t n, void* stream) try {
  FEC_FIRST_DEVICE(ctx);  // canonical mode is not sharded: a multi-device ctx runs it on devices[0]
  if (!ctx || (n && (!d_scalars || !d_points_xy || !d_out_xy || !d_status))) return FEC_E_ARG;
  if (!canon_curve_ok(curve)) return curve_ok(curve) ? FEC_E_UNSUPPORTED : FEC_E_ARG;
  if (!aligned16(d_scalars) || !aligned16(d_points_xy) || !aligned16(d_out_xy)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return launch_canon_mul(ctx, curve, d_scalars, d_points_xy, d_out_xy, d_status, n, stream);
} FEC_ABI_CATCH_STATUS

int fec_canon_mul(fec_ctx* ctx, fec_curve curve, const uint64_t* scalars, const uint64_t* points_xy,
                  uint64_t* out_xy, uint8_t* status, size_t n) try {
  FEC_FIRST_DEVICE(ctx);  // canonical mode is not sharded: a multi-device ctx runs it on devices[0]
  if (!ctx || (n && (!scalars || !points_xy || !out_xy || !status))) return FEC_E_ARG;
  if (!canon_curve_ok(curve)) return curve_ok(curve) ? FEC_E_UNSUPPORTED : FEC_E_ARG;
  if (n == 0) return FEC_OK;
  const void* const in[4] = {scalars, points_xy, nullptr, nullptr};
  const size_t in_stride[4] = {32, 64, 0, 0};
  void* const out[2] = {out_xy, status};
  const size_t out_stride[2] = {64, 1};
  return host_chunked(ctx, n, in, in_stride, out, out_stride, [&](void* const d[4], void* const o[2], size_t cnt) {
    return launch_canon_mul(ctx, curve, (const u64*)d[0], (const u64*)d[1], (u64*)o[0], (unsigned char*)o[1], cnt, nullptr);
  });
} FEC_ABI_CATCH_STATUS

int fec_canon_double_mul_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_u1, const uint64_t* d_u2,
                             const uint64_t* d_points_xy, uint64_t* d_out_xy, uint8_t* d_status, size_t n,
                             void* stream) try {
  FEC_FIRST_DEVICE(ctx);  // canonical mode is not sharded: a multi-device ctx runs it on devices[0]
  if (!ctx || (n && (!d_u1 || !d_u2 || !d_points_xy || !d_out_xy || !d_status))) return FEC_E_ARG;
  if (!canon_curve_ok(curve)) return FEC_E_ARG;
  if (!aligned16(d_u1) || !aligned16(d_u2) || !aligned16(d_points_xy) || !aligned16(d_out_xy)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  int rc = launch_canon_mul_base(ctx, curve, d_u1, d_out_xy, d_status, n, stream, false);
  if (rc != FEC_OK) return rc;
  return launch_canon_mul(ctx, curve, d_u2, d_points_xy, d_out_xy, d_status, n, stream, true);
} FEC_ABI_CATCH_STATUS

int fec_canon_double_mul(fec_ctx* ctx, fec_curve curve, const uint64_t* u1, const uint64_t* u2,
                         const uint64_t* points_xy, uint64_t* out_xy, uint8_t* status, size_t n) try {
  FEC_FIRST_DEVICE(ctx);  // canonical mode is not sharded: a multi-device ctx runs it on devices[0]
  if (!ctx || (n && (!u1 || !u2 || !points_xy || !out_xy || !status))) return FEC_E_ARG;
  if (!canon_curve_ok(curve)) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  const void* const in[4] = {u1, u2, points_xy, nullptr};
  const size_t in_stride[4] = {32, 32, 64, 0};
  void* const out[2] = {out_xy, status};
  const size_t out_stride[2] = {64, 1};
  return host_chunked(ctx, n, in, in_stride, out, out_stride, [&](void* const d[4], void* const o[2], size_t cnt) {
    int rc = launch_canon_mul_base(ctx, curve, (const u64*)d[0], (u64*)o[0], (unsigned char*)o[1], cnt, nullptr, false);
    if (rc != FEC_OK) return rc;
    return launch_canon_mul(ctx, curve, (const u64*)d[1], (const u64*)d[2], (u64*)o[0], (unsigned char*)o[1], cnt, nullptr,
                            true);
  });
} FEC_ABI_CATCH_STATUS

int fec_canon_ecdsa_verify_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_z, const uint64_t* d_r,
                               const uint64_t* d_s, const uint64_t* d_pk_xy, uint8_t* d_result, size_t n,
                               void* stream) try {
  FEC_FIRST_DEVICE(ctx);  // canonical mode is not sharded: a multi-device ctx runs it on devices[0]
  if (!ctx || (n && (!d_z || !d_r || !d_s || !d_pk_xy || !d_result))) return FEC_E_ARG;
  if (curve != FEC_SECP256K1 && curve != FEC_P256) return curve_ok(curve) ? FEC_E_UNSUPPORTED : FEC_E_ARG;
  if (!aligned16(d_z) || !aligned16(d_r) || !aligned16(d_s) || !aligned16(d_pk_xy)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return launch_canon_ecdsa_verify(ctx, curve, d_z, d_r, d_s, d_pk_xy, d_result, n, stream);
} FEC_ABI_CATCH_STATUS

int fec_canon_ecdsa_verify(fec_ctx* ctx, fec_curve curve, const uint64_t* z, const uint64_t* r, const uint64_t* s,
                           const uint64_t* pk_xy, uint8_t* result, size_t n) try {
  FEC_FIRST_DEVICE(ctx);  // canonical mode is not sharded: a multi-device ctx runs it on devices[0]
  if (!ctx || (n && (!z || !r || !s || !pk_xy || !result))) return FEC_E_ARG;
  if (curve != FEC_SECP256K1 && curve != FEC_P256) return curve_ok(curve) ? FEC_E_UNSUPPORTED : FEC_E_ARG;
  if (n == 0) return FEC_OK;
  const void* const in[4] = {z, r, s, pk_xy};
  const size_t in_stride[4] = {32, 32, 32, 64};
  void* const out[2] = {result, nullptr};
  const size_t out_stride[2] = {1, 0};
  return host_chunked(ctx, n, in, in_stride, out, out_stride, [&](void* const d[4], void* const o[2], size_t cnt) {
    return launch_canon_ecdsa_verify(ctx, curve, (const u64*)d[0], (const u64*)d[1], (const u64*)d[2], (const u64*)d[3],
                                     (unsigned char*)o[0], cnt, nullptr);
  });
} FEC_ABI_CATCH_STATUS

int fec_canon_bip340_verify_dev(fec_ctx* ctx, const uint64_t* d_pk_x, const uint64_t* d_r, const uint64_t* d_s,
                                const uint64_t* d_e, uint8_t* d_result, size_t n, void* stream) try {
  FEC_FIRST_DEVICE(ctx);  // canonical mode is not sharded: a multi-device ctx runs it on devices[0]
  if (!ctx || (n && (!d_pk_x || !d_r || !d_s || !d_e || !d_result))) return FEC_E_ARG;
  if (!aligned16(d_pk_x) || !aligned16(d_r) || !aligned16(d_s) || !aligned16(d_e)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return launch_canon_sig_verify(ctx, FEC_SECP256K1, d_pk_x, d_r, d_s, d_e, d_result, n, stream);
} FEC_ABI_CATCH_STATUS

int fec_canon_bip340_verify(fec_ctx* ctx, const uint64_t* pk_x, const uint64_t* r, const uint64_t* s,
                            const uint64_t* e, uint8_t* result, size_t n) try {
  FEC_FIRST_DEVICE(ctx);  // canonical mode is not sharded: a multi-device ctx runs it on devices[0]
  if (!ctx || (n && (!pk_x || !r || !s || !e || !result))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  const void* const in[4] = {pk_x, r, s, e};
  const size_t in_stride[4] = {32, 32, 32, 32};
  void* const out[2] = {result, nullptr};
  const size_t out_stride[2] = {1, 0};
  return host_chunked(ctx, n, in, in_stride, out, out_stride, [&](void* const d[4], void* const o[2], size_t cnt) {
    return launch_canon_sig_verify(ctx, FEC_SECP256K1, (const u64*)d[0], (const u64*)d[1], (const u64*)d[2], (const u64*)d[3],
                                   (unsigned char*)o[0], cnt, nullptr);
  });
} FEC_ABI_CATCH_STATUS

int fec_canon_eddsa_verify_dev(fec_ctx* ctx, const uint64_t* d_a_enc, const uint64_t* d_r_enc, const uint64_t* d_s,
                               const uint64_t* d_h, uint8_t* d_result, size_t n, void* stream) try {
  FEC_FIRST_DEVICE(ctx);  // canonical mode is not sharded: a multi-device ctx runs it on devices[0]
  if (!ctx || (n && (!d_a_enc || !d_r_enc || !d_s || !d_h || !d_result))) return FEC_E_ARG;
  if (!aligned16(d_a_enc) || !aligned16(d_r_enc) || !aligned16(d_s) || !aligned16(d_h)) return FEC_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return FEC_E_DEVICE;
  return launch_canon_sig_verify(ctx, FEC_ED25519, d_a_enc, d_r_enc, d_s, d_h, d_result, n, stream);
} FEC_ABI_CATCH_STATUS

int fec_canon_eddsa_verify(fec_ctx* ctx, const uint64_t* a_enc, const uint64_t* r_enc, const uint64_t* s,
                           const uint64_t* h, uint8_t* result, size_t n) try {
  FEC_FIRST_DEVICE(ctx);  // canonical mode is not sharded: a multi-device ctx runs it on devices[0]
  if (!ctx || (n && (!a_enc || !r_enc || !s || !h || !result))) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  const void* const in[4] = {a_enc, r_enc, s, h};
  const size_t in_stride[4] = {32, 32, 32, 32};
  void* const out[2] = {result, nullptr};
  const size_t out_stride[2] = {1, 0};
  return host_chunked(ctx, n, in, in_stride, out, out_stride, [&](void* const d[4], void* const o[2], size_t cnt) {
    return launch_canon_sig_verify(ctx, FEC_ED25519, (const u64*)d[0], (const u64*)d[1], (const u64*)d[2], (const u64*)d[3],
                                   (unsigned char*)o[0], cnt, nullptr);
  });
} FEC_ABI_CATCH_STATUS

int fec_canon_scalar_op(fec_ctx* ctx, fec_curve curve, int op, const uint64_t* a, const uint64_t* b,
                        const uint64_t* c, uint64_t* out, size_t n) try {
  FEC_FIRST_DEVICE(ctx);  // canonical mode is not sharded: a multi-device ctx runs it on devices[0]
  if (!ctx || !canon_curve_ok(curve) || op < 0 || op > 1 || (n && (!a || !out))) return FEC_E_ARG;
  if (op == 0 && n && (!b || !c)) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  const void* const in[4] = {a, op == 0 ? b : nullptr, op == 0 ? c : nullptr, nullptr};
  const size_t in_stride[4] = {32, 32, 32, 0};
  void* const outs[2] = {out, nullptr};
  const size_t out_stride[2] = {32, 0};
  return host_chunked(ctx, n, in, in_stride, outs, out_stride, [&](void* const d[4], void* const o[2], size_t cnt) {
    Launch L(ctx, nullptr, "k_canon_scalar_op");
    dim3 g(grid_for(cnt)), blk(TPB);
    const u32 *x = (const u32*)d[0], *y = (const u32*)d[1], *z = (const u32*)d[2];
    if (curve == FEC_SECP256K1) hipLaunchKernelGGL((k_canon_scalar_op<canon::NSecp>), g, blk, 0, L.s, op, x, y, z, (u32*)o[0], cnt);
    else if (curve == FEC_P256) hipLaunchKernelGGL((k_canon_scalar_op<canon::NP256>), g, blk, 0, L.s, op, x, y, z, (u32*)o[0], cnt);
    else hipLaunchKernelGGL((k_canon_scalar_op<canon::NEd>), g, blk, 0, L.s, op, x, y, z, (u32*)o[0], cnt);
    return L.done();
  });
} FEC_ABI_CATCH_STATUS

int fec_canon_field_op(fec_ctx* ctx, fec_curve curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* out,
                       size_t n) try {
  FEC_FIRST_DEVICE(ctx);  // canonical mode is not sharded: a multi-device ctx runs it on devices[0]
  if (!ctx || op < FEC_F_ADD || op > FEC_F_INV || (n && (!a || !out))) return FEC_E_ARG;
  if (!canon_curve_ok(curve)) return curve_ok(curve) ? FEC_E_UNSUPPORTED : FEC_E_ARG;
  const bool binary = op == FEC_F_ADD || op == FEC_F_SUB || op == FEC_F_MUL;
  if (binary && n && !b) return FEC_E_ARG;
  if (n == 0) return FEC_OK;
  const void* const in[4] = {a, binary ? b : nullptr, nullptr, nullptr};
  const size_t in_stride[4] = {32, 32, 0, 0};
  void* const outs[2] = {out, nullptr};
  const size_t out_stride[2] = {32, 0};
  return host_chunked(ctx, n, in, in_stride, outs, out_stride, [&](void* const d[4], void* const oo[2], size_t cnt) {
    const size_t n = cnt;
    void *x = d[0], *y = d[1], *o = oo[0];
    Launch L(ctx, nullptr, "k_canon_field_op");
    if (curve == FEC_SECP256K1) hipLaunchKernelGGL((k_canon_field_op<csecp>), dim3(grid_for(n)), dim3(TPB), 0, L.s, op, (const u32*)x, (const u32*)y, (u32*)o, n);
    else if (curve == FEC_P256) hipLaunchKernelGGL((k_canon_field_op<cp256>), dim3(grid_for(n)), dim3(TPB), 0, L.s, op, (const u32*)x, (const u32*)y, (u32*)o, n);
    else hipLaunchKernelGGL((k_canon_field_op<ced>), dim3(grid_for(n)), dim3(TPB), 0, L.s, op, (const u32*)x, (const u32*)y, (u32*)o, n);
    return L.done();
  });
} FEC_ABI_CATCH_STATUS

}  // extern "C"
