// kernels.hpp -- launchers of kernels that live in their own translation units (compiled in parallel).
#pragma once
#include <hip/hip_runtime.h>

#include "limbs.hpp"

namespace fecgpu {

// What a launch of one of the persistent scheduler kernels needs from the ctx it runs for.
//   err         the ctx's device-visible error word (pinned host memory mapped into the device): a scheduler whose
//               watchdog fires stores its FEC_DEVERR_* code there besides zero-filling its outputs, and every host-pointer entry
//               point reads it after its final synchronisation (fec_ctx_check for the *_dev callers) -- a scheduler
//               fault therefore surfaces as FEC_E_LAUNCH, never as FEC_OK with zeroed points.
//   cus         CU count of the ctx's OWN device (one persistent workgroup per CU)
//   force_fault debug hook (fec_ctx_debug_force_fault): the kernels raise their error word at once, so that the
//               whole error path can be exercised by a test
//   gen, gen_prefix, gen_prefix_bits   per curve: the device address of the reference's generator() and its fixed-base
//               prefix table (the state of multiply(G, k) after the first gen_prefix_bits steps, for every pattern of
//               those bits; null / 0 = none).  A fixed-base launch whose base IS that address starts from the table.
struct SchedEnv {
  unsigned* err = nullptr;
  unsigned cus = 256;
  unsigned force_fault = 0;
  const u32* gen[3] = {nullptr, nullptr, nullptr};
  const u32* gen_prefix[3] = {nullptr, nullptr, nullptr};
  unsigned gen_prefix_bits[3] = {0, 0, 0};
};
enum : unsigned { FEC_DEVERR_SCHED_WATCHDOG = 1u, FEC_DEVERR_SCHED_INDEX = 2u, FEC_DEVERR_FORCED = 4u };

// kernels_p256.hip: P-256 Curve::multiply, workgroup task scheduler.  out[i] = multiply(fixed ? points[0] : points[i], scalars[i])
// One persistent workgroup per CU the launch may take: env.cus (two launches that run side by side get a SchedEnv each,
// cu_split.hpp).
void p256_launch_mul(const SchedEnv& env, bool fixed, const u32* scalars, const u32* points, u32* out, size_t n, hipStream_t s);

// kernels_ed.hip: Ed25519 variable-base Curve::multiply, persistent workgroup task scheduler (one workgroup
// per CU, element state in LDS, slots refilled from the workgroup's range).
void ed_launch_mul(const SchedEnv& env, const u32* scalars, const u32* points, u32* out, size_t n, hipStream_t s,
                   unsigned cu_divisor = 1);
// kernels_ed.hip: Ed25519 fixed-base multiply from the 256-entry addend table of `base` (table[j] = 2^j * base by
// the reference's own doubling chain, built once per base by ed_build_table_launch; 256 * 32 words).
void ed_build_table_launch(const u32* base, u32* table, hipStream_t s);
// `work`: ed_fixed_work_bytes(n) bytes of device scratch owned by the launch's stream (0 bytes / null for small
// batches: the batch-wide popcount sort pays from 2^16 elements on).
size_t ed_fixed_work_bytes(size_t n);
void ed_fixed_launch(const SchedEnv& env, const u32* scalars, const u32* base, const u32* table, u32* out, size_t n, void* work,
                     hipStream_t s);

// kernels_secp.hip: secp256k1 Curve::multiply, lane-per-element ladder at three wavefronts per SIMD.
void secp_launch_mul(const SchedEnv& env, bool fixed, const u32* scalars, const u32* points, u32* out, size_t n, hipStream_t s);
// One level of the fixed-base prefix table: child[g] (48 words: r0, r1) = one ladder step from parent[g >> 1] with the
// bit g & 1; level 0 is the single entry (identity, base)
void secp_prefix_level_launch(const u32* parent, u32* child, size_t child_entries, hipStream_t s);

// kernels_codec.hip: op 0 = PointAffine::from_bytes (in: n*33 bytes -> out xy, out2 inf, out3 ok),
// op 1 = UncompressedPoint::to_affine (in: n*65 bytes -> xy, inf, ok), op 2 = UncompressedPoint::from_affine
// (in xy, in2 inf or null -> out n*65 bytes)
void codec_launch(int op, int curve, const void* in, const void* in2, void* out, void* out2, void* out3, size_t n,
                  hipStream_t s);

// kernels_ecdsa.hip: Ecdsa::<C, D>::verify for C = Secp256k1 / P256 (ecdsa.rs:213-281): scalar pre-pass, the two
// multiplications through the launchers above, finishing pass.  `work` holds ecdsa_work_bytes(n) bytes.
size_t ecdsa_work_bytes(size_t n);
void ecdsa_launch(const SchedEnv& env, int curve, const unsigned char* digests, const u32* r, const u32* s_, const u32* pk,
                  const unsigned char* pk_inf, const u32* gen, unsigned char* status, void* work, size_t n,
                  hipStream_t s, hipStream_t side = nullptr);   // side: a second stream for the fixed-base launch, or null

// kernels_ecdsa.hip: Ecdsa::<C, D>::batch_verify (ecdsa.rs:287-391) in three parts around the point fold that
// fecgpu.hip owns.  Work area (ecdsa_batch_work_bytes): u1 +0, u2 +32n, Q +64n, ta +160n, tb +256n, flags +352n
// (one byte per signature: 0 go, 1 the loop returns false here, 2 it panics here), a_i * r_i after that.
size_t ecdsa_batch_work_bytes(size_t n);
void ecdsa_batch_pre_launch(int curve, const unsigned char* digests, const u32* r, const u32* s_, const u32* pk,
                            const unsigned char* pk_inf, const u32* weights, void* work, size_t n, hipStream_t s);
void ecdsa_batch_mul_launch(const SchedEnv& env, int curve, const u32* gen, void* work, size_t n, hipStream_t s, hipStream_t side);
void ecdsa_batch_finish_launch(int curve, const u32* r_sum, const void* work, size_t n, unsigned char* result, u32* detail,
                               hipStream_t s);

// kernels_ecdsa.hip: Curve::validate_point per affine point (secp256k1 / P-256: is_on_curve; Ed25519: the trait default
// with its two multiplications).  `work` holds validate_work_bytes(curve, n) bytes (0 for the Weierstrass curves).
size_t validate_work_bytes(int curve, size_t n);
void validate_launch(const SchedEnv& env, int curve, const u32* xy, const unsigned char* inf, unsigned char* ok, void* work, size_t n, hipStream_t s);

// kernels_ecdsa.hip: KeyExchange::derive_shared_secret for secp256k1 / P-256 (secp256k1.rs:1884-1904, p256.rs:2281-2312):
// validation + from_affine, the variable-base multiplication, to_affine + x.to_bytes().  out: 8 words (32 bytes) per element.
size_t ecdh_work_bytes(size_t n);
void ecdh_launch(const SchedEnv& env, int curve, const u32* sk, const u32* pk, const unsigned char* pk_inf, u32* out, unsigned char* status,
                 void* work, size_t n, hipStream_t s);

// kernels_ecdsa.hip: Eddsa verify around the Ed25519 multiplications (eddsa.rs:174-211, 430-447).
// eddsa_pre_launch: a[i] = from_affine(pk[i]) (32 words); eddsa_finish_launch: status from sg = multiply(G, s),
// ka = multiply(A, k), R.
void eddsa_pre_launch(const u32* pk, const unsigned char* pk_inf, u32* a, size_t n, hipStream_t s);
void eddsa_finish_launch(const u32* sg, const u32* ka, const u32* r_xy, const unsigned char* r_inf, unsigned char* status,
                         size_t n, hipStream_t s);

// kernels_ecdsa.hip: Schnorr::<C, D>::verify per signature (schnorr.rs:90-140) around the curve's multiplications:
// a[i] = from_affine(pk[i]); status from sg = multiply(G, s), ep = multiply(A, e), R.  Work: A, sg, ep.
size_t schnorr_verify_work_bytes(int curve, size_t n);
void schnorr_verify_pre_launch(int curve, const u32* pk, const unsigned char* pk_inf, u32* a, size_t n, hipStream_t s);
void schnorr_verify_finish_launch(int curve, const u32* sg, const u32* ep, const u32* r_xy, const unsigned char* r_inf,
                                  unsigned char* status, size_t n, hipStream_t s);

}  // namespace fecgpu
