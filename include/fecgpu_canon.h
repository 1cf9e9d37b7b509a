/*
 * fecgpu_canon.h -- CANONICAL-MATH MODE of libfecgpu.so.
 *
 * *** NOT REFERENCE PARITY. ***  Everything in fecgpu.h reproduces forge-ec's CPU arithmetic bit
 * for bit, and that arithmetic is not the secp256k1 group: its outputs are
 * not public keys any other library would accept.  The entry points below compute the REAL curves
 * -- secp256k1 (SEC 2), NIST P-256 (FIPS 186-4) and Ed25519 (RFC 8032) -- for callers that want
 * standard results at GPU speed.  They replace the same loops as fec_batch_mul_fixed /
 * fec_batch_mul + fec_batch_to_affine (key generation forge-ec-examples/src/ecdh.rs:27-49,
 * forge-ec-signature/src/ecdsa.rs:111-112; ECDH ecdh.rs:51-70), but their results differ from the
 * reference's by design; they are validated against an independent big-integer model
 * (oracle/canon_model.py) and public standard vectors (SEC2 / BIP-340 multiples of G, the RFC 6979
 * A.2.5 P-256 key pair, the RFC 8032 section 7.1 Ed25519 key pairs), never against the reference.  A maintainer adopts them only together with a fix of the reference's
 * field arithmetic.
 *
 * Layout: scalars and coordinates are uint64_t[4] little-endian limbs holding the plain integer
 * (no Montgomery form); an affine point is x then y = 8 limbs; arrays are arrays-of-structs.
 * status[i]: 0 = finite point in out_xy[i]; 1 = the result is the point at infinity
 * (out_xy[i] = 0); 2 = input point i is not on the curve / not canonical (out_xy[i] = 0).
 *
 * Scalars may be any 256-bit value; k*P is computed for k as given (the group law reduces it
 * modulo n).  These routines are NOT constant-time with respect to the scalar: table lookups are
 * indexed by scalar digits.  Use them for public or batch-verification workloads, or accept the
 * GPU's threat model explicitly.
 *
 * The host-pointer entry points stream the batch through the device in chunks of fec_ctx_set_chunk elements
 * (default 2^18), so device staging and per-element scratch are bounded by one chunk for any n; the *_dev
 * entry points work on the whole resident batch.
 */
#ifndef FECGPU_CANON_H
#define FECGPU_CANON_H

#include "fecgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { FEC_CANON_FINITE = 0, FEC_CANON_INFINITY = 1, FEC_CANON_BAD_POINT = 2 } fec_canon_status;

/* field opcode for fec_canon_field_op beyond fec_field_opcode: modular inverse (0 -> 0) */
#define FEC_F_INV 5

/* curve: FEC_SECP256K1, FEC_P256 (affine Weierstrass x, y; Jacobian inside) or FEC_ED25519 (RFC 8032
 * twisted Edwards x, y; extended coordinates inside; no point at infinity: status is 0 or 2) */

/* out_xy[i] = scalars[i] * G, affine.  Fixed-base 8-bit comb: 32 mixed additions, no doublings, from a
 * ~512 KiB table of affine multiples of G built on the device at first use and gathered from L2
 * (Ed25519: signed bytes, affine Niels points).  FEC_CANON_COMB4=1 in the environment at ctx creation
 * selects a 4-bit comb held in LDS instead (64 additions). */
int fec_canon_mul_base(fec_ctx* ctx, fec_curve curve, const uint64_t* scalars /* n*4 */, uint64_t* out_xy /* n*8 */,
                       uint8_t* status /* n */, size_t n);
int fec_canon_mul_base_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_scalars, uint64_t* d_out_xy,
                           uint8_t* d_status, size_t n, void* stream);

/* out_xy[i] = scalars[i] * points_xy[i], affine in, affine out (ECDH).  Input points are checked
 * (coordinates < p, on the curve); a failing element gets status 2 and zero output. */
int fec_canon_mul(fec_ctx* ctx, fec_curve curve, const uint64_t* scalars /* n*4 */,
                  const uint64_t* points_xy /* n*8 */, uint64_t* out_xy /* n*8 */, uint8_t* status /* n */, size_t n);
int fec_canon_mul_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_scalars, const uint64_t* d_points_xy,
                      uint64_t* d_out_xy, uint8_t* d_status, size_t n, void* stream);

/* out_xy[i] = u1[i] * G + u2[i] * points_xy[i]  (the point computation of ECDSA / Schnorr verification):
 * the comb supplies u1 * G without doublings, the windowed ladder adds u2 * P onto it. */
int fec_canon_double_mul(fec_ctx* ctx, fec_curve curve, const uint64_t* u1 /* n*4 */, const uint64_t* u2 /* n*4 */,
                         const uint64_t* points_xy /* n*8 */, uint64_t* out_xy /* n*8 */, uint8_t* status /* n */,
                         size_t n);
int fec_canon_double_mul_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_u1, const uint64_t* d_u2,
                             const uint64_t* d_points_xy, uint64_t* d_out_xy, uint8_t* d_status, size_t n,
                             void* stream);

/* Standard ECDSA verification (FIPS 186-4 section 6.4 / SEC 1 section 4.1.4), FEC_SECP256K1 and FEC_P256:
 * z = the message digest as an integer (for SHA-256 and these curves: the 32 digest bytes read big-endian),
 * r, s = the signature, pk_xy = the affine public key; all as little-endian 64-bit limbs.
 * result[i] = 1 iff r, s in [1, n-1], the key is on the curve, R = (z/s) G + (r/s) Q is finite and
 * x(R) mod n == r.  Everything after the hash runs on the GPU: s^-1 by Fermat in Montgomery form mod n,
 * u1 G + u2 Q by comb + windowed accumulate, batched normalisation, comparison.  This is the loop
 * forge-ec-signature/src/ecdsa.rs:313-361 (batch_verify) runs per signature -- for the real curve. */
int fec_canon_ecdsa_verify(fec_ctx* ctx, fec_curve curve, const uint64_t* z /* n*4 */, const uint64_t* r /* n*4 */,
                           const uint64_t* s /* n*4 */, const uint64_t* pk_xy /* n*8 */, uint8_t* result /* n */,
                           size_t n);
int fec_canon_ecdsa_verify_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_z, const uint64_t* d_r,
                               const uint64_t* d_s, const uint64_t* d_pk_xy, uint8_t* d_result, size_t n,
                               void* stream);

/* BIP-340 Schnorr verification (secp256k1).  pk_x, r: the 32-byte big-endian x-only public key and the
 * signature's r read as integers; s: the signature's s; e: the challenge
 * int(tagged_hash("BIP0340/challenge", r || pk || m)) as an integer (reduced modulo n on the device).
 * result[i] = 1 iff pk lifts to a curve point, r < p, s < n, R = s G - e P is finite with even y and
 * x(R) == r.  The square root of lift_x, the scalar negation and everything after run on the GPU. */
int fec_canon_bip340_verify(fec_ctx* ctx, const uint64_t* pk_x /* n*4 */, const uint64_t* r /* n*4 */,
                            const uint64_t* s /* n*4 */, const uint64_t* e /* n*4 */, uint8_t* result /* n */,
                            size_t n);
int fec_canon_bip340_verify_dev(fec_ctx* ctx, const uint64_t* d_pk_x, const uint64_t* d_r, const uint64_t* d_s,
                                const uint64_t* d_e, uint8_t* d_result, size_t n, void* stream);

/* EdDSA verification (Ed25519, RFC 8032 section 5.1.7).  a_enc, r_enc: the 32-byte encodings of the public
 * key and of R read as little-endian 256-bit integers (i.e. the bytes copied into four limbs); s: the
 * signature's S; h: SHA-512(R || A || M) reduced modulo l by the caller.
 * result[i] = 1 iff both encodings decode (RFC 8032 5.1.3), S < l, h < l and S B - h A == R. */
int fec_canon_eddsa_verify(fec_ctx* ctx, const uint64_t* a_enc /* n*4 */, const uint64_t* r_enc /* n*4 */,
                           const uint64_t* s /* n*4 */, const uint64_t* h /* n*4 */, uint8_t* result /* n */,
                           size_t n);
int fec_canon_eddsa_verify_dev(fec_ctx* ctx, const uint64_t* d_a_enc, const uint64_t* d_r_enc, const uint64_t* d_s,
                               const uint64_t* d_h, uint8_t* d_result, size_t n, void* stream);

/* Arithmetic modulo the group order (n for secp256k1 / P-256, l for Ed25519) on any 256-bit inputs, results
 * in [0, order): op 0: out = a * b + c, op 1: out = a^-1 (0 for a = 0 mod order; b, c ignored).  With
 * fec_canon_mul_base this is the device side of signing, e.g. ECDSA  r = x(k G) mod n (a * 1 + 0),
 * s = k^-1 (z + r d);  EdDSA  S = h a + r (mod l).  Nonces and hashes are the caller's. */
int fec_canon_scalar_op(fec_ctx* ctx, fec_curve curve, int op, const uint64_t* a /* n*4 */,
                        const uint64_t* b /* n*4 */, const uint64_t* c /* n*4 */, uint64_t* out /* n*4 */, size_t n);

/* element-wise F_p arithmetic on canonical values (inputs must be < p): op is a fec_field_opcode
 * or FEC_F_INV; b is ignored for unary ops */
int fec_canon_field_op(fec_ctx* ctx, fec_curve curve, int op, const uint64_t* a /* n*4 */,
                       const uint64_t* b /* n*4 */, uint64_t* out /* n*4 */, size_t n);

#ifdef __cplusplus
}
#endif
#endif
