/*
 * fecgpu.h -- C ABI of the MI355X (gfx950) batched scalar-multiplication backend for forge-ec.
 *
 * This is the drop-in boundary: exactly what a Rust `extern "C"` block in a forge-ec shim crate
 * binds (INTEGRATION.md shows that binding).  The reference has no FFI of its own; each entry
 * point replaces a *loop over* one trait method of forge-ec-core (citations relative to
 * /root/reference):
 *
 *   fec_batch_mul         out[i] = C::multiply(&points[i], &scalars[i])
 *                         Curve::multiply, forge-ec-core/src/lib.rs:832; impls
 *                         forge-ec-curves/src/secp256k1.rs:2635-2692, p256.rs:2120-2156,
 *                         ed25519.rs:2062-2097; loops it replaces: forge-ec-signature/src/
 *                         ecdsa.rs:313-361, schnorr.rs:268-284, core lib.rs:944-948.
 *   fec_batch_mul_fixed   out[i] = C::multiply(&base, &scalars[i])       (key generation pattern,
 *                         forge-ec-examples/src/ecdh.rs:27-49; ecdsa.rs:111)
 *   fec_batch_double_mul  out[i] = C::multiply(&G,&u1[i]) + C::multiply(&q[i],&u2[i])
 *                         (ECDSA verify point computation, forge-ec-signature/src/ecdsa.rs:254-256)
 *   fec_batch_to_affine   xy[i] = C::to_affine(&points[i])  (Curve::to_affine, core lib.rs:820-826; impls
 *                         secp256k1.rs:1342-1363 + invert 599-632, p256.rs:1835-1857 + 343-393,
 *                         ed25519.rs:1793-1811 + 410-431/603-621) -- what every caller does right
 *                         after multiply (ecdsa.rs:112, 264)
 *   fec_multi_scalar_mul  C::multi_scalar_multiply(&points, &scalars): the products on the GPU in
 *                         parallel, then the reference's strictly sequential `result += product`
 *                         fold (core lib.rs:934-951, p256.rs:2193-2211) -- the order is part of the
 *                         result because the reference's Add is not associative
 *   fec_ecdsa_verify_secp256k1   Ecdsa::<Secp256k1, D>::verify per signature, digest supplied
 *                         (forge-ec-signature/src/ecdsa.rs:213-281; scalar field secp256k1.rs:1953-1969,
 *                         2162-2195, 2270-2297, 2410-2456; FieldElement::to_bytes 138-178)
 *   fec_ecdsa_verify_p256 Ecdsa::<P256, D>::verify per signature, digest supplied (ecdsa.rs:213-281; scalar
 *                         field p256.rs:875-1100, 1409-1432; default Scalar::ct_lt core lib.rs:497-531;
 *                         FieldElement::to_bytes 288-300)
 *   fec_batch_validate_point   Curve::validate_point (secp256k1.rs:2722-2726, p256.rs:2187-2191, core lib.rs:905-925)
 *   fec_batch_ecdh        KeyExchange::derive_shared_secret for secp256k1 / P-256 (secp256k1.rs:1884-1904,
 *                         p256.rs:2281-2312)
 *   fec_ecdsa_batch_verify   Ecdsa::<C, D>::batch_verify for secp256k1 / P-256 (ecdsa.rs:287-391; scalar Add
 *                         secp256k1.rs:2358-2378, p256.rs:1352-1375)
 *   fec_eddsa_verify_ed25519   Eddsa::<Ed25519, D>::verify / Ed25519::verify after the hash and the decoding
 *                         (forge-ec-signature/src/eddsa.rs:174-211, 430-447; from_affine ed25519.rs:1813-1826,
 *                         negate 1834-1841, Sub 1936-1947, to_affine 1793-1811)
 *   fec_batch_compress    out[i] = PointAffine::to_bytes(&points[i]) -> [u8; 33] (secp256k1.rs:875-896,
 *                         p256.rs:1558-1578, ed25519.rs:1505-1525; the bytes forge-ec-encoding's
 *                         CompressedPoint::from_affine builds, point.rs:38-67), with each curve's
 *                         FieldElement::to_bytes (secp256k1.rs:138-178, p256.rs:288-300, ed25519.rs:295-310)
 *   fec_schnorr_verify    Schnorr::<C, D>::verify per signature after the hash (forge-ec-signature/src/schnorr.rs:90-140)
 *   fec_schnorr_batch_verify   schnorr::batch_verify::<C, D> for C = Secp256k1 / P256 / Ed25519 (194-290; Ed25519 with its
 *                         Scalar Mul as the release profile runs it, ed25519.rs:1256-1376: fec_schnorr_batch_verify_ed25519)
 *   fec_schnorr_batch_verify_secp256k1   schnorr::batch_verify::<Secp256k1, D> (forge-ec-signature/src/
 *                         schnorr.rs:194-290): the 3n scalar multiplications in parallel, then the two
 *                         strictly sequential `+=` folds (268, 281) and the affine comparison (286).
 *                         The challenges e_i (236-256, a hash) and the random weights a_i (228-233,
 *                         OsRng) are computed by the caller with the reference's own code
 *   fec_field_op          FieldElement trait ops (core lib.rs:173-241): Add/Sub/Mul/Neg/square
 *   fec_point_op          PointProjective trait ops (core lib.rs:699-748): Add / double / negate
 *
 * Data layout (same as the reference's in-memory representation):
 *   field element / scalar : uint64_t[4], little-endian limbs (limb 0 least significant),
 *                            == FieldElement::to_raw() / Scalar::to_raw()
 *   Weierstrass point      : X,Y,Z  = 12 limbs (Jacobian), secp256k1 and P-256
 *   Ed25519 point          : X,Y,Z,T = 16 limbs (extended)
 *   Arrays are arrays-of-structs, element i at  base + i * limbs.
 *
 * Results are bit-exact with the reference's CPU arithmetic, including its quirks.  There is NO
 * CPU fallback: every entry point runs hand-written HIP kernels on the ctx's GPU or fails.
 *
 * Threading: a fec_ctx made by fec_ctx_create owns one HIP device, two streams and its staging
 * buffers; calls on one ctx must be serialised by the caller, different ctxs are independent.
 * Multi-GPU comes in two forms: fec_ctx_create_multi -- ONE ctx whose
 * element-wise host-pointer calls are sharded over several devices inside the library (what a Rust
 * caller binds) -- or one process per GPU, each with its own single-device ctx and the *_dev entry
 * points (what bench.py does under torch.distributed).  Every call makes its ctx's device the calling
 * thread's current HIP device (hipSetDevice) and leaves it so.  When a host-pointer call returns -- with
 * any status -- nothing it queued is still reading or writing the caller's arrays.
 *
 * Errors: 0 on success, negative fec_status otherwise.  The library itself never calls abort() and no
 * C++ exception leaves it (every entry point is a function-try-block).  A fault that a KERNEL reports
 * -- the watchdog or the index guard of a scheduler kernel -- is carried to the host in a per-ctx
 * device error word: host-pointer calls return FEC_E_LAUNCH (never FEC_OK with unusable outputs), callers
 * of the *_dev entry points ask fec_ctx_check().  What the library cannot promise is what the HIP runtime
 * underneath does on a GPU memory fault or queue exception: by default it aborts the process; a host that
 * prefers an error code sets HIP_SKIP_ABORT_ON_GPU_ERROR=1 before its first HIP call.
 *
 * Aliasing: the output array of fec_batch_mul / fec_batch_mul_dev may be the `points` array itself (an
 * element's point is not read after its result is stored; tests/test_gpu_parity.py:
 * test_batch_mul_in_place_output); any other overlap of an output with an input is undefined.
 */
#ifndef FECGPU_H
#define FECGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { FEC_SECP256K1 = 0, FEC_P256 = 1, FEC_ED25519 = 2 } fec_curve;

typedef enum {
  FEC_OK = 0,
  FEC_E_ARG = -1,         /* null pointer, unknown curve/op, misaligned device pointer */
  FEC_E_DEVICE = -2,      /* no such GPU / HIP runtime failure */
  FEC_E_OOM = -3,         /* device or pinned-host allocation failed */
  FEC_E_LAUNCH = -4,      /* kernel launch or execution failed, or a kernel reported a fault (outputs unusable) */
  FEC_E_UNSUPPORTED = -5, /* op not defined for this curve / for a multi-device ctx */
  FEC_E_COMM = -6         /* multi-device ctx: a shard worker could not be started, or a copy between two devices failed */
} fec_status;

typedef enum { FEC_F_ADD = 0, FEC_F_SUB = 1, FEC_F_MUL = 2, FEC_F_SQR = 3, FEC_F_NEG = 4 } fec_field_opcode;
typedef enum {
  FEC_P_ADD = 0,          /* impl Add for ProjectivePoint / ExtendedPoint */
  FEC_P_DOUBLE = 1,       /* the double() that Curve::multiply reaches (secp256k1: inherent) */
  FEC_P_NEGATE = 2,
  FEC_P_DOUBLE_TRAIT = 3  /* secp256k1 only: trait PointProjective::double (secp256k1.rs:1375) */
} fec_point_opcode;

typedef struct fec_ctx fec_ctx;

/* limbs per point: 12 (secp256k1, P-256), 16 (Ed25519); 0 for an unknown curve */
int fec_point_limbs(fec_curve curve);

/* device = HIP device ordinal (honours HIP_VISIBLE_DEVICES).  Fails with FEC_E_DEVICE when no
 * gfx950 GPU is usable -- there is no host fallback. */
int fec_ctx_create(fec_ctx** out, int device);
/* Multi-device ctx (the callers it serves loop over Curve::multiply per element:
 * forge-ec-signature/src/ecdsa.rs:313-361, schnorr.rs:268-284).  devices[0..n_devices) are HIP
 * ordinals, 1 <= n_devices <= 16; an ordinal may appear more than once (several shard workers on
 * one GPU).  The element-wise host-pointer entry points -- fec_batch_mul, fec_batch_mul_fixed,
 * fec_batch_double_mul, fec_batch_to_affine, fec_batch_compress, fec_batch_decompress,
 * fec_batch_encode_uncompressed, fec_batch_decode_uncompressed, fec_ecdsa_verify_secp256k1,
 * fec_ecdsa_verify_p256, fec_eddsa_verify_ed25519, fec_schnorr_verify, fec_batch_ecdh, fec_batch_validate_point,
 * fec_field_op, fec_point_op (tests/test_gpu_multi_ctx.py runs every one of them sharded) -- then
 * split the batch into n_devices contiguous shards
 * [g*n/N, (g+1)*n/N), run each shard on its device from its own host thread with that device's
 * chunked copy/compute pipeline, and write results straight into the caller's output array: the
 * "gather" is the D2H copy of each shard, there is no device-to-device exchange.  Results are
 * identical to a single-device ctx.  Entry points that are not element-wise (fec_multi_scalar_mul,
 * fec_ecdsa_batch_verify, fec_schnorr_batch_verify*, fec_generator*, the measurement hooks)
 * run on devices[0]; fec_ctx_wipe, fec_ctx_check, fec_ctx_set_chunk, the fec_ctx_set_fixed_prefix_* calls, fec_ctx_build_fixed_prefix, fec_ctx_set_side_stream_max and fec_ctx_debug_force_fault apply
 * to every shard worker;
 * the *_dev entry points take device pointers of ONE device and return FEC_E_UNSUPPORTED; shards that are already
 * RESIDENT in the devices' memory go through fec_multi_batch_*_dev (below), which also gathers the results onto one
 * device over xGMI.
 * devices == NULL means ordinals 0..n_devices-1. */
int fec_ctx_create_multi(fec_ctx** out, const int* devices, int n_devices);
/* number of shard workers of the ctx (1 for fec_ctx_create) */
int fec_ctx_device_count(fec_ctx* ctx);
void fec_ctx_destroy(fec_ctx* ctx);

/* Curve::generator() exactly as the reference builds it (secp256k1.rs:2608-2625 through its own
 * to_montgomery; p256.rs:2092-2110; ed25519.rs:2015-2052 with t = x*y), evaluated on the device
 * with the same field kernels at ctx creation.  out: fec_point_limbs(curve) limbs. */
int fec_generator(fec_ctx* ctx, fec_curve curve, uint64_t* out);

/* Device address of that generator (valid for the ctx's lifetime), for the *_dev entry points.
 * Passing it to fec_batch_mul_fixed_dev lets the Ed25519 fixed-base addend table be built once. */
const uint64_t* fec_generator_dev(fec_ctx* ctx, fec_curve curve);

/* ---- host-pointer entry points (caller-owned memory; nothing retained after return) ---- */
int fec_batch_mul(fec_ctx* ctx, fec_curve curve, const uint64_t* scalars /* n*4 */,
                  const uint64_t* points /* n*limbs */, uint64_t* out /* n*limbs */, size_t n);
int fec_batch_mul_fixed(fec_ctx* ctx, fec_curve curve, const uint64_t* scalars /* n*4 */,
                        const uint64_t* base /* limbs */, uint64_t* out /* n*limbs */, size_t n);
int fec_batch_double_mul(fec_ctx* ctx, fec_curve curve, const uint64_t* u1 /* n*4 */,
                         const uint64_t* u2 /* n*4 */, const uint64_t* q /* n*limbs */,
                         uint64_t* out /* n*limbs */, size_t n);
/* xy[i] = (x, y) limbs of to_affine(points[i]) (8 limbs per element); inf[i] = 1 for the identity
 * (x = y = 0 then).  Uses the reference's own field inversion, so results are bit-identical to the
 * reference even where its arithmetic is not a field.  A non-identity input with Z = 0 (the
 * reference would panic on CtOption::unwrap) yields x = y = 0, inf = 0 for Ed25519. */
int fec_batch_to_affine(fec_ctx* ctx, fec_curve curve, const uint64_t* points /* n*limbs */,
                        uint64_t* xy /* n*8 */, uint8_t* inf /* n */, size_t n);
/* out (one point) = sum over i of multiply(points[i], scalars[i]), folded left to right from the
 * identity exactly as the reference does; n == 0 gives the identity.  The fold is inherently serial
 * (about 7 us per term on one lane): meant for the moderate n the trait method is used with. */
int fec_multi_scalar_mul(fec_ctx* ctx, fec_curve curve, const uint64_t* scalars /* n*4 */,
                         const uint64_t* points /* n*limbs */, uint64_t* out /* limbs */, size_t n);
/* ECDSA verification as the reference computes it, one signature per element, everything after the
 * hash on the GPU.  digests: n*32 bytes exactly as the hash emits them (the reference reads them
 * big-endian); r, s: raw scalar limbs; pk_xy: the AffinePoint's x and y raw field limbs (8 per
 * element); pk_inf: its infinity flag per element, or NULL for none.  status[i] = 1 valid, 0 invalid,
 * 2 where the reference panics (CtOption::unwrap on None: digest or affine x >= n as a scalar). */
int fec_ecdsa_verify_secp256k1(fec_ctx* ctx, const uint8_t* digests /* n*32 */, const uint64_t* r /* n*4 */,
                               const uint64_t* s /* n*4 */, const uint64_t* pk_xy /* n*8 */,
                               const uint8_t* pk_inf /* n or NULL */, uint8_t* status /* n */, size_t n);
/* The same for C = P256.  Parity mode means the reference's P-256 scalar field exactly: its Mul is the
 * exact product followed by reduce_wide (p256.rs:924-1020), whose second folding round drops the high
 * half of high2 * (2^256 - n), so products are NOT a*b mod n; and its range check on r and s is the
 * Scalar trait's default ct_lt (forge-ec-core/src/lib.rs:497-531), a top-byte <= comparison that every
 * value passes.  A signature made by a conforming signer therefore does not verify here (nor in the
 * reference); fec_canon_ecdsa_verify is the standard verification. */
int fec_ecdsa_verify_p256(fec_ctx* ctx, const uint8_t* digests /* n*32 */, const uint64_t* r /* n*4 */,
                          const uint64_t* s /* n*4 */, const uint64_t* pk_xy /* n*8 */,
                          const uint8_t* pk_inf /* n or NULL */, uint8_t* status /* n */, size_t n);
/* ok[i] = Curve::validate_point(&points[i]) for AffinePoint limbs xy (8 per element) and infinity flags (or NULL):
 * Secp256k1 (secp256k1.rs:2722-2726) and P256 (p256.rs:2187-2191) override it with PointAffine::is_on_curve (an
 * infinite point counts as on the curve); Ed25519 keeps the trait default (forge-ec-core/src/lib.rs:905-925):
 * on the curve AND multiply(clear_cofactor(from_affine(p)), order()) is the identity, clear_cofactor being
 * the default multiply by 8 (885-897) -- two variable-base multiplications per point.  Under the reference's
 * arithmetic the generators of all three curves FAIL this check; that is reproduced. */
int fec_batch_validate_point(fec_ctx* ctx, fec_curve curve, const uint64_t* xy /* n*8 */, const uint8_t* inf /* n or NULL */,
                             uint8_t* ok /* n */, size_t n);
/* KeyExchange::derive_shared_secret per element (forge-ec-curves/src/secp256k1.rs:1884-1904, p256.rs:2281-2302;
 * the pattern of forge-ec-examples/src/ecdh.rs:40-49), curve = FEC_SECP256K1 or FEC_P256 (Ed25519 implements no
 * KeyExchange: FEC_E_UNSUPPORTED).  secrets[i] = the 32 bytes of Ok(x.to_bytes()) of to_affine(multiply(
 * from_affine(pk_i), sk_i)); status[i] = 0 Ok, 1 Err(InvalidPublicKey) -- P-256 only: validate_public_key
 * (2304-2312) = not the identity and is_on_curve (1636-1656), which under the reference's Sub rejects about half
 * of the true curve points -- 2 Err because the product is the identity (secrets[i] is zero for 1 and 2).
 * secp256k1 does not validate the key.  SECRETS: the host-pointer form clears its device staging (keys, shared
 * points, secrets) before returning; the P-256 multiplication is a task scheduler whose batch composition
 * depends on the key bits, i.e. NOT constant-time -- like the rest of parity mode this reproduces reference
 * behaviour and is not a hardened ECDH. */
int fec_batch_ecdh(fec_ctx* ctx, fec_curve curve, const uint64_t* private_keys /* n*4 */, const uint64_t* pk_xy /* n*8 */,
                   const uint8_t* pk_inf /* n or NULL */, uint8_t* secrets /* n*32 */, uint8_t* status /* n */, size_t n);
/* Ecdsa::<C, D>::batch_verify (forge-ec-signature/src/ecdsa.rs:287-391) for curve = FEC_SECP256K1 or FEC_P256
 * (FEC_E_UNSUPPORTED otherwise), everything after the hashes: digests, r, s, pk as for fec_ecdsa_verify_*;
 * a = the n weights the reference draws at 302-306 (the caller draws them with the reference's own
 * Scalar::random and passes the limbs).  *result = 1 true, 0 false, 2 where the reference panics (unwrap at
 * 334 or 381); n == 0 gives false (289-291).  The scalars and the 2n multiplications run in parallel; the
 * loop's early return at the first failing signature, the ordered fold r_sum += r_i and the ordered scalar
 * sum are the reference's (its Add is not associative), so this is meant for moderate n (about 4 us per
 * signature in the fold).  detail (16 limbs, or NULL): r_sum (12 Jacobian limbs) and r_scalar_sum (4), zero
 * when the loop returned early.  Host pointers only. */
int fec_ecdsa_batch_verify(fec_ctx* ctx, fec_curve curve, const uint8_t* digests /* n*32 */, const uint64_t* r /* n*4 */,
                           const uint64_t* s /* n*4 */, const uint64_t* pk_xy /* n*8 */,
                           const uint8_t* pk_inf /* n or NULL */, const uint64_t* a /* n*4 */, size_t n,
                           uint8_t* result, uint64_t* detail /* 16 or NULL */);
/* EdDSA verification as the reference computes it, from the point computation on
 * (Eddsa::<Ed25519, D>::verify, forge-ec-signature/src/eddsa.rs:174-211, and Ed25519::verify, 430-447 -- the
 * same lines of arithmetic).  The caller hashes and decodes with the reference's own code (or
 * fec_batch_decompress) and passes: r_xy / r_inf = the signature point R (AffinePoint limbs and infinity
 * flag; the generic verify returns false for an infinite R), pk_xy / pk_inf = the public key A, s = the
 * signature scalar, k = Scalar::from_bytes_reduced(hash).  The message special cases at 157-170 / 361-374
 * are the caller's.  status[i] = 1 true, 0 false, 2 where the reference panics (to_affine unwraps the
 * inverse of a zero z of a point that is not the identity, ed25519.rs:1805). */
int fec_eddsa_verify_ed25519(fec_ctx* ctx, const uint64_t* r_xy /* n*8 */, const uint8_t* r_inf /* n or NULL */,
                             const uint64_t* pk_xy /* n*8 */, const uint8_t* pk_inf /* n or NULL */,
                             const uint64_t* s /* n*4 */, const uint64_t* k /* n*4 */, uint8_t* status /* n */,
                             size_t n);
/* xy: n*8 limbs (x then y, e.g. from fec_batch_to_affine), inf: n flags or NULL (all finite), out: n*33 bytes */
int fec_batch_compress(fec_ctx* ctx, fec_curve curve, const uint64_t* xy, const uint8_t* inf, uint8_t* out,
                       size_t n);
/* ---- point decoding, and the uncompressed (65-byte) form both ways ----
 * ok[i] = 1 where the reference returns Some(point), 0 where it returns None (xy[i] and inf[i] are then
 * zero); inf[i] = 1 for the identity.  Everything is the reference's own arithmetic, including the
 * parts that make most inputs decode to None:
 *   fec_batch_decompress          PointAffine::from_bytes(&[u8; 33]) -- secp256k1.rs:896-976 (its
 *                                 FieldElement::sqrt raises to (p+1)/4 written as 16-bit words, 112-131),
 *                                 p256.rs:1580-1639, ed25519.rs:1526-1582 (evaluates the Montgomery-curve
 *                                 equation; FieldElement::from_bytes rejects any limb above p's, 315-357)
 *   fec_batch_encode_uncompressed UncompressedPoint::from_affine, forge-ec-encoding/src/point.rs:186-211:
 *                                 0x04 || x.to_bytes() || y.to_bytes(), 65 zero bytes for the identity
 *   fec_batch_decode_uncompressed UncompressedPoint::to_affine, point.rs:214-281 (C::Field::from_bytes,
 *                                 x*x*x + a*x + b with the curve's get_a / get_b, then C::PointAffine::new) */
int fec_batch_decompress(fec_ctx* ctx, fec_curve curve, const uint8_t* in /* n*33 */, uint64_t* xy /* n*8 */,
                         uint8_t* inf /* n */, uint8_t* ok /* n */, size_t n);
int fec_batch_encode_uncompressed(fec_ctx* ctx, fec_curve curve, const uint64_t* xy /* n*8 */,
                                  const uint8_t* inf /* n or NULL */, uint8_t* out /* n*65 */, size_t n);
int fec_batch_decode_uncompressed(fec_ctx* ctx, fec_curve curve, const uint8_t* in /* n*65 */, uint64_t* xy /* n*8 */,
                                  uint8_t* inf /* n */, uint8_t* ok /* n */, size_t n);
/* *result = 1 if the reference's batch_verify returns true for these inputs, else 0.  pk_xy / r_xy:
 * AffinePoint x, y raw limbs (n*8), pk_inf / r_inf their infinity flags (may be NULL = all finite);
 * s, a, e: Scalar::to_raw() limbs (n*4).  sides_xy (16 limbs, may be NULL) receives x, y of
 * to_affine(s_g) then of to_affine(r_e_p) -- the two points line 286 compares -- and sides_inf (2
 * bytes, may be NULL) their infinity flags; both stay zero when the call returns false early. */
int fec_schnorr_batch_verify_secp256k1(fec_ctx* ctx, const uint64_t* pk_xy, const uint8_t* pk_inf,
                                       const uint64_t* r_xy, const uint8_t* r_inf, const uint64_t* s,
                                       const uint64_t* a, const uint64_t* e, size_t n, uint8_t* result,
                                       uint64_t* sides_xy, uint8_t* sides_inf);
/* The same for any curve (schnorr::batch_verify is generic over C: Curve, schnorr.rs:194; the P-256 instance uses that
 * curve's point arithmetic and its Scalar Mul, p256.rs:1409-1432; FEC_ED25519: see fec_schnorr_batch_verify_ed25519,
 * which this calls without the extra flag -- *result may then also be 2). */
int fec_schnorr_batch_verify(fec_ctx* ctx, fec_curve curve, const uint64_t* pk_xy, const uint8_t* pk_inf,
                             const uint64_t* r_xy, const uint8_t* r_inf, const uint64_t* s, const uint64_t* a,
                             const uint64_t* e, size_t n, uint8_t* result, uint64_t* sides_xy, uint8_t* sides_inf);
/* schnorr::batch_verify::<Ed25519, D>.  The `s_i * a_i` of line 264 is Ed25519's `impl Mul for Scalar`
 * (ed25519.rs:1256-1376), which sums up to four 128-bit products -- and then a carry -- into a u128 without widening
 * (1268-1272, 1278).  What happens when such a sum passes 2^128 depends on the build profile: with overflow checks (a
 * debug build) it panics, under the reference's release profile (/root/reference/Cargo.toml:53-58, no `overflow-checks`:
 * the profile whose CPU throughput BASELINE times) it wraps modulo 2^128 and the function carries on -- for full-size
 * scalars that is the usual case, not a corner.  This entry point reproduces the RELEASE behaviour and says when the two
 * differ: *debug_build_panics (1 byte, may be NULL) = 1 when, for at least one signature, one of those sums wrapped --
 * a debug build would have panicked at the first such signature instead of returning *result.
 * *result: 1 true, 0 false, 2 = the reference panics in BOTH profiles (286: to_affine unwraps the inverse of a zero z
 * of a point that is not the identity, ed25519.rs:1805; sides stay zero).  Other arguments as above. */
int fec_schnorr_batch_verify_ed25519(fec_ctx* ctx, const uint64_t* pk_xy, const uint8_t* pk_inf, const uint64_t* r_xy,
                                     const uint8_t* r_inf, const uint64_t* s, const uint64_t* a, const uint64_t* e, size_t n,
                                     uint8_t* result, uint64_t* sides_xy, uint8_t* sides_inf, uint8_t* debug_build_panics);
/* Schnorr::<C, D>::verify per signature (forge-ec-signature/src/schnorr.rs:90-140), all three curves, from the point
 * computation on: the caller keeps the two message special cases (92-99) and hashes -- e = from_bytes_reduced(H(R || P
 * || m)), 107-123, raw limbs.  status[i] = 1 true, 0 false, 2 where the reference panics (Ed25519 only: to_affine
 * unwraps the inverse of a zero z of a point that is not the identity).  The reference re-validates to_affine(e * P)
 * with PointAffine::new(x, -y) under its own arithmetic (130-134), whose None is `false`: practically every input on
 * secp256k1 and Ed25519, and every P-256 key that fails that curve's own is_on_curve, is answered false -- reproduced. */
int fec_schnorr_verify(fec_ctx* ctx, fec_curve curve, const uint64_t* pk_xy /* n*8 */, const uint8_t* pk_inf /* n or NULL */,
                       const uint64_t* r_xy /* n*8 */, const uint8_t* r_inf /* n or NULL */, const uint64_t* s /* n*4 */,
                       const uint64_t* e /* n*4 */, uint8_t* status /* n */, size_t n);
int fec_field_op(fec_ctx* ctx, fec_curve curve, fec_field_opcode op, const uint64_t* a /* n*4 */,
                 const uint64_t* b /* n*4, may be NULL for unary ops */, uint64_t* out /* n*4 */,
                 size_t n);
int fec_point_op(fec_ctx* ctx, fec_curve curve, fec_point_opcode op, const uint64_t* p /* n*limbs */,
                 const uint64_t* q /* n*limbs, may be NULL for unary ops */,
                 uint64_t* out /* n*limbs */, size_t n);

/* ---- device-pointer entry points: pointers are HIP device pointers on the ctx's device,
 * 16-byte aligned; the launch is enqueued on `stream` (a hipStream_t; NULL = the ctx's own
 * stream) and NOT synchronised -- the caller orders it like any other stream work.  Calls on one
 * ctx are serialised by the caller on the host; they MAY name different streams: some entry points
 * use ctx-owned device scratch (the Ed25519 addend table, the scratch of the composed P-256 /
 * Ed25519 double-mul, the canonical-mode work areas), so a launch that goes to another stream than
 * the ctx's previous launch is ordered after it with an event (no host blocking) -- launches of
 * one ctx therefore execute in call order whatever streams they name.  Use one ctx per stream for
 * concurrent streams.  A stream handed to a *_dev call must stay alive until the next call on the same ctx has
 * returned (or the ctx is destroyed): that call records an event on it to order itself after it. ---- */
int fec_batch_mul_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_scalars,
                      const uint64_t* d_points, uint64_t* d_out, size_t n, void* stream);
int fec_batch_mul_fixed_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_scalars,
                            const uint64_t* d_base, uint64_t* d_out, size_t n, void* stream);
int fec_batch_double_mul_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_u1,
                             const uint64_t* d_u2, const uint64_t* d_q, uint64_t* d_out, size_t n,
                             void* stream);

int fec_ecdsa_verify_secp256k1_dev(fec_ctx* ctx, const uint8_t* d_digests, const uint64_t* d_r, const uint64_t* d_s,
                                   const uint64_t* d_pk_xy, const uint8_t* d_pk_inf, uint8_t* d_status, size_t n,
                                   void* stream);
int fec_ecdsa_verify_p256_dev(fec_ctx* ctx, const uint8_t* d_digests, const uint64_t* d_r, const uint64_t* d_s,
                              const uint64_t* d_pk_xy, const uint8_t* d_pk_inf, uint8_t* d_status, size_t n,
                              void* stream);
int fec_batch_validate_point_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_xy, const uint8_t* d_inf, uint8_t* d_ok,
                                 size_t n, void* stream);
/* d_secrets 16-byte aligned; the caller owns (and clears) every buffer */
int fec_batch_ecdh_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_private_keys, const uint64_t* d_pk_xy,
                       const uint8_t* d_pk_inf, uint8_t* d_secrets, uint8_t* d_status, size_t n, void* stream);
int fec_eddsa_verify_ed25519_dev(fec_ctx* ctx, const uint64_t* d_r_xy, const uint8_t* d_r_inf, const uint64_t* d_pk_xy,
                                 const uint8_t* d_pk_inf, const uint64_t* d_s, const uint64_t* d_k, uint8_t* d_status,
                                 size_t n, void* stream);
int fec_schnorr_verify_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_pk_xy, const uint8_t* d_pk_inf,
                           const uint64_t* d_r_xy, const uint8_t* d_r_inf, const uint64_t* d_s, const uint64_t* d_e,
                           uint8_t* d_status, size_t n, void* stream);
/* d_out must be 4-byte aligned */
int fec_batch_compress_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_xy, const uint8_t* d_inf,
                           uint8_t* d_out, size_t n, void* stream);
int fec_batch_to_affine_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* d_points, uint64_t* d_xy,
                            uint8_t* d_inf, size_t n, void* stream);

/* ---- device-RESIDENT shards of a multi-device ctx (north_star: "independent scalar-muls shard
 * trivially across the 8 GPUs of one node with RCCL over xGMI only to gather results").  ctx is a fec_ctx_create_multi
 * ctx with N = fec_ctx_device_count(ctx) shard workers; every array argument has N entries.  Shard g -- counts[g]
 * elements; scalars[g], points[g] (or bases[g], or u1[g], u2[g], q[g]) and out[g] are HIP device pointers in the memory
 * of the ctx's g-th device, 16-byte aligned -- is multiplied on its own device by the same kernels as fec_batch_*_dev,
 * in chunks of fec_ctx_set_chunk elements; there is no exchange during the compute.  out[g] receives the shard's
 * counts[g] * limbs results.  If `gathered` is not NULL it is an array of (sum of counts) * limbs words in the memory of
 * the ctx's consumer-th device, and every shard's results are ALSO copied into it at the shard's offset (the sum of the
 * counts before it): one peer copy per chunk from each device straight to the consumer over their own xGMI link --
 * the direct pattern, nothing relayed, no ring -- on a stream of its own, so that a chunk's copy runs under the
 * next chunk's kernels.  (The copies are hipMemcpyPeerAsync with peer access enabled where the devices allow it; the
 * library does not link RCCL.  One process per GPU is the other way to run this: bench.py, forge_ec_amd/dist.py.)
 * streams: NULL, or N hipStream_t handles (one per device, NULL entries allowed): the stream of device g on which the
 * caller's producers of shard g were queued -- the kernels are queued behind them; NULL = the worker's own stream.
 * SYNCHRONOUS: returns when every kernel and copy has completed and every device's error word has been read
 * (FEC_E_LAUNCH as for the host-pointer calls; FEC_E_COMM when a copy between two devices failed).
 * bases (fixed base): NULL, or NULL entries = the reference's generator() (each device's own copy, with its prefix
 * table); a single-device ctx returns FEC_E_UNSUPPORTED (it has fec_batch_*_dev).
 * Measured on one GPU only (devices = {0, 0}: tests/test_gpu_multi_ctx.py); unmeasured on N > 1 hardware. ---- */
int fec_multi_batch_mul_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* const* scalars, const uint64_t* const* points,
                            uint64_t* const* out, const size_t* counts, uint64_t* gathered, int consumer,
                            void* const* streams);
int fec_multi_batch_mul_fixed_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* const* scalars,
                                  const uint64_t* const* bases, uint64_t* const* out, const size_t* counts,
                                  uint64_t* gathered, int consumer, void* const* streams);
int fec_multi_batch_double_mul_dev(fec_ctx* ctx, fec_curve curve, const uint64_t* const* u1, const uint64_t* const* u2,
                                   const uint64_t* const* q, uint64_t* const* out, const size_t* counts,
                                   uint64_t* gathered, int consumer, void* const* streams);

/* Zeroes every device buffer the ctx owns that can hold copies of caller data (host-call staging, the
 * per-stream scratch of composed launches, the canonical-mode work areas).  Synchronises the device.
 * fec_ctx_destroy calls it; call it yourself after a batch whose inputs were sensitive. */
int fec_ctx_wipe(fec_ctx* ctx);

/* The sticky device error state of the ctx, for callers of the *_dev entry points (those return once the work
 * is enqueued, so a fault reported by a kernel can only be seen afterwards; the host-pointer entry points do this
 * check themselves).  Synchronises the ctx's device(s); FEC_E_LAUNCH if a kernel launched through this ctx since
 * the last check reported a fault -- the outputs of those launches must not be used -- else FEC_OK.  Reading
 * clears the state. */
int fec_ctx_check(fec_ctx* ctx);
/* Debug / test hook: while enabled, every scheduler-kernel launch of this ctx (P-256 and Ed25519 variable-base
 * multiplication, also inside the composed entry points) raises its fault word at once, exactly as its watchdog
 * would: outputs are zero-filled and the call (or fec_ctx_check) returns FEC_E_LAUNCH. */
int fec_ctx_debug_force_fault(fec_ctx* ctx, int enabled);

/* Fixed-base prefix tables.  The state of Curve::multiply(generator(), k) after its first `bits` steps depends on the
 * first `bits` scalar bits alone, so it is computed once -- one step of the reference's loop per entry and level, with
 * the same arithmetic -- for all 2^bits patterns and kept in HBM; every multiplication by the generator
 * (fec_batch_mul_fixed with fec_generator's point, the u1*G of the ECDSA / Schnorr entry points, fec_batch_double_mul)
 * then fetches its entry and runs the remaining steps.  Results are bit-identical with or without a table.
 *
 * Policy.  A table costs device memory (24 bits: secp256k1 3.0 GiB + 1.5 GiB while it is built, P-256 1.5 + 0.75 GiB,
 * Ed25519 2.0 GiB) and its build ends in a host synchronisation, so:
 *  - there is ONE table per device, curve and size in the process: every ctx on that device -- the shard workers of a
 *    {0, 0, 0} multi-device ctx, the ctxs of several host threads -- holds a reference to the same allocation; the last
 *    reference frees it;
 *  - a table and its build scratch never take more than the budget -- 25 % by default -- of the device memory that is
 *    FREE at that moment (hipMemGetInfo): the size shrinks from the wanted bits down to 16, below that there is no
 *    table; refused memory is never an error, the launches then run the whole ladder, and the ctx asks again after
 *    another 2^21 multiplications by the generator;
 *  - a ctx left to its defaults (24 bits wanted) builds a curve's table only from a HOST-POINTER entry point (those are
 *    synchronous anyway), and only once it has multiplied 2^21 scalars by that curve's generator -- a ctx that
 *    multiplies a few thousand scalars never allocates one; its *_dev entry points only ever enqueue: they use a table
 *    that already exists on their device and never allocate or wait;
 *  - fec_ctx_set_fixed_prefix_bits is the caller ASKING for tables: it sets the wanted size (at most 28 bits; 0 = off,
 *    which also switches the per-launch tables below off), drops the ctx's references, and from then on the next
 *    multiplication by a curve's generator -- from a *_dev entry point too -- attaches or builds that curve's table before
 *    it returns (2-4 ms of kernels at 24 bits, plus the allocation).  fec_ctx_build_fixed_prefix does the same at a
 *    point of the caller's choosing: it attaches or builds `curve`'s table now and waits for it.
 * The environment variables FEC_FIXED_PREFIX_BITS / FEC_FIXED_PREFIX_AFTER / FEC_SIDE_STREAM_MAX are read once at ctx
 * creation as overrides of the defaults (experiments); the calls below are the interface.
 * A fixed base that is NOT the generator (fec_batch_mul_fixed with a point of the caller's own) gets a table for the one
 * launch, in the launch stream's scratch, sized to the batch (2^(log2(n) - 2) entries, from 2^16 elements on, never more
 * than the wanted bits), with no host synchronisation. */
int fec_ctx_set_fixed_prefix_bits(fec_ctx* ctx, unsigned bits);
/* attach or build the table of `curve` now (synchronous); FEC_OK also when no memory could be had -- ask
 * fec_ctx_fixed_prefix_bits what there is */
int fec_ctx_build_fixed_prefix(fec_ctx* ctx, fec_curve curve);
/* a ctx left to its defaults builds a curve's table once it has multiplied this many scalars by its generator (default 2^21) */
int fec_ctx_set_fixed_prefix_after(fec_ctx* ctx, size_t elements);
/* share of the device's free memory a table and its build scratch may take, in percent (0..100, default 25; 0 = never build) */
int fec_ctx_set_fixed_prefix_budget(fec_ctx* ctx, unsigned percent_of_free_memory);
/* bits of the prefix table `curve` has at this moment (0 = none: not built yet, switched off, or memory refused);
 * negative fec_status on a bad argument.  Multi-device ctx: the first shard worker's. */
int fec_ctx_fixed_prefix_bits(fec_ctx* ctx, fec_curve curve);
/* u1*G + u2*Q (fec_batch_double_mul*, the verify pipelines): multiply(G, u1) runs on the ctx's second stream beside
 * multiply(Q, u2) for launches of up to this many elements (default: every size; 0 = never).  Measurement knob
 * (tools/double_mul_small_perf.py); results do not depend on it. */
int fec_ctx_set_side_stream_max(fec_ctx* ctx, size_t elements);

/* Host-pointer batches are processed as a two-lane pipeline of `elements`-sized chunks (default
 * 2^18): copies of one chunk overlap the kernel of the other, and device staging memory is bounded
 * by two chunks for any n.  Tuning/test knob; results do not depend on it. */
int fec_ctx_set_chunk(fec_ctx* ctx, size_t elements);

/* ---- measurement hooks ---- */
/* When enabled, every kernel launched through this ctx is bracketed by HIP events recorded on
 * the launch stream. */
int fec_ctx_set_timing(fec_ctx* ctx, int enabled);
/* Synchronises the last timed launch and returns its duration in milliseconds and its name. */
int fec_ctx_last_kernel_ms(fec_ctx* ctx, float* ms, const char** kernel_name);
/* Dependency-free v_mad_u64_u32 micro-kernel: measured peak 32x32->64 multiply-adds per second
 * of this GPU (the integer-VALU roofline the scalar-mul kernels are priced against). */
int fec_measure_peak_mad32(fec_ctx* ctx, double* mad32_per_sec);
/* name / CU count / clock of the ctx's device */
int fec_ctx_device_info(fec_ctx* ctx, char* name, size_t name_len, int* compute_units, int* clock_khz);

const char* fec_strerror(int status);

#ifdef __cplusplus
}
#endif
#endif /* FECGPU_H */
