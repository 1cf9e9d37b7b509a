/*
 * forge_ec_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT A PRODUCT PATH)
 *
 * A plain-C restatement of the batched scalar-multiplication hot path of
 * tanm-sys/forge-ec (forge-ec-curves/src/{secp256k1,p256,ed25519}.rs), written
 * from the reference's *integer op sequences*, including its arithmetic quirks
 * (SURVEY.md section 8a).  Each function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The shipped library (libfecgpu.so) never links, loads or
 * calls it, and has no CPU fallback.
 *
 * PARITY PINNING: the Rust reference cannot be compiled in this environment (no
 * rustc/cargo, no network).  This oracle is therefore pinned by (1) every
 * known-answer value the reference's own unit tests hold for this path
 * (tests/golden/reference_kats.json, cited per vector) and (2) bit-exact
 * agreement with an independently written Python restatement
 * (oracle/py_model.py).  The *ladder's* numeric output is asserted by none of the
 * reference's tests (only "not identity"), so scalar-multiplication outputs are
 * restatement-derived: "parity pinned by KATs at the field/point level, unpinned
 * at the full-ladder level".
 *
 * Layout: every field element / scalar is uint64_t[4], little-endian limbs.
 * Weierstrass points are X,Y,Z (12 limbs, Jacobian); Ed25519 points X,Y,Z,T (16).
 */
#ifndef FORGE_EC_ORACLE_H
#define FORGE_EC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { FO_SECP256K1 = 0, FO_P256 = 1, FO_ED25519 = 2 };

/* number of 64-bit limbs of one projective point of `curve` (12 or 16), 0 if unknown */
int fo_point_limbs(int curve);

/* ---- field ops: op in {"add","sub","mul","sqr","neg","inv"}; b ignored for unary ops ---- */
int fo_field_op(int curve, const char* op, const uint64_t a[4], const uint64_t b[4], uint64_t r[4]);

/* ---- point ops ---- */
void fo_identity(int curve, uint64_t* p);
void fo_generator(int curve, uint64_t* p);
int  fo_is_identity(int curve, const uint64_t* p);
void fo_point_add(int curve, const uint64_t* p, const uint64_t* q, uint64_t* r);
/* the `double` the reference's multiply actually reaches (secp256k1: inherent, 1502-1540) */
void fo_point_double(int curve, const uint64_t* p, uint64_t* r);
/* secp256k1 only: the trait PointProjective::double (1375-1418) */
void fo_secp256k1_point_double_trait(const uint64_t* p, uint64_t* r);
void fo_point_negate(int curve, const uint64_t* p, uint64_t* r);
/* out = x,y (8 limbs) + returns 1 if infinity */
int  fo_to_affine(int curve, const uint64_t* p, uint64_t* xy);

/* ---- Curve::multiply, the reference's full work (discarded doublings included) ---- */
void fo_multiply(int curve, const uint64_t* point, const uint64_t scalar[4], uint64_t* out);

/* ---- secp256k1 scalar field + ECDSA verify (forge-ec-signature/src/ecdsa.rs:213-281) ---- */
int fo_secp256k1_scalar_op(const char* op, const uint64_t a[4], const uint64_t b[4], uint64_t r[4]);
/* 1 valid, 0 invalid, 2 = the reference panics (CtOption::unwrap on None) */
int fo_secp256k1_ecdsa_verify(const unsigned char digest[32], const uint64_t r[4], const uint64_t s[4],
                              const uint64_t pk_xy[8], int pk_inf);
void fo_batch_secp256k1_ecdsa_verify(const unsigned char* digests, const uint64_t* r, const uint64_t* s,
                                     const uint64_t* pk_xy, const uint8_t* pk_inf, uint8_t* out, size_t n,
                                     int nthreads);

/* ---- P-256 scalar field (p256.rs:875-1038, 1409-1432) + ECDSA verify for C = P256 ---- */
int fo_p256_scalar_op(const char* op, const uint64_t a[4], const uint64_t b[4], uint64_t r[4]);
int fo_p256_ecdsa_verify(const unsigned char digest[32], const uint64_t r[4], const uint64_t s[4],
                         const uint64_t pk_xy[8], int pk_inf);
void fo_batch_p256_ecdsa_verify(const unsigned char* digests, const uint64_t* r, const uint64_t* s,
                                const uint64_t* pk_xy, const uint8_t* pk_inf, uint8_t* out, size_t n,
                                int nthreads);

/* Ecdsa::<C, D>::batch_verify (ecdsa.rs:287-391), C = Secp256k1 (0) or P256 (1), digests and weights a_i given:
 * 1 true, 0 false, 2 the reference panics, -1 bad curve; detail (16 limbs or NULL) = r_sum, r_scalar_sum */
int fo_ecdsa_batch_verify(int curve, const unsigned char* digests, const uint64_t* r, const uint64_t* s,
                          const uint64_t* pk_xy, const uint8_t* pk_inf, const uint64_t* a, size_t n,
                          uint64_t* detail);

/* Curve::validate_point: is_on_curve for secp256k1 / P-256 (their overrides), the trait default for Ed25519
 * (on the curve AND order * (8 * P) is the identity): 1 / 0 */
int fo_validate_point(int curve, const uint64_t xy[8], int inf);
void fo_batch_validate_point(int curve, const uint64_t* xy, const uint8_t* inf, uint8_t* ok, size_t n, int nthreads);

/* KeyExchange::derive_shared_secret (secp256k1.rs:1884-1904, p256.rs:2281-2312): status 0 = Ok(out), 1 =
 * Err(InvalidPublicKey) (P-256 validation), 2 = Err for an identity result; -1 for a curve without KeyExchange */
int fo_ecdh(int curve, const uint64_t sk[4], const uint64_t pk_xy[8], int pk_inf, unsigned char out[32]);
void fo_batch_ecdh(int curve, const uint64_t* sk, const uint64_t* pk_xy, const uint8_t* pk_inf, unsigned char* out,
                   uint8_t* status, size_t n, int nthreads);

/* Eddsa::<Ed25519, D>::verify / Ed25519::verify from the point computation on (eddsa.rs:174-211, 430-447),
 * s and k = from_bytes_reduced(hash) given: 1 true, 0 false, 2 the reference panics */
int fo_ed25519_eddsa_verify(const uint64_t r_xy[8], int r_inf, const uint64_t pk_xy[8], int pk_inf,
                            const uint64_t s[4], const uint64_t k[4]);
void fo_batch_ed25519_eddsa_verify(const uint64_t* r_xy, const uint8_t* r_inf, const uint64_t* pk_xy,
                                   const uint8_t* pk_inf, const uint64_t* s, const uint64_t* k, uint8_t* out,
                                   size_t n, int nthreads);

/* schnorr::batch_verify::<Secp256k1, D> (forge-ec-signature/src/schnorr.rs:194-290) with the challenges
 * e_i and the random weights a_i supplied; 1 = true, 0 = false.  sides / sides_inf (optional): the
 * two affine points the reference compares at 286 */
int fo_secp256k1_schnorr_batch_verify(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint64_t* r_xy,
                                      const uint8_t* r_inf, const uint64_t* s, const uint64_t* a,
                                      const uint64_t* e, size_t n, uint64_t* sides, uint8_t* sides_inf);

/* the same for C = P256 (its own point and scalar arithmetic) */
int fo_p256_schnorr_batch_verify(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint64_t* r_xy,
                                 const uint8_t* r_inf, const uint64_t* s, const uint64_t* a,
                                 const uint64_t* e, size_t n, uint64_t* sides, uint8_t* sides_inf);
/* the same for C = Ed25519 with the RELEASE-profile scalar Mul (ed25519.rs:1256-1376: u128 sums wrap; Cargo.toml:53-58);
 * 2 = the reference panics in to_affine; *debug_build_panics (optional) = 1 when some s_i * a_i wrapped a u128, i.e. a
 * debug build (overflow checks on) panics on these inputs */
int fo_ed25519_schnorr_batch_verify(const uint64_t* pk_xy, const uint8_t* pk_inf, const uint64_t* r_xy,
                                    const uint8_t* r_inf, const uint64_t* s, const uint64_t* a, const uint64_t* e,
                                    size_t n, uint64_t* sides, uint8_t* sides_inf, uint8_t* debug_build_panics);
/* impl Mul for Scalar (Ed25519) under the release profile: out = a * b, returns 1 when a u128 sum wrapped */
int fo_ed25519_scalar_mul_release(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
/* Schnorr::<C, D>::verify per signature (schnorr.rs:90-140) from the point computation on, curve 0 / 1 / 2:
 * 1 true, 0 false, 2 = the reference panics (Ed25519 only); e = from_bytes_reduced(hash) supplied */
int fo_schnorr_verify(int curve, const uint64_t pk_xy[8], int pk_inf, const uint64_t r_xy[8], int r_inf,
                      const uint64_t s[4], const uint64_t e[4]);
void fo_batch_schnorr_verify(int curve, const uint64_t* pk_xy, const uint8_t* pk_inf, const uint64_t* r_xy,
                             const uint8_t* r_inf, const uint64_t* s, const uint64_t* e, uint8_t* status, size_t n,
                             int nthreads);

/* ---- batched drivers (nthreads host threads over contiguous shards) ---- */
void fo_batch_mul(int curve, const uint64_t* scalars, const uint64_t* points, uint64_t* out,
                  size_t n, int nthreads);
void fo_batch_mul_fixed(int curve, const uint64_t* scalars, const uint64_t* base, uint64_t* out,
                        size_t n, int nthreads);
/* out[i] = multiply(G,u1[i]) + multiply(Q[i],u2[i])   (forge-ec-signature/src/ecdsa.rs:254-256) */
void fo_batch_double_mul(int curve, const uint64_t* u1, const uint64_t* u2, const uint64_t* q,
                         uint64_t* out, size_t n, int nthreads);
/* out[i] = to_affine(points[i]) as x,y (8 limbs); inf[i] = 1 when identity */
void fo_batch_to_affine(int curve, const uint64_t* points, uint64_t* xy, uint8_t* inf, size_t n,
                        int nthreads);

/* out[i] = PointAffine::to_bytes (33 bytes) of the affine point (xy[i], inf[i]) */
void fo_batch_compress(int curve, const uint64_t* xy, const uint8_t* inf, unsigned char* out, size_t n);

/* ---- point decoding (see the block comment in forge_ec_oracle.c): ok[i] = 1 for Some, 0 for None ---- */
void fo_batch_decompress(int curve, const unsigned char* in33, uint64_t* xy, uint8_t* inf, uint8_t* ok, size_t n);
void fo_batch_encode_uncompressed(int curve, const uint64_t* xy, const uint8_t* inf, unsigned char* out65, size_t n);
void fo_batch_decode_uncompressed(int curve, const unsigned char* in65, uint64_t* xy, uint8_t* inf, uint8_t* ok, size_t n);

#ifdef __cplusplus
}
#endif
#endif
