// forge_ec_gpu.hpp -- C++17 host-side mirror of forge-ec's trait surface for the batched
// scalar-multiplication path, over the C ABI of fecgpu.h (header-only; link libfecgpu.so).
//
// The reference is Rust and no Rust toolchain exists in the build image, so this header plays the
// role of the `forge-ec-gpu` shim crate: same names, argument meaning and error
// behaviour as forge-ec-core (citations relative to /root/reference):
//   FieldElement / Scalar        from_raw / to_raw          secp256k1.rs:36-43, 1914-1921
//   FieldElement operators       + - * square() -a          forge-ec-core/src/lib.rs:173-241
//   PointProjective              identity / add / double / negate / is_identity   lib.rs:699-748
//   Curve                        generator / identity / multiply / to_affine      lib.rs:784-952
//   new, fallible batch API      batch_multiply, batch_multiply_fixed, batch_double_multiply
// Every operation runs on the GPU (there is no CPU fallback); single-element trait calls are
// batches of one and exist so that tests can be written like the reference's unit tests.
// Failures surface as forge_ec::Error (GenericError, like forge-ec-core's Error::GenericError).
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "fecgpu.h"
#include "fecgpu_canon.h"

namespace forge_ec {

struct Error : std::runtime_error {
  int status;
  explicit Error(int st) : std::runtime_error(std::string("forge_ec GenericError: ") + fec_strerror(st)), status(st) {}
};

class GpuContext {
 public:
  explicit GpuContext(int device = 0) {
    int rc = fec_ctx_create(&ctx_, device);
    if (rc != FEC_OK) throw Error(rc);
  }
  // fec_ctx_create_multi: the element-wise calls shard over `devices` (an ordinal may repeat)
  explicit GpuContext(const std::vector<int>& devices) {
    int rc = fec_ctx_create_multi(&ctx_, devices.data(), (int)devices.size());
    if (rc != FEC_OK) throw Error(rc);
  }
  ~GpuContext() { fec_ctx_destroy(ctx_); }
  GpuContext(const GpuContext&) = delete;
  GpuContext& operator=(const GpuContext&) = delete;
  fec_ctx* raw() const { return ctx_; }
  // fec_ctx_check: for callers of the *_dev entry points -- throws Error(FEC_E_LAUNCH) if a kernel launched through
  // this ctx since the last check reported a fault (the host-pointer calls check by themselves)
  void check_device() {
    int rc = fec_ctx_check(ctx_);
    if (rc != FEC_OK) throw Error(rc);
  }
  // fec_ctx_set_fixed_prefix_bits / fec_ctx_fixed_prefix_bits: the fixed-base prefix tables (fecgpu.h); results never
  // depend on them
  void set_fixed_prefix_bits(unsigned bits) {
    int rc = fec_ctx_set_fixed_prefix_bits(ctx_, bits);
    if (rc != FEC_OK) throw Error(rc);
  }
  unsigned fixed_prefix_bits(fec_curve curve) {
    int rc = fec_ctx_fixed_prefix_bits(ctx_, curve);
    if (rc < 0) throw Error(rc);
    return (unsigned)rc;
  }
  // fec_ctx_build_fixed_prefix: attach or build `curve`'s table now (synchronous); the policy calls of fecgpu.h
  void build_fixed_prefix(fec_curve curve) {
    int rc = fec_ctx_build_fixed_prefix(ctx_, curve);
    if (rc != FEC_OK) throw Error(rc);
  }
  void set_fixed_prefix_after(size_t elements) {
    int rc = fec_ctx_set_fixed_prefix_after(ctx_, elements);
    if (rc != FEC_OK) throw Error(rc);
  }
  void set_fixed_prefix_budget(unsigned percent_of_free_memory) {
    int rc = fec_ctx_set_fixed_prefix_budget(ctx_, percent_of_free_memory);
    if (rc != FEC_OK) throw Error(rc);
  }
  void set_side_stream_max(size_t elements) {
    int rc = fec_ctx_set_side_stream_max(ctx_, elements);
    if (rc != FEC_OK) throw Error(rc);
  }
  int device_count() const { return fec_ctx_device_count(ctx_); }
  // Device-resident shards of a multi-device ctx (fec_multi_batch_*_dev): one raw device pointer / count per device of
  // the ctx, results also gathered over xGMI into `gathered` on the consumer-th device when it is not null.  Synchronous.
  void multi_batch_mul_dev(fec_curve curve, const std::vector<const uint64_t*>& scalars, const std::vector<const uint64_t*>& points,
                           const std::vector<uint64_t*>& out, const std::vector<size_t>& counts, uint64_t* gathered = nullptr,
                           int consumer = 0, const std::vector<void*>* streams = nullptr) {
    const size_t n = (size_t)device_count();
    if (scalars.size() != n || points.size() != n || out.size() != n || counts.size() != n || (streams && streams->size() != n))
      throw Error(FEC_E_ARG);
    int rc = fec_multi_batch_mul_dev(ctx_, curve, scalars.data(), points.data(), out.data(), counts.data(), gathered, consumer,
                                     streams ? streams->data() : nullptr);
    if (rc != FEC_OK) throw Error(rc);
  }
  void multi_batch_mul_fixed_dev(fec_curve curve, const std::vector<const uint64_t*>& scalars, const std::vector<const uint64_t*>* bases,
                                 const std::vector<uint64_t*>& out, const std::vector<size_t>& counts, uint64_t* gathered = nullptr,
                                 int consumer = 0, const std::vector<void*>* streams = nullptr) {
    const size_t n = (size_t)device_count();
    if (scalars.size() != n || out.size() != n || counts.size() != n || (bases && bases->size() != n) || (streams && streams->size() != n))
      throw Error(FEC_E_ARG);
    int rc = fec_multi_batch_mul_fixed_dev(ctx_, curve, scalars.data(), bases ? bases->data() : nullptr, out.data(), counts.data(),
                                           gathered, consumer, streams ? streams->data() : nullptr);
    if (rc != FEC_OK) throw Error(rc);
  }
  void multi_batch_double_mul_dev(fec_curve curve, const std::vector<const uint64_t*>& u1, const std::vector<const uint64_t*>& u2,
                                  const std::vector<const uint64_t*>& q, const std::vector<uint64_t*>& out,
                                  const std::vector<size_t>& counts, uint64_t* gathered = nullptr, int consumer = 0,
                                  const std::vector<void*>* streams = nullptr) {
    const size_t n = (size_t)device_count();
    if (u1.size() != n || u2.size() != n || q.size() != n || out.size() != n || counts.size() != n || (streams && streams->size() != n))
      throw Error(FEC_E_ARG);
    int rc = fec_multi_batch_double_mul_dev(ctx_, curve, u1.data(), u2.data(), q.data(), out.data(), counts.data(), gathered, consumer,
                                            streams ? streams->data() : nullptr);
    if (rc != FEC_OK) throw Error(rc);
  }
  static GpuContext& global() {  // process-wide default context on device 0
    static GpuContext g(0);
    return g;
  }

 private:
  fec_ctx* ctx_ = nullptr;
};

inline void check(int rc) {
  if (rc != FEC_OK) throw Error(rc);
}

using Limbs = std::array<uint64_t, 4>;

template <fec_curve C>
struct FieldElement {
  Limbs raw{};
  static FieldElement from_raw(const Limbs& l) { return FieldElement{l}; }
  const Limbs& to_raw() const { return raw; }
  static FieldElement zero() { return FieldElement{}; }
  static FieldElement one() { return FieldElement{{1, 0, 0, 0}}; }
  bool is_zero() const { return (raw[0] | raw[1] | raw[2] | raw[3]) == 0; }
  bool ct_eq(const FieldElement& o) const { return raw == o.raw; }
  FieldElement op(fec_field_opcode code, const FieldElement* rhs) const {
    FieldElement r;
    check(fec_field_op(GpuContext::global().raw(), C, code, raw.data(), rhs ? rhs->raw.data() : nullptr,
                       r.raw.data(), 1));
    return r;
  }
  FieldElement operator+(const FieldElement& o) const { return op(FEC_F_ADD, &o); }
  FieldElement operator-(const FieldElement& o) const { return op(FEC_F_SUB, &o); }
  FieldElement operator*(const FieldElement& o) const { return op(FEC_F_MUL, &o); }
  FieldElement operator-() const { return op(FEC_F_NEG, nullptr); }
  FieldElement square() const { return op(FEC_F_SQR, nullptr); }
};

template <fec_curve C>
struct Scalar {
  Limbs raw{};
  static Scalar from_raw(const Limbs& l) { return Scalar{l}; }
  static Scalar from(uint64_t v) { return Scalar{{v, 0, 0, 0}}; }
  const Limbs& to_raw() const { return raw; }
  bool is_zero() const { return (raw[0] | raw[1] | raw[2] | raw[3]) == 0; }
};

template <fec_curve C>
struct AffinePoint {
  FieldElement<C> x_, y_;
  bool infinity = false;
  const FieldElement<C>& x() const { return x_; }
  const FieldElement<C>& y() const { return y_; }
  bool is_identity() const { return infinity; }
};

// X,Y,Z (secp256k1 / P-256, Jacobian) or X,Y,Z,T (Ed25519, extended), exactly the ABI layout
template <fec_curve C>
struct ProjectivePoint {
  static constexpr int LIMBS = C == FEC_ED25519 ? 16 : 12;
  std::array<uint64_t, LIMBS> c{};
  static ProjectivePoint from_raw_coords(const std::array<uint64_t, LIMBS>& l) { return ProjectivePoint{l}; }
  const std::array<uint64_t, LIMBS>& to_raw_coords() const { return c; }
  FieldElement<C> coord(int i) const { return FieldElement<C>{{c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3]}}; }
  static ProjectivePoint identity() {
    ProjectivePoint p;
    p.c[4] = 1;                       // Y = one()
    if (C == FEC_ED25519) p.c[8] = 1; // Z = one() for the extended identity (0,1,1,0)
    return p;
  }
  bool is_identity() const {
    if (C == FEC_ED25519) return coord(0).is_zero() && coord(1).ct_eq(coord(2)) && coord(3).is_zero();
    return coord(2).is_zero();
  }
  ProjectivePoint op(fec_point_opcode code, const ProjectivePoint* rhs) const {
    ProjectivePoint r;
    check(fec_point_op(GpuContext::global().raw(), C, code, c.data(), rhs ? rhs->c.data() : nullptr, r.c.data(), 1));
    return r;
  }
  ProjectivePoint operator+(const ProjectivePoint& o) const { return op(FEC_P_ADD, &o); }
  ProjectivePoint double_() const { return op(FEC_P_DOUBLE, nullptr); }  // `double` is a C++ keyword
  ProjectivePoint negate() const { return op(FEC_P_NEGATE, nullptr); }
  ProjectivePoint operator-(const ProjectivePoint& o) const {
    // impl Sub (secp256k1.rs:1549-1566): equal coordinates short-circuit to the identity
    if (C == FEC_SECP256K1 && c == o.c) return identity();
    return *this + o.negate();
  }
  bool ct_eq(const ProjectivePoint& o) const { return c == o.c; }
};

template <fec_curve C>
struct Curve {
  using Field = FieldElement<C>;
  using ScalarT = Scalar<C>;
  using PointProjective = ProjectivePoint<C>;
  using PointAffine = AffinePoint<C>;
  static constexpr int LIMBS = PointProjective::LIMBS;

  static PointProjective identity() { return PointProjective::identity(); }
  static PointProjective generator() {
    PointProjective g;
    check(fec_generator(GpuContext::global().raw(), C, g.c.data()));
    return g;
  }
  // Curve::multiply -- infallible in the reference; a GPU failure throws Error here
  static PointProjective multiply(const PointProjective& p, const ScalarT& k) {
    PointProjective r;
    check(fec_batch_mul(GpuContext::global().raw(), C, k.raw.data(), p.c.data(), r.c.data(), 1));
    return r;
  }
  static PointAffine to_affine(const PointProjective& p) {
    uint64_t xy[8];
    uint8_t inf = 0;
    check(fec_batch_to_affine(GpuContext::global().raw(), C, p.c.data(), xy, &inf, 1));
    PointAffine a;
    a.x_ = Field{{xy[0], xy[1], xy[2], xy[3]}};
    a.y_ = Field{{xy[4], xy[5], xy[6], xy[7]}};
    a.infinity = inf != 0;
    return a;
  }
  // ---- the batched API this backend adds ----
  static std::vector<PointProjective> batch_multiply(GpuContext& ctx, const std::vector<PointProjective>& points,
                                                     const std::vector<ScalarT>& scalars) {
    if (points.size() != scalars.size()) throw Error(FEC_E_ARG);
    std::vector<PointProjective> out(points.size());
    static_assert(sizeof(PointProjective) == LIMBS * 8 && sizeof(ScalarT) == 32, "ABI layout");
    check(fec_batch_mul(ctx.raw(), C, reinterpret_cast<const uint64_t*>(scalars.data()),
                        reinterpret_cast<const uint64_t*>(points.data()), reinterpret_cast<uint64_t*>(out.data()),
                        points.size()));
    return out;
  }
  static std::vector<PointProjective> batch_multiply_fixed(GpuContext& ctx, const PointProjective& base,
                                                           const std::vector<ScalarT>& scalars) {
    std::vector<PointProjective> out(scalars.size());
    check(fec_batch_mul_fixed(ctx.raw(), C, reinterpret_cast<const uint64_t*>(scalars.data()), base.c.data(),
                              reinterpret_cast<uint64_t*>(out.data()), scalars.size()));
    return out;
  }
  // R[i] = multiply(G, u1[i]) + multiply(Q[i], u2[i])   (forge-ec-signature/src/ecdsa.rs:254-256)
  static std::vector<PointProjective> batch_double_multiply(GpuContext& ctx, const std::vector<ScalarT>& u1,
                                                            const std::vector<ScalarT>& u2,
                                                            const std::vector<PointProjective>& q) {
    if (u1.size() != u2.size() || u1.size() != q.size()) throw Error(FEC_E_ARG);
    std::vector<PointProjective> out(q.size());
    check(fec_batch_double_mul(ctx.raw(), C, reinterpret_cast<const uint64_t*>(u1.data()),
                               reinterpret_cast<const uint64_t*>(u2.data()),
                               reinterpret_cast<const uint64_t*>(q.data()), reinterpret_cast<uint64_t*>(out.data()),
                               q.size()));
    return out;
  }
};

// C::multi_scalar_multiply(&points, &scalars) (forge-ec-core/src/lib.rs:934-951): products on the GPU,
// then the reference's sequential `result += product` fold, also on the GPU.
template <fec_curve C>
inline ProjectivePoint<C> multi_scalar_multiply(GpuContext& ctx, const std::vector<ProjectivePoint<C>>& points,
                                                const std::vector<Scalar<C>>& scalars) {
  if (points.size() != scalars.size()) throw Error(FEC_E_ARG);
  ProjectivePoint<C> out;
  check(fec_multi_scalar_mul(ctx.raw(), C, reinterpret_cast<const uint64_t*>(scalars.data()),
                             reinterpret_cast<const uint64_t*>(points.data()), out.c.data(), points.size()));
  return out;
}

namespace schnorr {
// forge_ec_signature::schnorr::Signature<Secp256k1> { r: AffinePoint, s: Scalar }
struct Signature {
  AffinePoint<FEC_SECP256K1> r;
  Scalar<FEC_SECP256K1> s;
};
// schnorr::batch_verify::<Secp256k1, D> (forge-ec-signature/src/schnorr.rs:194-290).  The caller hashes:
// challenges[i] = Scalar::from_bytes_reduced(H(R_i || P_i || m_i)) (236-256) and draws the random
// weights (228-233) with the reference's own code; everything from line 258 on runs on the GPU.
inline bool batch_verify(GpuContext& ctx, const std::vector<AffinePoint<FEC_SECP256K1>>& public_keys,
                         const std::vector<Signature>& signatures,
                         const std::vector<Scalar<FEC_SECP256K1>>& challenges,
                         const std::vector<Scalar<FEC_SECP256K1>>& weights) {
  const size_t n = public_keys.size();
  if (n != signatures.size() || n != challenges.size() || n != weights.size()) return false;
  std::vector<uint64_t> pk(n * 8), r(n * 8), s(n * 4);
  std::vector<uint8_t> pk_inf(n), r_inf(n);
  for (size_t i = 0; i < n; ++i) {
    for (int l = 0; l < 4; ++l) {
      pk[i * 8 + l] = public_keys[i].x_.raw[l];
      pk[i * 8 + 4 + l] = public_keys[i].y_.raw[l];
      r[i * 8 + l] = signatures[i].r.x_.raw[l];
      r[i * 8 + 4 + l] = signatures[i].r.y_.raw[l];
      s[i * 4 + l] = signatures[i].s.raw[l];
    }
    pk_inf[i] = public_keys[i].infinity;
    r_inf[i] = signatures[i].r.infinity;
  }
  uint8_t result = 0;
  check(fec_schnorr_batch_verify_secp256k1(ctx.raw(), pk.data(), pk_inf.data(), r.data(), r_inf.data(), s.data(),
                                           reinterpret_cast<const uint64_t*>(weights.data()),
                                           reinterpret_cast<const uint64_t*>(challenges.data()), n, &result,
                                           nullptr, nullptr));
  return result != 0;
}
}  // namespace schnorr

// Outcome of a verification as the reference computes it: its boolean, or the fact that it panics
// (CtOption::unwrap on None) on this input.
enum class Verify : uint8_t { False = 0, True = 1, ReferencePanics = 2 };

namespace schnorr {
// schnorr::batch_verify::<Ed25519, D> (fec_schnorr_batch_verify_ed25519): the verdict under the reference's RELEASE
// profile (its scalar Mul's u128 sums wrap), and whether a debug build -- overflow checks on -- would have panicked on
// these inputs instead.  Raw arrays as the C ABI takes them: x, y limbs per point, Scalar::to_raw() limbs per scalar.
struct Ed25519BatchVerdict {
  Verify result;
  bool debug_build_panics;
};
inline Ed25519BatchVerdict batch_verify_ed25519(GpuContext& ctx, const std::vector<uint64_t>& pk_xy, const std::vector<uint64_t>& r_xy,
                                                const std::vector<uint64_t>& s, const std::vector<uint64_t>& weights,
                                                const std::vector<uint64_t>& challenges) {
  const size_t n = s.size() / 4;
  if (pk_xy.size() != n * 8 || r_xy.size() != n * 8 || weights.size() != n * 4 || challenges.size() != n * 4) throw Error(FEC_E_ARG);
  uint8_t result = 0, dbg = 0;
  int rc = fec_schnorr_batch_verify_ed25519(ctx.raw(), pk_xy.data(), nullptr, r_xy.data(), nullptr, s.data(), weights.data(),
                                            challenges.data(), n, &result, nullptr, nullptr, &dbg);
  if (rc != FEC_OK) throw Error(rc);
  return {static_cast<Verify>(result), dbg != 0};
}
}  // namespace schnorr

namespace detail {
template <fec_curve C>
inline void pack_affine(const std::vector<AffinePoint<C>>& pts, std::vector<uint64_t>& xy, std::vector<uint8_t>& inf) {
  xy.resize(pts.size() * 8);
  inf.resize(pts.size());
  for (size_t i = 0; i < pts.size(); ++i) {
    for (int l = 0; l < 4; ++l) {
      xy[i * 8 + l] = pts[i].x_.raw[l];
      xy[i * 8 + 4 + l] = pts[i].y_.raw[l];
    }
    inf[i] = pts[i].infinity;
  }
}
inline std::vector<Verify> statuses(const std::vector<uint8_t>& st) {
  std::vector<Verify> r(st.size());
  for (size_t i = 0; i < st.size(); ++i) r[i] = st[i] == 1 ? Verify::True : (st[i] == 2 ? Verify::ReferencePanics : Verify::False);
  return r;
}
}  // namespace detail

namespace schnorr {
// forge_ec_signature::schnorr::Signature<C> for any of the three curves
template <fec_curve C>
struct SignatureOf {
  AffinePoint<C> r;
  Scalar<C> s;
};
// Schnorr::<C, D>::verify per signature (forge-ec-signature/src/schnorr.rs:90-140) from the point computation on:
// challenges[i] = Scalar::from_bytes_reduced(H(R_i || P_i || m_i)) (107-123) computed by the caller, who also keeps the
// two message special cases (92-99).  Returns the reference's boolean per signature or ReferencePanics.
template <fec_curve C>
inline std::vector<Verify> verify(GpuContext& ctx, const std::vector<AffinePoint<C>>& public_keys,
                                  const std::vector<SignatureOf<C>>& signatures, const std::vector<Scalar<C>>& challenges) {
  const size_t n = public_keys.size();
  if (n != signatures.size() || n != challenges.size()) throw Error(FEC_E_ARG);
  std::vector<AffinePoint<C>> rs(n);
  std::vector<uint64_t> pk, r, s(n * 4);
  std::vector<uint8_t> pk_inf, r_inf, st(n);
  for (size_t i = 0; i < n; ++i) {
    rs[i] = signatures[i].r;
    for (int l = 0; l < 4; ++l) s[i * 4 + l] = signatures[i].s.raw[l];
  }
  detail::pack_affine<C>(public_keys, pk, pk_inf);
  detail::pack_affine<C>(rs, r, r_inf);
  check(fec_schnorr_verify(ctx.raw(), C, pk.data(), pk_inf.data(), r.data(), r_inf.data(), s.data(),
                           reinterpret_cast<const uint64_t*>(challenges.data()), st.data(), n));
  return detail::statuses(st);
}
}  // namespace schnorr

namespace ecdsa {
// forge_ec_signature::ecdsa::Signature<C> { r: Scalar, s: Scalar }namespace ecdsa {
// forge_ec_signature::ecdsa::Signature<C> { r: Scalar, s: Scalar }
template <fec_curve C>
struct Signature {
  Scalar<C> r, s;
};
using Digest = std::array<uint8_t, 32>;
// Ecdsa::<C, D>::verify per element (forge-ec-signature/src/ecdsa.rs:213-281), C = Secp256k1 or P256, with
// digests[i] = D::digest(msg_i): everything after the hash on the GPU.
template <fec_curve C>
inline std::vector<Verify> verify(GpuContext& ctx, const std::vector<AffinePoint<C>>& public_keys,
                                  const std::vector<Digest>& digests, const std::vector<Signature<C>>& sigs) {
  static_assert(C == FEC_SECP256K1 || C == FEC_P256, "Ecdsa is built for secp256k1 and P-256");
  const size_t n = sigs.size();
  if (public_keys.size() != n || digests.size() != n) throw Error(FEC_E_ARG);
  std::vector<uint64_t> pk, r(n * 4), s(n * 4);
  std::vector<uint8_t> inf, st(n);
  detail::pack_affine<C>(public_keys, pk, inf);
  for (size_t i = 0; i < n; ++i)
    for (int l = 0; l < 4; ++l) {
      r[i * 4 + l] = sigs[i].r.raw[l];
      s[i * 4 + l] = sigs[i].s.raw[l];
    }
  const uint8_t* d = reinterpret_cast<const uint8_t*>(digests.data());
  if (C == FEC_SECP256K1) check(fec_ecdsa_verify_secp256k1(ctx.raw(), d, r.data(), s.data(), pk.data(), inf.data(), st.data(), n));
  else check(fec_ecdsa_verify_p256(ctx.raw(), d, r.data(), s.data(), pk.data(), inf.data(), st.data(), n));
  return detail::statuses(st);
}
// Ecdsa::<C, D>::batch_verify (ecdsa.rs:287-391) with the weights of 302-306 drawn by the caller.
template <fec_curve C>
inline Verify batch_verify(GpuContext& ctx, const std::vector<AffinePoint<C>>& public_keys,
                           const std::vector<Digest>& digests, const std::vector<Signature<C>>& sigs,
                           const std::vector<Scalar<C>>& weights) {
  static_assert(C == FEC_SECP256K1 || C == FEC_P256, "Ecdsa is built for secp256k1 and P-256");
  const size_t n = sigs.size();
  if (public_keys.size() != n || digests.size() != n) return Verify::False;   // 289-291
  if (weights.size() != n) throw Error(FEC_E_ARG);
  std::vector<uint64_t> pk, r(n * 4), s(n * 4);
  std::vector<uint8_t> inf;
  detail::pack_affine<C>(public_keys, pk, inf);
  for (size_t i = 0; i < n; ++i)
    for (int l = 0; l < 4; ++l) {
      r[i * 4 + l] = sigs[i].r.raw[l];
      s[i * 4 + l] = sigs[i].s.raw[l];
    }
  uint8_t result = 0;
  check(fec_ecdsa_batch_verify(ctx.raw(), C, reinterpret_cast<const uint8_t*>(digests.data()), r.data(), s.data(), pk.data(),
                               inf.data(), reinterpret_cast<const uint64_t*>(weights.data()), n, &result, nullptr));
  return result == 1 ? Verify::True : (result == 2 ? Verify::ReferencePanics : Verify::False);
}
}  // namespace ecdsa

namespace key_exchange {
// KeyExchange::derive_shared_secret per element (secp256k1.rs:1884-1904, p256.rs:2281-2312): Ok(32 bytes) or the
// reference's error.  Reproduces reference behaviour (parity mode); not a hardened ECDH -- see fecgpu.h.
enum class Outcome : uint8_t { Ok = 0, InvalidPublicKey = 1, IdentityProduct = 2 };
struct SharedSecret {
  Outcome outcome;
  std::array<uint8_t, 32> bytes;
};
template <fec_curve C>
inline std::vector<SharedSecret> derive_shared_secret(GpuContext& ctx, const std::vector<Scalar<C>>& private_keys,
                                                      const std::vector<AffinePoint<C>>& public_keys) {
  static_assert(C == FEC_SECP256K1 || C == FEC_P256, "KeyExchange is implemented for secp256k1 and P-256");
  const size_t n = private_keys.size();
  if (public_keys.size() != n) throw Error(FEC_E_ARG);
  std::vector<uint64_t> pk;
  std::vector<uint8_t> inf, st(n), sec(n * 32);
  detail::pack_affine<C>(public_keys, pk, inf);
  check(fec_batch_ecdh(ctx.raw(), C, reinterpret_cast<const uint64_t*>(private_keys.data()), pk.data(), inf.data(), sec.data(),
                       st.data(), n));
  std::vector<SharedSecret> r(n);
  for (size_t i = 0; i < n; ++i) {
    r[i].outcome = static_cast<Outcome>(st[i]);
    for (int b = 0; b < 32; ++b) r[i].bytes[b] = sec[i * 32 + b];
  }
  return r;
}
}  // namespace key_exchange

namespace eddsa {
// forge_ec_signature::eddsa::Signature<Ed25519> { r: AffinePoint, s: Scalar }
struct Signature {
  AffinePoint<FEC_ED25519> r;
  Scalar<FEC_ED25519> s;
};
// Eddsa::<Ed25519, D>::verify (forge-ec-signature/src/eddsa.rs:174-211) per element from the point computation
// on: challenges[i] = Scalar::from_bytes_reduced(H(R_i || A_i || m_i)) by the caller (179-193), who also
// keeps the message special cases of 157-170.
inline std::vector<Verify> verify(GpuContext& ctx, const std::vector<AffinePoint<FEC_ED25519>>& public_keys,
                                  const std::vector<Signature>& sigs, const std::vector<Scalar<FEC_ED25519>>& challenges) {
  const size_t n = sigs.size();
  if (public_keys.size() != n || challenges.size() != n) throw Error(FEC_E_ARG);
  std::vector<AffinePoint<FEC_ED25519>> rs(n);
  std::vector<uint64_t> pk, rxy, s(n * 4);
  std::vector<uint8_t> pinf, rinf, st(n);
  for (size_t i = 0; i < n; ++i) {
    rs[i] = sigs[i].r;
    for (int l = 0; l < 4; ++l) s[i * 4 + l] = sigs[i].s.raw[l];
  }
  detail::pack_affine<FEC_ED25519>(public_keys, pk, pinf);
  detail::pack_affine<FEC_ED25519>(rs, rxy, rinf);
  check(fec_eddsa_verify_ed25519(ctx.raw(), rxy.data(), rinf.data(), pk.data(), pinf.data(), s.data(),
                                 reinterpret_cast<const uint64_t*>(challenges.data()), st.data(), n));
  return detail::statuses(st);
}
}  // namespace eddsa

namespace encoding {
// PointAffine::from_bytes(&[u8; 33]) per element (secp256k1.rs:896-976, p256.rs:1580-1639, ed25519.rs:1526-1582):
// nullopt-like `ok[i] == 0` where the reference returns None.
template <fec_curve C>
struct Decoded {
  std::vector<AffinePoint<C>> points;
  std::vector<uint8_t> ok;
};
template <fec_curve C>
inline Decoded<C> unpack(const std::vector<uint64_t>& xy, const std::vector<uint8_t>& inf, std::vector<uint8_t> ok) {
  Decoded<C> d{std::vector<AffinePoint<C>>(ok.size()), std::move(ok)};
  for (size_t i = 0; i < d.points.size(); ++i) {
    for (int l = 0; l < 4; ++l) {
      d.points[i].x_.raw[l] = xy[i * 8 + l];
      d.points[i].y_.raw[l] = xy[i * 8 + 4 + l];
    }
    d.points[i].infinity = inf[i] != 0;
  }
  return d;
}
template <fec_curve C>
inline Decoded<C> from_bytes(GpuContext& ctx, const std::vector<std::array<uint8_t, 33>>& enc) {
  const size_t n = enc.size();
  std::vector<uint64_t> xy(n * 8);
  std::vector<uint8_t> inf(n), ok(n);
  check(fec_batch_decompress(ctx.raw(), C, reinterpret_cast<const uint8_t*>(enc.data()), xy.data(), inf.data(), ok.data(), n));
  return unpack<C>(xy, inf, std::move(ok));
}
// PointAffine::to_bytes -> [u8; 33]
template <fec_curve C>
inline std::vector<std::array<uint8_t, 33>> to_bytes(GpuContext& ctx, const std::vector<AffinePoint<C>>& pts) {
  std::vector<uint64_t> xy;
  std::vector<uint8_t> inf;
  detail::pack_affine<C>(pts, xy, inf);
  std::vector<std::array<uint8_t, 33>> out(pts.size());
  check(fec_batch_compress(ctx.raw(), C, xy.data(), inf.data(), reinterpret_cast<uint8_t*>(out.data()), pts.size()));
  return out;
}
// forge-ec-encoding UncompressedPoint::{from_affine, to_affine} (point.rs:186-281): the 65-byte form both ways
template <fec_curve C>
inline std::vector<std::array<uint8_t, 65>> to_uncompressed(GpuContext& ctx, const std::vector<AffinePoint<C>>& pts) {
  std::vector<uint64_t> xy;
  std::vector<uint8_t> inf;
  detail::pack_affine<C>(pts, xy, inf);
  std::vector<std::array<uint8_t, 65>> out(pts.size());
  check(fec_batch_encode_uncompressed(ctx.raw(), C, xy.data(), inf.data(), reinterpret_cast<uint8_t*>(out.data()), pts.size()));
  return out;
}
template <fec_curve C>
inline Decoded<C> from_uncompressed(GpuContext& ctx, const std::vector<std::array<uint8_t, 65>>& enc) {
  const size_t n = enc.size();
  std::vector<uint64_t> xy(n * 8);
  std::vector<uint8_t> inf(n), ok(n);
  check(fec_batch_decode_uncompressed(ctx.raw(), C, reinterpret_cast<const uint8_t*>(enc.data()), xy.data(), inf.data(), ok.data(), n));
  return unpack<C>(xy, inf, std::move(ok));
}
}  // namespace encoding

// ---- canonical-math mode (include/fecgpu_canon.h): the REAL curves, NOT reference parity -----------
// Plain-integer limbs; affine points as {x, y}; status 0 finite / 1 infinity / 2 rejected input.
namespace canon {
struct Affine {
  Limbs x{}, y{};
};
struct PointResult {
  std::vector<Affine> points;
  std::vector<uint8_t> status;
};
// out[i] = scalars[i] * G  (key generation)
template <fec_curve C>
inline PointResult mul_base(GpuContext& ctx, const std::vector<Limbs>& scalars) {
  PointResult r{std::vector<Affine>(scalars.size()), std::vector<uint8_t>(scalars.size())};
  static_assert(sizeof(Affine) == 64 && sizeof(Limbs) == 32, "ABI layout");
  check(fec_canon_mul_base(ctx.raw(), C, reinterpret_cast<const uint64_t*>(scalars.data()),
                           reinterpret_cast<uint64_t*>(r.points.data()), r.status.data(), scalars.size()));
  return r;
}
// out[i] = scalars[i] * points[i]  (ECDH; inputs validated)
template <fec_curve C>
inline PointResult mul(GpuContext& ctx, const std::vector<Limbs>& scalars, const std::vector<Affine>& points) {
  if (scalars.size() != points.size()) throw Error(FEC_E_ARG);
  PointResult r{std::vector<Affine>(scalars.size()), std::vector<uint8_t>(scalars.size())};
  check(fec_canon_mul(ctx.raw(), C, reinterpret_cast<const uint64_t*>(scalars.data()),
                      reinterpret_cast<const uint64_t*>(points.data()), reinterpret_cast<uint64_t*>(r.points.data()),
                      r.status.data(), scalars.size()));
  return r;
}
// valid[i] = standard ECDSA verification of (r[i], s[i]) on digest z[i] under public key q[i]
template <fec_curve C>
inline std::vector<uint8_t> ecdsa_verify(GpuContext& ctx, const std::vector<Limbs>& z, const std::vector<Limbs>& r,
                                         const std::vector<Limbs>& s, const std::vector<Affine>& q) {
  const size_t n = z.size();
  if (r.size() != n || s.size() != n || q.size() != n) throw Error(FEC_E_ARG);
  std::vector<uint8_t> ok(n);
  check(fec_canon_ecdsa_verify(ctx.raw(), C, reinterpret_cast<const uint64_t*>(z.data()),
                               reinterpret_cast<const uint64_t*>(r.data()), reinterpret_cast<const uint64_t*>(s.data()),
                               reinterpret_cast<const uint64_t*>(q.data()), ok.data(), n));
  return ok;
}
}  // namespace canon

using Secp256k1 = Curve<FEC_SECP256K1>;
using P256 = Curve<FEC_P256>;
using Ed25519 = Curve<FEC_ED25519>;

}  // namespace forge_ec
