// GPU test program written like the reference's own unit tests (forge-ec-curves/src/
// {secp256k1,p256,ed25519}.rs `mod tests`), through the C++ mirror of the trait surface.
// Each check cites the reference assertion it reproduces.  Asserts that the reference's own code
// does not satisfy (p256.rs:2472, 2526, 2538) are replaced by what its code actually computes.
// build: g++ -std=c++17 -I include tests/cpp/test_traits.cpp -L forge_ec_amd -lfecgpu -o test_traits
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "forge_ec_gpu.hpp"
using namespace forge_ec;

static int failures = 0;
#define CHECK(cond, what) do { if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, what); ++failures; } } while (0)

static void secp256k1_field_arithmetic() {  // secp256k1.rs:2737-2761
  using F = Secp256k1::Field;
  F a = F::from_raw({1, 0, 0, 0}), b = F::from_raw({2, 0, 0, 0});
  F c = a + b;
  CHECK(c.to_raw()[0] == 3, "1 + 2 == 3");
  F d = c - a;
  CHECK(d.to_raw()[0] == 2, "3 - 1 == 2");
  F e = a * b;
  CHECK(e.to_raw()[0] == 12713681792961361445ULL, "Mul KAT raw1*raw2 (2750-2751)");
  F g = a + (-a);
  CHECK(g.is_zero(), "a + (-a) == 0");
  F h = b.square();
  CHECK(h.to_raw()[0] == 4, "square KAT raw2 (2759-2760)");
}
static void secp256k1_point_arithmetic() {  // secp256k1.rs:2763-2787
  auto g = Secp256k1::generator();
  auto g2 = g + g;
  auto g2_double = g.double_();
  auto a1 = Secp256k1::to_affine(g2), a2 = Secp256k1::to_affine(g2_double);
  CHECK(a1.x().to_raw() == a2.x().to_raw(), "(g+g).x == g.double().x (2775)");
  CHECK(a1.y().to_raw() == a2.y().to_raw(), "(g+g).y == g.double().y (2776)");
  auto g_again = g2 - g;
  CHECK(!g_again.is_identity(), "2G - G is not the identity (2782)");
  auto inf = g - g;
  CHECK(inf.is_identity(), "G - G is the identity (2785-2786)");
}
static void secp256k1_scalar_multiplication() {  // secp256k1.rs:2789-2821
  auto g = Secp256k1::generator();
  auto g2 = Secp256k1::multiply(g, Secp256k1::ScalarT::from(2));
  CHECK(!g2.is_identity(), "2*G is not the identity (2805)");
  auto g3 = Secp256k1::multiply(g, Secp256k1::ScalarT::from(3));
  CHECK(!g3.is_identity(), "3*G is not the identity (2819)");
  CHECK(Secp256k1::multiply(Secp256k1::identity(), Secp256k1::ScalarT::from(5)).is_identity(), "identity * k");
  CHECK(Secp256k1::multiply(g, Secp256k1::ScalarT::from(0)).is_identity(), "P * 0 (2637-2639)");
}
static void p256_field_and_point_arithmetic() {  // p256.rs:2354-2484
  using F = P256::Field;
  F a = F::from_raw({5, 0, 0, 0}), b = F::from_raw({7, 0, 0, 0});
  CHECK((a + b).to_raw() == (Limbs{12, 0, 0, 0}), "5 + 7 == 12");
  CHECK((b - a).to_raw() == (Limbs{2, 0, 0, 0}), "7 - 5 == 2");
  CHECK((a * b).to_raw() == (Limbs{35, 0, 0, 0}), "5 * 7 == 35");
  CHECK(a.square().to_raw() == (Limbs{25, 0, 0, 0}), "5^2 == 25");
  auto g = P256::generator();
  F x = g.coord(0), y = g.coord(1);
  F x2 = x.square();
  CHECK(x2.to_raw() == (Limbs{12074202155401100ULL, 3726334282074508753ULL, 9331909631644438744ULL, 11022199779588240050ULL}), "Gx^2 (2457)");
  CHECK((x2 * x).to_raw() == (Limbs{6985818112209442057ULL, 5293983511093485517ULL, 13285487596276262425ULL, 4350650246863171228ULL}), "Gx^3 (2458)");
  CHECK((F::from_raw({3, 0, 0, 0}) * x).to_raw() == (Limbs{15988812018543642563ULL, 7280764249650076386ULL, 16876875322344915671ULL, 4703857913423513302ULL}), "3*Gx (2459)");
  CHECK(y.square().to_raw() == (Limbs{13753198298469232017ULL, 5299206390010787296ULL, 9373276401007028734ULL, 6187767046927055789ULL}), "Gy^2 (2480)");
  // p256.rs:2494-2499: multiply(G, 2) == G.double()
  auto g2 = P256::multiply(g, P256::ScalarT::from(2));
  CHECK(P256::to_affine(g2).x().to_raw() == P256::to_affine(g.double_()).x().to_raw(), "2*G == G.double() (2499)");
  // p256.rs:2526 (3G == G + 2G) does not hold for the reference's own arithmetic; what its code
  // computes is multiply(G,3) == Add(double(G), G) bit for bit
  auto g3 = P256::multiply(g, P256::ScalarT::from(3));
  CHECK(g3.ct_eq(g.double_() + g), "3*G == Add(2G, G) (what p256.rs:2120-2156 computes)");
}
static void ed25519_field_and_scalar_multiplication() {  // ed25519.rs:2126-2166, 2405-2436
  using F = Ed25519::Field;
  F a = F::from_raw({1, 0, 0, 0}), b = F::from_raw({2, 0, 0, 0});
  CHECK((a + b).ct_eq(F::from_raw({3, 0, 0, 0})), "1 + 2 == 3");
  CHECK((b - a).ct_eq(F::from_raw({1, 0, 0, 0})), "2 - 1 == 1");
  CHECK((a * b).ct_eq(F::from_raw({2, 0, 0, 0})), "1 * 2 == 2");
  CHECK((a + (-a)).is_zero(), "a + (-a) == 0");
  CHECK(a.square().ct_eq(a * a), "square == self * self (623-625)");
  auto g = Ed25519::generator();
  CHECK(Ed25519::multiply(g, Ed25519::ScalarT::from(0)).is_identity(), "g * 0 is the identity (2436)");
  CHECK(Ed25519::multiply(Ed25519::identity(), Ed25519::ScalarT::from(5)).is_identity(), "identity * 5 (2428)");
  CHECK(Ed25519::multiply(g, Ed25519::ScalarT::from(1)).ct_eq(g), "g * 1 == g");
  CHECK(Ed25519::multiply(g, Ed25519::ScalarT::from(2)).ct_eq(g.double_()), "g * 2 == g.double()");
  CHECK(Ed25519::multiply(g, Ed25519::ScalarT::from(3)).ct_eq(g + g.double_()), "g * 3 == g + 2g (result + addend order)");
}
static void batch_api() {
  GpuContext ctx(0);
  auto g = Secp256k1::generator();
  std::vector<Secp256k1::ScalarT> ks;
  std::vector<Secp256k1::PointProjective> ps;
  for (uint64_t i = 1; i <= 300; ++i) { ks.push_back(Secp256k1::ScalarT::from_raw({i * 0x9E3779B97F4A7C15ULL, i, ~i, i << 7})); ps.push_back(g); }
  auto out = Secp256k1::batch_multiply(ctx, ps, ks);
  auto fixed = Secp256k1::batch_multiply_fixed(ctx, g, ks);
  bool same = true;
  for (size_t i = 0; i < out.size(); ++i) same = same && out[i].ct_eq(fixed[i]) && out[i].ct_eq(Secp256k1::multiply(ps[i], ks[i]));
  CHECK(same, "batch_multiply == batch_multiply_fixed == per-element multiply");
  // ecdsa.rs:254-256: R = u1*G + u2*Q
  auto r = Secp256k1::batch_double_multiply(ctx, ks, ks, ps);
  CHECK(r[7].ct_eq(Secp256k1::multiply(g, ks[7]) + Secp256k1::multiply(ps[7], ks[7])), "double-mul == r1 + r2");
  // core lib.rs:934-951: multi_scalar_multiply == identity + k0*P0 + k1*P1 + ... folded left to right
  {
    std::vector<Secp256k1::PointProjective> p3(ps.begin(), ps.begin() + 3);
    std::vector<Secp256k1::ScalarT> k3(ks.begin(), ks.begin() + 3);
    auto msm = multi_scalar_multiply<FEC_SECP256K1>(ctx, p3, k3);
    auto fold = ((Secp256k1::identity() + out[0]) + out[1]) + out[2];
    CHECK(msm.ct_eq(fold), "multi_scalar_multiply == sequential += of the products");
  }
  // schnorr.rs:194-290: all weights zero -> both folds stay at the identity -> ct_eq true via the
  // infinity flags; an identity public key rejects (218-220); an empty batch is false (197-199)
  {
    std::vector<Secp256k1::PointAffine> pks(4);
    std::vector<schnorr::Signature> sigs(4);
    std::vector<Secp256k1::ScalarT> e(4), a0(4), a1(4);
    for (uint64_t i = 0; i < 4; ++i) {
      pks[i] = Secp256k1::to_affine(out[i]);
      sigs[i].r = Secp256k1::to_affine(out[i + 4]);
      sigs[i].s = ks[i];
      e[i] = ks[i + 8];
      a1[i] = ks[i + 12];
    }
    CHECK(schnorr::batch_verify(ctx, pks, sigs, e, a0), "batch_verify with zero weights is true (286, inf & inf)");
    CHECK(!schnorr::batch_verify(ctx, pks, sigs, e, a1), "unrelated points do not verify");
    pks[2].infinity = true;
    CHECK(!schnorr::batch_verify(ctx, pks, sigs, e, a0), "identity public key rejects (218-220)");
    CHECK(!schnorr::batch_verify(ctx, {}, {}, {}, {}), "empty batch is false (197-199)");
  }
  bool threw = false;
  try { ks.pop_back(); Secp256k1::batch_multiply(ctx, ps, ks); } catch (const Error&) { threw = true; }
  CHECK(threw, "length mismatch -> Error");
}

// The signature layer and the codecs through the C++ mirror, in the style of the reference's own tests:
// properties that hold under the reference's arithmetic, no external expected values.
static void signature_layer_and_codecs() {
  GpuContext ctx(0);
  // Ecdsa::verify (ecdsa.rs:213-281): r = 0 or s = 0 is false (215-217); a digest >= n panics in the reference (239)
  for (int which = 0; which < 2; ++which) {
    auto run = [&](auto curve_tag) {
      constexpr fec_curve C = decltype(curve_tag)::value;
      using Cv = Curve<C>;
      std::vector<typename Cv::PointAffine> pks(3);
      std::vector<ecdsa::Digest> dg(3);
      std::vector<ecdsa::Signature<C>> sigs(3);
      auto g = Cv::generator();
      for (uint64_t i = 0; i < 3; ++i) {
        pks[i] = Cv::to_affine(Cv::multiply(g, Scalar<C>::from(5 + i)));
        dg[i].fill(0);
        dg[i][31] = (uint8_t)(7 + i);
        sigs[i].r = Scalar<C>::from(11 + i);
        sigs[i].s = Scalar<C>::from(13 + i);
      }
      sigs[1].r = Scalar<C>::from(0);
      dg[2].fill(0xFF);
      auto st = ecdsa::verify<C>(ctx, pks, dg, sigs);
      CHECK(st[1] == Verify::False, "ECDSA verify: r = 0 is false");
      CHECK(st[2] == Verify::ReferencePanics, "ECDSA verify: digest >= n is where the reference panics");
      // batch_verify (287-391): mismatched lengths and the empty batch are false; the first failing signature decides
      std::vector<Scalar<C>> w{Scalar<C>::from(3), Scalar<C>::from(5), Scalar<C>::from(7)};
      CHECK(ecdsa::batch_verify<C>(ctx, {}, {}, {}, {}) == Verify::False, "ECDSA batch_verify: empty batch is false (289-291)");
      CHECK(ecdsa::batch_verify<C>(ctx, pks, dg, sigs, w) == Verify::False, "ECDSA batch_verify: r = 0 at index 1 returns false before the panic at 2");
      sigs[1].r = Scalar<C>::from(12);
      CHECK(ecdsa::batch_verify<C>(ctx, pks, dg, sigs, w) == Verify::ReferencePanics, "ECDSA batch_verify: now the digest at index 2 panics");
    };
    if (which == 0) run(std::integral_constant<fec_curve, FEC_SECP256K1>{});
    else run(std::integral_constant<fec_curve, FEC_P256>{});
  }
  // Eddsa::verify (eddsa.rs:174-211): with the public key at infinity R + k*A = R, so R = to_affine(s*G) verifies;
  // any other R does not; an infinite R is false (174-177)
  {
    auto g = Ed25519::generator();
    std::vector<Ed25519::PointAffine> pks(3);
    std::vector<eddsa::Signature> sigs(3);
    std::vector<Ed25519::ScalarT> k(3);
    for (uint64_t i = 0; i < 3; ++i) {
      pks[i].infinity = true;
      sigs[i].s = Ed25519::ScalarT::from(1000 + i);
      sigs[i].r = Ed25519::to_affine(Ed25519::multiply(g, sigs[i].s));
      k[i] = Ed25519::ScalarT::from(77 + i);
    }
    sigs[1].r = Ed25519::to_affine(Ed25519::multiply(g, Ed25519::ScalarT::from(999)));
    sigs[2].r.infinity = true;
    auto st = eddsa::verify(ctx, pks, sigs, k);
    CHECK(st[0] == Verify::True, "EdDSA verify: R = s*G with A at infinity verifies");
    CHECK(st[1] == Verify::False, "EdDSA verify: another R does not");
    CHECK(st[2] == Verify::False, "EdDSA verify: an infinite R is false (174-177)");
  }
  // Schnorr::verify (schnorr.rs:90-140) on P-256 with the challenge e = 1: e*P is P itself (multiply by 1 returns the
  // point as it is), so R = to_affine(s*G + (x, -y, 1)) verifies (136-142) -- for a public key whose negation passes
  // the reference's PointAffine::new (132-134).  G itself does not (x^3 < 3x takes Sub's wrapping branch, p256.rs:470-496);
  // the true 3*G (canonical-math mode) does.  Another R does not verify; an infinite R is false (103-105)
  {
    auto g = P256::generator();
    auto g3 = canon::mul_base<FEC_P256>(ctx, {Limbs{3, 0, 0, 0}});
    CHECK(g3.status[0] == 0, "canonical 3*G is finite");
    P256::PointAffine pa;
    pa.x_ = P256::Field::from_raw(g3.points[0].x);
    pa.y_ = P256::Field::from_raw(g3.points[0].y);
    std::vector<P256::PointAffine> pks(3, pa);
    std::vector<schnorr::SignatureOf<FEC_P256>> sigs(3);
    std::vector<P256::ScalarT> e(3, P256::ScalarT::from(1));
    P256::PointProjective neg_p;
    const auto ny = -pa.y_;
    for (int l = 0; l < 4; ++l) {
      neg_p.c[l] = pa.x_.raw[l];
      neg_p.c[4 + l] = ny.raw[l];
    }
    neg_p.c[8] = 1;
    for (uint64_t i = 0; i < 3; ++i) {
      sigs[i].s = P256::ScalarT::from(4242 + i);
      sigs[i].r = P256::to_affine(P256::multiply(g, sigs[i].s) + neg_p);
    }
    sigs[1].r = P256::to_affine(P256::multiply(g, P256::ScalarT::from(5)));
    sigs[2].r.infinity = true;
    auto st = schnorr::verify<FEC_P256>(ctx, pks, sigs, e);
    CHECK(st[0] == Verify::True, "Schnorr verify: R = s*G - e*P verifies (P-256, e = 1)");
    CHECK(st[1] == Verify::False, "Schnorr verify: another R does not");
    CHECK(st[2] == Verify::False, "Schnorr verify: an infinite R is false (103-105)");
  }
  // KeyExchange::derive_shared_secret (secp256k1.rs:1884-1904): the secret is x.to_bytes() of multiply(pk, sk) --
  // the same bytes PointAffine::to_bytes carries after its tag; a zero private key gives the identity -> Err
  {
    auto g = Secp256k1::generator();
    std::vector<Secp256k1::ScalarT> sk{Secp256k1::ScalarT::from(0x1234567), Secp256k1::ScalarT::from(0)};
    std::vector<Secp256k1::PointAffine> pk(2, Secp256k1::to_affine(Secp256k1::multiply(g, Secp256k1::ScalarT::from(99))));
    auto r = key_exchange::derive_shared_secret<FEC_SECP256K1>(ctx, sk, pk);
    // from_affine (1365-1373): (x, y, one()) -- the product depends on the REPRESENTATION of the point under the
    // reference's arithmetic, so the shared point is rebuilt from the affine key exactly as derive_shared_secret does
    Secp256k1::PointProjective q;
    for (int l = 0; l < 4; ++l) {
      q.c[l] = pk[0].x_.raw[l];
      q.c[4 + l] = pk[0].y_.raw[l];
    }
    q.c[8] = 1;
    auto shared = Secp256k1::to_affine(Secp256k1::multiply(q, sk[0]));
    auto enc = encoding::to_bytes<FEC_SECP256K1>(ctx, {shared});
    bool same = r[0].outcome == key_exchange::Outcome::Ok;
    for (int b = 0; b < 32; ++b) same = same && r[0].bytes[b] == enc[0][1 + b];
    CHECK(same, "derive_shared_secret == x.to_bytes() of to_affine(multiply(pk, sk))");
    CHECK(r[1].outcome == key_exchange::Outcome::IdentityProduct, "zero private key: the product is the identity -> Err");
  }
  // codecs: to_bytes of the identity is 0x00 + zeros and decodes back to the identity on every curve;
  // the uncompressed form of an affine point carries its coordinates' to_bytes after the 0x04 tag
  {
    std::vector<Secp256k1::PointAffine> id(1);
    id[0].infinity = true;
    auto enc = encoding::to_bytes<FEC_SECP256K1>(ctx, id);
    bool zeros = true;
    for (uint8_t b : enc[0]) zeros = zeros && b == 0;
    CHECK(zeros, "to_bytes(identity) is 33 zero bytes");
    auto dec = encoding::from_bytes<FEC_SECP256K1>(ctx, enc);
    CHECK(dec.ok[0] == 1 && dec.points[0].infinity, "from_bytes of the identity encoding is Some(identity)");
    std::vector<P256::PointAffine> p(1);
    p[0] = P256::to_affine(P256::generator());
    auto u = encoding::to_uncompressed<FEC_P256>(ctx, p);
    CHECK(u[0][0] == 0x04 && u[0][1] == 0x6B && u[0][32] == 0x96 && u[0][33] == 0x4F && u[0][64] == 0xF5,
          "UncompressedPoint::from_affine(G) = 04 || Gx || Gy (P-256: to_bytes is the raw limbs)");
    auto back = encoding::from_uncompressed<FEC_P256>(ctx, u);
    CHECK(back.ok.size() == 1, "UncompressedPoint::to_affine returns one verdict per input");
  }
}

// canonical-math mode through the C++ mirror: published points and a round trip
// the fixed-base prefix tables through the C++ mirror: same products with a 10-bit table, without one, and by default
static void fixed_base_prefix_tables() {
  GpuContext ctx(0);
  auto g = P256::generator();
  std::vector<P256::ScalarT> k;
  for (uint64_t i = 0; i < 300; ++i) k.push_back(P256::ScalarT::from_raw(Limbs{i * 0x9E3779B97F4A7C15ULL + 1, i, ~i, (i * 77) << 40}));
  auto plain = P256::batch_multiply_fixed(ctx, g, k);
  ctx.set_fixed_prefix_bits(10);
  auto tabled = P256::batch_multiply_fixed(ctx, g, k);
  CHECK(ctx.fixed_prefix_bits(FEC_P256) == 10, "an explicit set_fixed_prefix_bits builds the table at the next fixed-base launch");
  bool same = plain.size() == tabled.size();
  for (size_t i = 0; same && i < plain.size(); ++i) same = plain[i].c == tabled[i].c;
  CHECK(same, "batch_multiply_fixed(G): identical with and without the prefix table");
  ctx.set_fixed_prefix_bits(0);
  auto off = P256::batch_multiply_fixed(ctx, g, k);
  same = off.size() == plain.size();
  for (size_t i = 0; same && i < plain.size(); ++i) same = plain[i].c == off[i].c;
  CHECK(same && ctx.fixed_prefix_bits(FEC_P256) == 0, "tables off: same products, no table");
}

// the device-resident multi-GPU calls through the C++ mirror, as far as a program without a HIP allocator can go: a
// single-device ctx refuses them, a {0, 0} ctx accepts empty shards; the policy setters are callable
static void multi_device_resident_calls() {
  GpuContext one(0);
  bool refused = false;
  try {
    one.multi_batch_mul_dev(FEC_P256, {nullptr}, {nullptr}, {nullptr}, {0});
  } catch (const Error& e) {
    refused = e.status == FEC_E_UNSUPPORTED;
  }
  CHECK(refused, "fec_multi_batch_mul_dev on a single-device ctx is FEC_E_UNSUPPORTED");
  GpuContext two(std::vector<int>{0, 0});
  CHECK(two.device_count() == 2, "a {0, 0} ctx has two shard workers");
  two.multi_batch_mul_dev(FEC_P256, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {0, 0});
  two.multi_batch_mul_fixed_dev(FEC_ED25519, {nullptr, nullptr}, nullptr, {nullptr, nullptr}, {0, 0});
  two.multi_batch_double_mul_dev(FEC_SECP256K1, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {0, 0});
  two.set_fixed_prefix_after(1u << 21);
  two.set_fixed_prefix_budget(25);
  two.set_side_stream_max((size_t)-1);
  CHECK(true, "empty shards and the policy setters");
}

static void canonical_mode() {
  GpuContext ctx(0);
  // secp256k1: 1*G, 2*G, 3*G (3*G.x is the BIP-340 vector-0 public key)
  auto r = canon::mul_base<FEC_SECP256K1>(ctx, {Limbs{1, 0, 0, 0}, Limbs{2, 0, 0, 0}, Limbs{3, 0, 0, 0}, Limbs{0, 0, 0, 0}});
  CHECK(r.status[0] == 0 && r.status[3] == 1, "status: finite / infinity for k = 0");
  CHECK((r.points[0].x == Limbs{0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL, 0x79BE667EF9DCBBACULL}), "1*G = G (SEC 2)");
  CHECK((r.points[2].x == Limbs{0x8601F113BCE036F9ULL, 0xB531C845836F99B0ULL, 0x49344F85F89D5229ULL, 0xF9308A019258C310ULL}), "3*G.x = BIP-340 vector 0 public key");
  // ECDH symmetry: a*(b*G) == b*(a*G)
  std::vector<Limbs> a{Limbs{0x1234567, 7, 9, 0x0FFFFFFF}}, b{Limbs{0x7654321, 5, 3, 0x0EEEEEEE}};
  auto A = canon::mul_base<FEC_SECP256K1>(ctx, a), B = canon::mul_base<FEC_SECP256K1>(ctx, b);
  auto s1 = canon::mul<FEC_SECP256K1>(ctx, a, B.points), s2 = canon::mul<FEC_SECP256K1>(ctx, b, A.points);
  CHECK(s1.status[0] == 0 && s1.points[0].x == s2.points[0].x && s1.points[0].y == s2.points[0].y, "ECDH is symmetric");
  // an off-curve point is rejected
  auto bad = B.points;
  bad[0].y[0] ^= 1;
  CHECK(canon::mul<FEC_SECP256K1>(ctx, a, bad).status[0] == 2, "off-curve input -> status 2");
  // Ed25519 base point comes back for k = 1 (RFC 8032: y = 4/5)
  auto e = canon::mul_base<FEC_ED25519>(ctx, {Limbs{1, 0, 0, 0}});
  CHECK((e.points[0].y == Limbs{0x6666666666666658ULL, 0x6666666666666666ULL, 0x6666666666666666ULL, 0x6666666666666666ULL}), "Ed25519 B.y = 4/5");
}

int main() {
  try {
    canonical_mode();
    fixed_base_prefix_tables();
    multi_device_resident_calls();
    secp256k1_field_arithmetic();
    secp256k1_point_arithmetic();
    secp256k1_scalar_multiplication();
    p256_field_and_point_arithmetic();
    ed25519_field_and_scalar_multiplication();
    batch_api();
    signature_layer_and_codecs();
  } catch (const Error& e) {
    std::printf("FAIL: %s\n", e.what());
    return 2;
  }
  std::printf(failures ? "%d check(s) failed\n" : "all trait-surface checks passed\n", failures);
  return failures ? 1 : 0;
}
