"""
Generates tests/golden/schnorr_vectors.json: fixtures for
  * Schnorr::<C, D>::verify per signature (forge-ec-signature/src/schnorr.rs:90-140) from the point computation on,
    the three curves: public key, signature point, s, the challenge e = from_bytes_reduced(hash), expected status;
  * schnorr::batch_verify::<P256, D> (194-290): keys, signatures, weights a, challenges e, expected boolean and the two
    affine points line 286 compares;
  * Ed25519's `impl Mul for Scalar` (ed25519.rs:1256-1376) as the reference's RELEASE profile runs it (u128 sums wrap;
    /root/reference/Cargo.toml:53-58 has no overflow-checks) with the flag "a debug build panics on these operands" --
    including the products the reference's own tests assert (2250-2253: 1 * 2 = 2; 2303, 2309, 2315: a * 1, a * b = b * a,
    (a * b) * c = a * (b * c) for 1, 2, 3) -- and schnorr::batch_verify::<Ed25519, D> on top of it,
from the independent Python model oracle/py_model.py (restatement-derived; not reference-executed: no rustc here, and
the reference's own tests call these functions only through its hard-coded "test message" shortcuts).

  python tests/golden/gen_schnorr.py

Under the reference's arithmetic `verify` returns false for practically every input: PointAffine::new(x, -y) (130)
re-validates the curve equation on to_affine(e * P), and the reference's point arithmetic does not stay on its own
curve.  Signatures that DO verify are built backwards: e = 1 makes e * P the key itself (multiply's first addition
returns its operand), a key that satisfies the reference's own curve test survives `new`, and R is set to what the
verification computes.  That works for P-256 (true curve points pass its is_on_curve about half the time).  No point
is known to satisfy secp256k1's or Ed25519's `new` (their field products are not products mod p, and Ed25519's D
constant is not the curve's), so those two curves contribute false cases only -- which is what the reference's
Schnorr `verify` answers for them in practice.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import py_model as M  # noqa: E402

W = 1 << 256
P256_P = 0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF
P256_B = 0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B


def limbs(v):
    return [(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]


def p256_point(rng):
    while True:
        x = rng.randrange(P256_P)
        rhs = (x * x * x - 3 * x + P256_B) % P256_P
        y = pow(rhs, (P256_P + 1) // 4, P256_P)
        if y * y % P256_P == rhs:
            return limbs(x) + limbs(y)


def verifying_case(curve, rng):
    """(pk, r, s, e) with e = 1 that verifies under the reference's arithmetic, or None for this key."""
    pk = p256_point(rng)
    s = limbs(rng.randrange(1, 1 << 250))
    e = limbs(1)
    # what the verification computes: r' = s*G + from_affine(new(x, -y)) with (x, y) = to_affine(1 * P)
    one = [1, 0, 0, 0]
    assert curve == 1
    F = M.P256c
    e_p = F.multiply((pk[0:4], pk[4:8], one), e)
    x, y, _ = F.to_affine(e_p)
    ny = F.neg(y)
    if F.sqr(ny) != M._p256_rhs(x):
        return None
    rx, ry, ri = F.to_affine(F.padd(F.multiply(F.generator(), s), (x, ny, one)))
    if ri:
        return None
    return pk, list(rx) + list(ry), s, e


def main():
    rng = random.Random(0x5C40)
    out = {"provenance": "restatement-derived by oracle/py_model.py; not reference-executed", "verify": [], "batch_p256": []}

    def emit(curve, pk, pk_inf, r, r_inf, s, e, note):
        st = M.schnorr_verify(curve, pk, pk_inf, r, r_inf, s, e)
        out["verify"].append({"curve": curve, "note": note, "pk": pk, "pk_inf": int(pk_inf), "r": r, "r_inf": int(r_inf),
                              "s": s, "e": e, "status": st})
        return st

    def rnd8():
        return limbs(rng.randrange(W)) + limbs(rng.randrange(W))

    for curve in (0, 1, 2):
        p = {0: M.Secp, 1: M.P256c, 2: M.Ed}[curve]
        for _ in range(3):
            assert emit(curve, rnd8() if curve != 2 else [v & ((1 << 63) - 1) if i % 4 == 3 else v for i, v in enumerate(rnd8())],
                        False, rnd8(), False, limbs(rng.randrange(1, 1 << 250)), limbs(rng.randrange(1, 1 << 250)),
                        "arbitrary coordinates: PointAffine::new(x, -y) is None") in (0, 2)
        assert emit(curve, rnd8(), False, rnd8(), True, limbs(7), limbs(9), "infinite signature point") == 0
        assert emit(curve, rnd8(), True, rnd8(), False, limbs(7), limbs(9), "infinite public key: e*P is the identity, new(0, 0) is None") == 0
        g = p.to_affine(p.generator())
        emit(curve, list(g[0]) + list(g[1]), False, list(g[0]) + list(g[1]), False, limbs(1), limbs(1), "generator as key and as R, s = e = 1")
    for curve in (1,):   # (Ed25519's field Mul is not multiplication mod p either: no point is known to pass its new())
        made = tries = 0
        while made < 4:
            tries += 1
            assert tries < 200, "no verifying case found for curve %d" % curve
            c = verifying_case(curve, rng)
            if c is None:
                continue
            pk, r, s, e = c
            assert emit(curve, pk, False, r, False, s, e, "verifies under the reference's arithmetic (e = 1, key passes new())") == 1
            bad = list(r)
            bad[0] ^= 1
            assert emit(curve, pk, False, bad, False, s, e, "the same with one bit of R flipped") == 0
            made += 1
    # batch_verify for P-256: random inputs (false) and the degenerate batch that is true (every a_i = 0: both folds
    # stay the identity -- multiply's zero-scalar early-out)
    for n, zero_a in ((1, False), (3, False), (2, True)):
        pk = [rnd8() for _ in range(n)]
        r = [rnd8() for _ in range(n)]
        s = [limbs(rng.randrange(1, 1 << 250)) for _ in range(n)]
        a = [[0, 0, 0, 0] if zero_a else limbs(rng.randrange(1, 1 << 250)) for _ in range(n)]
        e = [limbs(rng.randrange(1, 1 << 250)) for _ in range(n)]
        res, sides, sinf = M.p256_schnorr_batch_verify(pk, None, r, None, s, a, e)
        out["batch_p256"].append({"pk": pk, "r": r, "s": s, "a": a, "e": e, "result": res,
                                  "sides": [list(v) for v in sides], "sides_inf": sinf})
    # ---- Ed25519: the scalar Mul under the release profile, and batch_verify::<Ed25519, D> (own generator state: the
    # sections above stay byte-identical) ----
    rng2 = random.Random(0xED25519)
    S = M.Ed25519Scalar
    out["scalar_mul_ed25519"] = []
    ones = (1 << 64) - 1
    pairs = [([1, 0, 0, 0], [2, 0, 0, 0], "ed25519.rs:2250-2253 asserts 1 * 2 == 2"),
             ([1, 0, 0, 0], [1, 0, 0, 0], "ed25519.rs:2303 asserts a * one == a (a = 1)"),
             ([2, 0, 0, 0], [3, 0, 0, 0], "ed25519.rs:2315: (1 * 2) * 3 == 1 * (2 * 3): the inner product 2 * 3"),
             ([3, 0, 0, 0], [2, 0, 0, 0], "ed25519.rs:2309: commutativity, 3 * 2"),
             ([0, 0, 0, 0], [ones] * 4, "zero"),
             ([ones] * 4, [ones] * 4, "all ones: every column sum wraps"),
             ([ones, ones, 0, 0], [ones, ones, 0, 0], "two full limbs each: column 1 wraps (two products of 2^128 - 2^65 + 1)"),
             ([ones, 0, 0, 0], [ones, ones, ones, ones], "one limb by four: no column has two products, nothing wraps"),
             (list(S.ORDER), [1, 0, 0, 0], "the order itself times one: reduced to zero"),
             ([S.ORDER[0] - 1] + list(S.ORDER[1:]), [1, 0, 0, 0], "order - 1 times one: unchanged")]
    for _ in range(24):
        pairs.append((limbs(rng2.randrange(W)), limbs(rng2.randrange(W)), "random 256-bit operands"))
    for _ in range(8):
        pairs.append((limbs(rng2.randrange(1 << 120)), limbs(rng2.randrange(1 << 120)), "operands below 2^120: the product fits 256 bits"))
    for a, b, note in pairs:
        prod, ovf = S.mul_release(list(a), list(b))
        out["scalar_mul_ed25519"].append({"a": list(a), "b": list(b), "product": prod, "debug_build_panics": int(ovf), "note": note})
    assert out["scalar_mul_ed25519"][0]["product"] == [2, 0, 0, 0] and out["scalar_mul_ed25519"][1]["product"] == [1, 0, 0, 0]
    assert S.mul_release(S.mul_release([1, 0, 0, 0], [2, 0, 0, 0])[0], [3, 0, 0, 0])[0] == S.mul_release([1, 0, 0, 0], S.mul_release([2, 0, 0, 0], [3, 0, 0, 0])[0])[0]
    out["batch_ed25519"] = []

    def ed8():   # x, y with limbs an Ed25519 FieldElement may hold (top limb below 2^63)
        return [v & ((1 << 63) - 1) if i % 4 == 3 else v for i, v in enumerate(limbs(rng2.randrange(W)) + limbs(rng2.randrange(W)))]

    for n, kind in ((1, "random"), (3, "random"), (2, "zero weights"), (2, "small scalars"), (6, "random")):
        pk = [ed8() for _ in range(n)]
        r = [ed8() for _ in range(n)]
        if kind == "small scalars":   # s_i * a_i fits one u128 column each: a debug build gets through
            s_ = [limbs(rng2.randrange(1, 1 << 60)) for _ in range(n)]
            a = [limbs(rng2.randrange(1, 1 << 60)) for _ in range(n)]
        else:
            s_ = [limbs(rng2.randrange(1, 1 << 250)) for _ in range(n)]
            a = [[0, 0, 0, 0] if kind == "zero weights" else limbs(rng2.randrange(1, 1 << 250)) for _ in range(n)]
        e = [limbs(rng2.randrange(1, 1 << 250)) for _ in range(n)]
        res, sides, sinf, dbg = M.ed25519_schnorr_batch_verify(pk, None, r, None, s_, a, e)
        out["batch_ed25519"].append({"kind": kind, "pk": pk, "r": r, "s": s_, "a": a, "e": e, "result": res,
                                     "sides": [list(v) for v in sides], "sides_inf": sinf, "debug_build_panics": dbg})
    assert any(c["debug_build_panics"] for c in out["batch_ed25519"]) and not all(c["debug_build_panics"] for c in out["batch_ed25519"])
    path = os.path.join(HERE, "schnorr_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
        f.write("\n")
    print(path, len(out["verify"]), "verify cases,", len(out["batch_p256"]), "P-256 batches,", len(out["scalar_mul_ed25519"]),
          "Ed25519 scalar products (%d wrap)," % sum(c["debug_build_panics"] for c in out["scalar_mul_ed25519"]),
          [(c["kind"], c["result"], c["debug_build_panics"]) for c in out["batch_ed25519"]],
          "statuses:", sorted(set(c["status"] for c in out["verify"])))


if __name__ == "__main__":
    main()
