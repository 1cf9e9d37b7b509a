"""
Generates the committed fixtures under tests/golden/.

  python tests/golden/gen_golden.py

reference_kats.json   known-answer values the REFERENCE's own unit tests assert for this path
                      (data only: operands + expected limbs + the file:line that holds them).
golden_vectors.json   restatement-derived vectors (inputs + outputs) produced by the independent
                      Python model oracle/py_model.py -- NOT reference-executed (the Rust
                      reference cannot be built here: no rustc/cargo, no network).  They pin the
                      C oracle and the GPU path to the Python model on field ops, point ops
                      (every early-out branch) and full scalar multiplications.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import py_model as M  # noqa: E402
import vectors as V  # noqa: E402


def L(x):
    return V.limbs_of(x)


def reference_kats():
    gx = [0xF4A13945D898C296, 0x77037D812DEB33A0, 0xF8BCE6E563A440F2, 0x6B17D1F2E12C4247]
    gy = [0xCBB6406837BF51F5, 0x2BCE33576B315ECE, 0x8EE7EB4A7C0F9E16, 0x4FE342E2FE1A7F9B]
    gx2 = [12074202155401100, 3726334282074508753, 9331909631644438744, 11022199779588240050]
    field = [
        # secp256k1.rs test_field_arithmetic
        {"curve": 0, "op": "add", "a": L(1), "b": L(2), "expect_limb0": 3, "src": "secp256k1.rs:2740-2743"},
        {"curve": 0, "op": "sub", "a": L(3), "b": L(1), "expect_limb0": 2, "src": "secp256k1.rs:2746-2747"},
        {"curve": 0, "op": "mul", "a": L(1), "b": L(2), "expect_limb0": 12713681792961361445,
         "src": "secp256k1.rs:2750-2751"},
        {"curve": 0, "op": "sqr", "a": L(2), "b": None, "expect_limb0": 4, "src": "secp256k1.rs:2759-2760"},
        # p256.rs test_field_arithmetic / test_point_arithmetic
        {"curve": 1, "op": "add", "a": L(5), "b": L(7), "expect": L(12), "src": "p256.rs:2357-2360"},
        {"curve": 1, "op": "sub", "a": L(7), "b": L(5), "expect": L(2), "src": "p256.rs:2363-2364"},
        {"curve": 1, "op": "mul", "a": L(5), "b": L(7), "expect": L(35), "src": "p256.rs:2367-2369"},
        {"curve": 1, "op": "sqr", "a": L(5), "b": None, "expect": L(25), "src": "p256.rs:2372-2374"},
        {"curve": 1, "op": "sqr", "a": gx, "b": None, "expect": gx2, "src": "p256.rs:2457"},
        {"curve": 1, "op": "mul", "a": gx2, "b": gx,
         "expect": [6985818112209442057, 5293983511093485517, 13285487596276262425, 4350650246863171228],
         "src": "p256.rs:2458"},
        {"curve": 1, "op": "mul", "a": L(3), "b": gx,
         "expect": [15988812018543642563, 7280764249650076386, 16876875322344915671, 4703857913423513302],
         "src": "p256.rs:2459"},
        {"curve": 1, "op": "sqr", "a": gy, "b": None,
         "expect": [13753198298469232017, 5299206390010787296, 9373276401007028734, 6187767046927055789],
         "src": "p256.rs:2480"},
        # ed25519.rs test_field_arithmetic
        {"curve": 2, "op": "add", "a": L(1), "b": L(2), "expect": L(3), "src": "ed25519.rs:2136-2141"},
        {"curve": 2, "op": "sub", "a": L(2), "b": L(1), "expect": L(1), "src": "ed25519.rs:2144-2146"},
        {"curve": 2, "op": "mul", "a": L(1), "b": L(2), "expect": L(2), "src": "ed25519.rs:2149-2151"},
    ]
    props = [
        # property-style assertions of the reference's tests that hold for its arithmetic
        {"curve": 0, "name": "neg_plus_self_is_zero", "a": L(1), "src": "secp256k1.rs:2754-2756"},
        {"curve": 1, "name": "five_times_inverse_is_one", "a": L(5), "src": "p256.rs:2427-2433"},
        {"curve": 1, "name": "fermat_five", "a": L(5), "src": "p256.rs:2416-2424"},
        {"curve": 2, "name": "one_times_inverse_is_one", "a": L(1), "src": "ed25519.rs:2164-2166"},
        {"curve": 0, "name": "g_plus_g_affine_eq_double_affine", "src": "secp256k1.rs:2769-2776"},
        {"curve": 0, "name": "g_minus_g_is_identity", "src": "secp256k1.rs:2785-2786"},
        {"curve": 1, "name": "two_g_eq_double", "src": "p256.rs:2494-2499"},
        {"curve": 1, "name": "three_g_eq_g_plus_2g", "src": "p256.rs:2526",
         "holds_for_reference_arithmetic": False,
         "why": "multiply(G,3) = Add(2G,G) but the test builds Add(G,2G); Add is not symmetric because "
                "Sub (470-496) is off by 2^256-p whenever it borrows"},
        {"curve": 2, "name": "mul_small_scalars", "src": "ed25519.rs:2405-2436"},
    ]
    note = ("Known-divergent (recorded, not asserted): p256.rs:2472 expects the TRUE x^3-3x+b; the "
            "reference's own Sub (470-496) yields the value in 'p256_code_value' because x^3 < 3x takes "
            "the wrapping branch.  The build follows the code.")
    return {"field": field, "properties": props, "note": note,
            "p256_code_value": [13753198298469232018, 5299206385715820000, 9373276401007028734,
                                6187767051222023084]}


def golden_vectors():
    rng = random.Random(0x600D5EED)
    out = {"provenance": "restatement-derived by oracle/py_model.py; not reference-executed",
           "field": [], "point": [], "multiply": [], "double_mul": []}
    for curve, F in M.CURVES.items():
        p = V.PRIME[curve]
        edges = V.edge_field_values(curve)
        pairs = [(rng.choice(edges), rng.choice(edges)) for _ in range(24)]
        pairs += [(rng.randrange(p), rng.randrange(p)) for _ in range(24)]
        pairs += [(rng.randrange(1 << 256), rng.randrange(1 << 256)) for _ in range(12)]
        for a, b in pairs:
            la, lb = L(a), L(b)
            out["field"].append({"curve": curve, "a": la, "b": lb, "add": F.add(la, lb), "sub": F.sub(la, lb),
                                 "mul": F.mul(la, lb), "sqr": F.sqr(la), "neg": F.neg(la)})
        g = F.generator()
        g2 = F.double(g)
        g3 = F.padd(g, g2)
        ident = F.identity()
        nc = 3 if curve != 2 else 4
        rnd = [tuple(L(rng.randrange(p)) for _ in range(nc)) for _ in range(3)]
        if curve == 2:
            neg = lambda q: (F.neg(q[0]), q[1], q[2], F.neg(q[3]))  # noqa: E731
        else:
            neg = lambda q: (q[0], F.neg(q[1]), q[2])  # noqa: E731
        pts = [g, g2, g3, ident, neg(g), neg(g3)] + rnd
        for a in pts:
            for b in pts:
                out["point"].append({"curve": curve, "p": M.flat(a), "q": M.flat(b), "add": M.flat(F.padd(a, b))})
            out["point"].append({"curve": curve, "p": M.flat(a), "double": M.flat(F.double(a))})
        scal = [1, 2, 3, 5, 1 << 255, (1 << 256) - 1, 1 << 248, 0x80, 0xFF,
                0x0102030405060708090A0B0C0D0E0F101112131415161718191A1B1C1D1E1F20]
        cases = [(g, s) for s in scal] + [(rnd[0], 1 << 255), (ident, 5), (g, 0)]
        cases += [(rnd[i % 3], rng.randrange(1, 1 << 256)) for i in range(6)]
        for pt, s in cases:
            out["multiply"].append({"curve": curve, "point": M.flat(pt), "scalar": L(s),
                                    "out": M.flat(F.multiply(pt, L(s)))})
        for _ in range(2):
            u1, u2 = rng.randrange(1, 1 << 256), rng.randrange(1, 1 << 256)
            q = rnd[1]
            out["double_mul"].append({"curve": curve, "u1": L(u1), "u2": L(u2), "q": M.flat(q),
                                      "out": M.flat(M.double_mul(curve, L(u1), L(u2), q))})
        u = rng.randrange(1, 1 << 256)
        out["double_mul"].append({"curve": curve, "u1": L(u), "u2": L(u), "q": M.flat(g),
                                  "out": M.flat(M.double_mul(curve, L(u), L(u), g))})
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "reference_kats.json"), "w") as f:
        json.dump(reference_kats(), f, indent=1)
    gv = golden_vectors()
    with open(os.path.join(HERE, "golden_vectors.json"), "w") as f:
        json.dump(gv, f, separators=(",", ":"))
    print("field %d, point %d, multiply %d, double_mul %d" % (len(gv["field"]), len(gv["point"]),
                                                             len(gv["multiply"]), len(gv["double_mul"])))
