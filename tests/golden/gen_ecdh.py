"""
Generates tests/golden/ecdh_vectors.json: KeyExchange::derive_shared_secret fixtures (secp256k1.rs:1884-1904,
p256.rs:2281-2312) -- private key, public key, expected status and secret -- from the independent Python model
oracle/py_model.py (restatement-derived; not reference-executed).

  python tests/golden/gen_ecdh.py

secp256k1 (no validation): arbitrary coordinates, the generator, a key flagged infinite (the product is the
identity: Err), a zero private key (identity as well).  P-256: true curve points -- about half of which the
reference's is_on_curve accepts (its Sub leaves x^3 - 3x + b non-canonical for the rest) -- off-curve
coordinates and an infinite key (Err(InvalidPublicKey)), a zero private key on an accepted point (Err).
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


def main():
    rng = random.Random(0xECD4)
    out = {"provenance": "restatement-derived by oracle/py_model.py; not reference-executed", "cases": []}

    def emit(curve, sk, pk, inf, note):
        st, sec = M.ecdh(curve, sk, pk, inf)
        out["cases"].append({"curve": curve, "note": note, "sk": sk, "pk": pk, "pk_inf": int(inf), "status": st, "secret": sec.hex()})
        return st

    g = M.Secp.to_affine(M.Secp.generator())
    emit(0, limbs(rng.randrange(1, W)), limbs(rng.randrange(W)) + limbs(rng.randrange(W)), False, "arbitrary coordinates (no validation)")
    emit(0, limbs(rng.randrange(1, W)), list(g[0]) + list(g[1]), False, "generator as public key")
    assert emit(0, limbs(rng.randrange(1, W)), limbs(5) + limbs(7), True, "key flagged infinite: identity product") == 2
    assert emit(0, [0, 0, 0, 0], list(g[0]) + list(g[1]), False, "zero private key: identity product") == 2
    acc = rej = 0
    while acc < 3 or rej < 2:
        pk = p256_point(rng)
        st = M.ecdh(1, limbs(3), pk, False)[0]
        if st == 0 and acc < 3:
            emit(1, limbs(rng.randrange(1, W)), pk, False, "true curve point the reference accepts")
            if acc == 0:
                assert emit(1, [0, 0, 0, 0], pk, False, "zero private key on an accepted point") == 2
            acc += 1
        elif st == 1 and rej < 2:
            assert emit(1, limbs(rng.randrange(1, W)), pk, False, "true curve point the reference REJECTS (non-canonical rhs)") == 1
            rej += 1
    assert emit(1, limbs(rng.randrange(1, W)), limbs(rng.randrange(W)) + limbs(rng.randrange(W)), False, "off-curve coordinates") == 1
    assert emit(1, limbs(rng.randrange(1, W)), p256_point(rng), True, "infinite key") == 1
    with open(os.path.join(HERE, "ecdh_vectors.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(len(out["cases"]), "cases; statuses", [(c["curve"], c["status"]) for c in out["cases"]])


if __name__ == "__main__":
    main()
