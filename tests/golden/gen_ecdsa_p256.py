"""
Generates tests/golden/ecdsa_p256_vectors.json: Ecdsa::<P256, D>::verify fixtures (inputs + expected
status) and P-256 scalar-field products from the independent Python model oracle/py_model.py
(restatement-derived; not reference-executed).

  python tests/golden/gen_ecdsa_p256.py

"scalar_mul": {"a", "b", "mul"} limbs -- operands chosen to take every branch of reduce_wide
(p256.rs:924-1020): no second round, a second round, its carry round.
"verify": {"digest" hex, "r", "s", "pk" (x||y limbs), "pk_inf", "status"}: random signatures (status 0),
signatures that VERIFY under the reference's arithmetic (public key at infinity: R = multiply(G, h*s^-1)
does not depend on r, so r is set to the x the reference derives), r = 0, s = 0, r and s >= n (which the
default ct_lt lets through), digests >= n (status 2: the reference panics), h = 0.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import py_model as M  # noqa: E402

S = M.P256Scalar
W = 1 << 256


def main():
    rng = random.Random(0xEC0256)
    out = {"provenance": "restatement-derived by oracle/py_model.py; not reference-executed", "scalar_mul": [], "verify": []}
    n = S.N
    vals = [0, 1, 2, (1 << 64) - 1, 1 << 128, n - 1, n, n + 1, W - 1, 1 << 255, W - n]
    pairs = [(a, b) for a in vals for b in vals[::2]] + [(rng.randrange(W), rng.randrange(W)) for _ in range(60)]
    for a, b in pairs:
        out["scalar_mul"].append({"a": S.limbs(a), "b": S.limbs(b), "mul": S.mul(S.limbs(a), S.limbs(b))})

    def case(digest, r, s, pk, inf):
        st = M.p256_ecdsa_verify(digest, r, s, pk, pk_inf=inf)
        out["verify"].append({"digest": digest.hex(), "r": r, "s": s, "pk": pk, "pk_inf": int(inf), "status": st})
        return st

    def rnd_case(**kw):
        d = kw.get("digest", rng.randrange(n).to_bytes(32, "big"))
        r = kw.get("r", S.limbs(rng.randrange(1, n)))
        s = kw.get("s", S.limbs(rng.randrange(1, n)))
        pk = S.limbs(rng.randrange(W)) + S.limbs(rng.randrange(W))
        return case(d, r, s, pk, kw.get("inf", False))

    for _ in range(3):
        rnd_case()
    made = 0
    while made < 4:
        d = rng.randrange(n).to_bytes(32, "big")
        s = S.limbs(rng.randrange(1, n))
        h, _ = S.from_bytes_be(list(d))
        u1 = S.mul(h, S.inv(s))
        rp = M.P256c.multiply(M.P256c.generator(), u1)
        if M.P256c.is_identity(rp):
            continue
        x, _, _ = M.P256c.to_affine(rp)
        if S.val(x) >= n or S.val(x) == 0:
            continue
        pk = S.limbs(rng.randrange(W)) + S.limbs(rng.randrange(W))
        assert case(d, list(x), s, pk, True) == 1
        made += 1
    rnd_case(r=[0, 0, 0, 0])
    rnd_case(s=[0, 0, 0, 0])
    rnd_case(r=S.limbs(n))
    rnd_case(r=S.limbs(W - 1))
    rnd_case(s=S.limbs(n + 5))
    rnd_case(s=S.limbs(W - 1))
    assert rnd_case(digest=b"\xff" * 32) == 2
    assert rnd_case(digest=n.to_bytes(32, "big")) == 2
    rnd_case(digest=(n - 1).to_bytes(32, "big"))
    rnd_case(digest=b"\x00" * 32)
    rnd_case(inf=True)
    rnd_case(digest=b"\xff" * 32, r=[0, 0, 0, 0])  # the zero check comes before the unwrap: 0, not 2
    with open(os.path.join(HERE, "ecdsa_p256_vectors.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(len(out["scalar_mul"]), "products,", len(out["verify"]), "verifications; statuses",
          sorted(set(c["status"] for c in out["verify"])))


if __name__ == "__main__":
    main()
