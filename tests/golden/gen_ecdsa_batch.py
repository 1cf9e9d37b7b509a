"""
Generates tests/golden/ecdsa_batch_vectors.json: Ecdsa::<C, D>::batch_verify fixtures (ecdsa.rs:287-391) for
secp256k1 and P-256 -- inputs, expected status and the two folded sums -- from the independent Python model
oracle/py_model.py (restatement-derived; not reference-executed).

  python tests/golden/gen_ecdsa_batch.py

Per curve: random signatures (the final comparison fails), a batch that VERIFIES under the reference's
arithmetic (all public keys at infinity, so r_sum does not depend on r; the last weight is 1 and the last r is
chosen so that the ordered scalar sum equals x(r_sum)), an early `false` (r = 0 in the middle), a panic
(digest >= n) before a later r = 0 and the reverse order, and n = 1.
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
ORDER = {0: 0xFFFFFFFFFFFFFFFEFFFFFFFFFFFFFFFFBAAEDCE6AF48A03BBFD25E8CD0364141,   # the reference's secp256k1 N (limbs swapped)
         1: M.P256Scalar.N}


def limbs(v):
    return [(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]


def val(l):
    return sum(int(x) << (64 * i) for i, x in enumerate(l))


def main():
    rng = random.Random(0xBA7C4)
    out = {"provenance": "restatement-derived by oracle/py_model.py; not reference-executed", "cases": []}

    def emit(curve, dg, r, s, pk, inf, a, note):
        st, rs, tot = M.ecdsa_batch_verify(curve, dg, r, s, pk, inf, a)
        out["cases"].append({"curve": curve, "note": note, "digests": [d.hex() for d in dg], "r": r, "s": s, "pk": pk,
                             "pk_inf": inf, "a": a, "status": st,
                             "r_sum": [x for c in rs for x in c] if rs else None, "scalar_sum": tot})
        return st, rs, tot

    for curve in (0, 1):
        n_ord = ORDER[curve]
        S, F = (M.SecpScalar, M.Secp) if curve == 0 else (M.P256Scalar, M.P256c)

        def draw(n):
            dg = [rng.randrange(n_ord).to_bytes(32, "big") for _ in range(n)]
            r = [limbs(rng.randrange(1, n_ord)) for _ in range(n)]
            s = [limbs(rng.randrange(1, n_ord)) for _ in range(n)]
            pk = [limbs(rng.randrange(W)) + limbs(rng.randrange(W)) for _ in range(n)]
            a = [limbs(rng.randrange(1, n_ord)) for _ in range(n)]
            return dg, r, s, pk, a

        dg, r, s, pk, a = draw(3)
        assert emit(curve, dg, r, s, pk, [0, 0, 0], a, "random: final comparison fails")[0] == 0
        # a batch that verifies
        for n in (3, 1):
            while True:
                dg, r, s, pk, a = draw(n)
                a[n - 1] = [1, 0, 0, 0]
                st, rs, tot = M.ecdsa_batch_verify(curve, dg, r, s, pk, [1] * n, a)
                if rs is None or F.is_identity(rs):
                    continue
                x, _, _ = F.to_affine(rs)
                xs = val(F.mul(x, [1, 0, 0, 0])) if curve == 0 else val(x)
                partial = [0, 0, 0, 0]
                for i in range(n - 1):
                    partial = S.add(partial, S.mul(a[i], r[i]))
                if 0 < xs < n_ord and xs > val(partial):
                    r[n - 1] = limbs(xs - val(partial))
                    break
            assert emit(curve, dg, r, s, pk, [1] * n, a, "verifies under the reference's arithmetic, n = %d" % n)[0] == 1
        dg, r, s, pk, a = draw(4)
        r[2] = [0, 0, 0, 0]
        assert emit(curve, dg, r, s, pk, [0] * 4, a, "r = 0 at index 2: false at 317-319")[0] == 0
        dg, r, s, pk, a = draw(4)
        dg[1] = b"\xff" * 32
        r[3] = [0, 0, 0, 0]
        assert emit(curve, dg, r, s, pk, [0] * 4, a, "digest >= n at index 1 before r = 0 at index 3: panics")[0] == 2
        dg, r, s, pk, a = draw(4)
        s[0] = [0, 0, 0, 0]
        dg[2] = b"\xff" * 32
        assert emit(curve, dg, r, s, pk, [0] * 4, a, "s = 0 at index 0 before digest >= n at index 2: false")[0] == 0
    with open(os.path.join(HERE, "ecdsa_batch_vectors.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(len(out["cases"]), "cases; statuses", [c["status"] for c in out["cases"]])


if __name__ == "__main__":
    main()
