"""
Generates tests/golden/forcing_vectors.json: operands built to FORCE the rare continuations of the
device field arithmetic, with expected results from the independent Python model oracle/py_model.py
(restatement-derived, not reference-executed -- same provenance as golden_vectors.json).

  python tests/golden/gen_forcing.py

Families (each entry: curve, op, a, b, expect):
  p256_noncanonical   P-256 operands >= p / with an all-ones top word: the reference's Sub (p256.rs:470-496)
                      produces such values about once per 2^20 scalar-muls; Add's `while` loops (436-464),
                      Sub's `a < b` branch on a non-canonical a, Mul/square of non-canonical inputs
  secp_mul_borrow     secp256k1 Mul (442-507) operands for which V = T_hi + M - Q borrows out of word 1
                      (the device continues that borrow behind a rare branch): a = M*c mod 2^256, b = 1
                      with the low 64 bits of M smaller than Q
  secp_mul_ge_p       secp256k1 Mul / square results that need the closing conditional subtraction
  ed_small_add_carry  Ed25519 reduce_wide (260-289): the carry*19 addition, and the +19 for bit 255, carry
                      out of word 0; values that reach p after the fold
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

L = V.limbs_of
W = 1 << 256


def entry(curve, op, a, b=None):
    F = M.CURVES[curve]
    la, lb = L(a % W), (L(b % W) if b is not None else None)
    fn = getattr(F, op)
    exp = fn(la, lb) if lb is not None else fn(la)
    return {"curve": curve, "op": op, "a": la, "b": lb, "expect": exp}


def main():
    rng = random.Random(0xF02C1A6)
    out = {"provenance": "restatement-derived by oracle/py_model.py; not reference-executed", "cases": []}
    add = out["cases"].append
    # ---- P-256 non-canonical operands ----
    p = V.PRIME[1]
    nc = [p, p + 1, p + 2, W - 1, W - 2, W - (1 << 224), W - (1 << 224) + 1, (0xFFFFFFFF << 224) | rng.getrandbits(224),
          (0xFFFFFFFF << 224) | rng.getrandbits(224), (0xFFFFFFFF << 224), W - (1 << 96), W - p, W - p + 1,
          ((1 << 256) - (1 << 224)) + (1 << 192) + (1 << 96) - 2]
    can = [0, 1, 2, p - 1, p - 2, rng.randrange(p), rng.randrange(p), (1 << 224) - 1, 1 << 255]
    for a in nc:
        for b in nc[::2] + can[::2]:
            for op in ("add", "sub", "mul"):
                add(dict(entry(1, op, a, b), family="p256_noncanonical"))
                add(dict(entry(1, op, b, a), family="p256_noncanonical"))
        add(dict(entry(1, "sqr", a), family="p256_noncanonical"))
        add(dict(entry(1, "neg", a), family="p256_noncanonical"))
    # ---- secp256k1 Mul: borrow out of word 1 of T_hi + M - Q ----
    c = (1 << 32) + 977
    ps = V.PRIME[0]
    for r in range(48):
        m = rng.getrandbits(256) | (3 << 254)
        m = (m >> 64 << 64) | (r & 7)          # low 64 bits of M tiny, Q = floor(M*c / 2^256) ~ 2^32
        if r & 8:
            m &= ~(((1 << 128) - 1) << 64) | 0  # words 2..5 zero: the borrow ripples further
        a = (m * c) % W
        add(dict(entry(0, "mul", a, 1), family="secp_mul_borrow"))
        add(dict(entry(0, "mul", 1, a), family="secp_mul_borrow"))
    # ---- secp256k1: results with words 2..7 all ones (closing reduce) -- searched on the model ----
    F0 = M.CURVES[0]
    found = 0
    for a in [ps - 1, ps - 2, W - 1, ps, ps + 1, W - c, W - c - 1, W - c + 1]:
        for b in [1, 2, 3, ps - 1, W - 1, c, c + 1, (1 << 256) - (1 << 32)]:
            add(dict(entry(0, "mul", a, b), family="secp_mul_ge_p"))
        add(dict(entry(0, "sqr", a), family="secp_mul_ge_p"))
    # ---- Ed25519 reduce_wide small additions ----
    inv19 = pow(19, -1, 1 << 31)
    a1 = (-inv19) % (1 << 31)                 # 38*a1 = -2 mod 2^32
    for hi_fill in (0xFFFFFFFF, 0xFFFFFFF0, 0x80000000):
        a = 0xFFFFFFFF | (a1 << 32) | (((hi_fill << 160) | rng.getrandbits(160)) << 64)
        add(dict(entry(2, "mul", a, 1 << 224), family="ed_small_add_carry"))
        add(dict(entry(2, "mul", 1 << 224, a), family="ed_small_add_carry"))
    pe = V.PRIME[2]
    for k in range(0, 24):
        a = (1 << 255) + (1 << 32) - 1 - k    # bit 255 set, word 0 within 19 of wrapping
        add(dict(entry(2, "mul", a, 1), family="ed_small_add_carry"))
        add(dict(entry(2, "sqr", a), family="ed_small_add_carry"))
    for a in [pe - 1, pe, pe + 1, pe - 19, pe - 20, (1 << 255) - 1, (1 << 255) - 20, W - 1, W - 19, W - 38, W - 39]:
        for b in [1, 2, 19, 38, pe - 1, W - 1]:
            add(dict(entry(2, "mul", a, b), family="ed_small_add_carry"))
        add(dict(entry(2, "sqr", a), family="ed_small_add_carry"))
        add(dict(entry(2, "add", a, 1), family="ed_small_add_carry"))
        add(dict(entry(2, "sub", 1, a), family="ed_small_add_carry"))
    with open(os.path.join(HERE, "forcing_vectors.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    fam = {}
    for e in out["cases"]:
        fam[e["family"]] = fam.get(e["family"], 0) + 1
    print(fam)


if __name__ == "__main__":
    main()
