"""Seeded synthetic inputs shared by bench.py, the parity tests and the golden generator (host-side,
numpy only).

SplitMix64 counter PRNG: CPU and GPU sides regenerate identical data from
(seed, stream).  Scalars: 4 uniform limbs made canonical the way the reference's Scalar::random
does (one conditional subtraction of the group order; Ed25519: field-style reduce, i.e. < 2^255-19
after clearing bit 255) with zero rejected.  Points: independent uniform canonical field elements
per coordinate -- legitimate because Curve::multiply never validates its input point
(secp256k1.rs:2635-2639, p256.rs:2120-2124, ed25519.rs:2062-2066).
"""
import numpy as np

SEED = 0xF0E1D2C3B4A59687
MASK64 = (1 << 64) - 1

PRIME = {
    0: (1 << 256) - (1 << 32) - 977,
    1: (1 << 256) - (1 << 224) + (1 << 192) + (1 << 96) - 1,
    2: (1 << 255) - 19,
}
ORDER = {
    0: 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
    1: 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551,
    2: (1 << 252) + 27742317777372353535851937790883648493,
}
POINT_LIMBS = {0: 12, 1: 12, 2: 16}


def splitmix64(n, seed, stream):
    """n uint64 values of stream `stream` (vectorised SplitMix64)."""
    with np.errstate(over="ignore"):
        start = np.uint64((seed + 0x9E3779B97F4A7C15 * (stream * 0x10000000 + 1)) & MASK64)
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = start + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def limbs_of(x):
    return [(x >> (64 * i)) & MASK64 for i in range(4)]


def int_of(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def _lt(a, m):
    """rows of a (n,4) < 256-bit constant m, vectorised lexicographic compare."""
    ml = limbs_of(m)
    lt = np.zeros(a.shape[0], dtype=bool)
    eq = np.ones(a.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        lt |= eq & (a[:, i] < np.uint64(ml[i]))
        eq &= a[:, i] == np.uint64(ml[i])
    return lt


def _sub_const(a, m):
    """a - m mod 2^256, vectorised."""
    ml = limbs_of(m)
    out = np.empty_like(a)
    borrow = np.zeros(a.shape[0], dtype=np.uint64)
    with np.errstate(over="ignore"):
        for i in range(4):
            mi = np.uint64(ml[i])
            d1 = a[:, i] - mi
            b1 = (a[:, i] < mi).astype(np.uint64)
            d2 = d1 - borrow
            b2 = (d1 < borrow).astype(np.uint64)
            out[:, i] = d2
            borrow = b1 | b2
    return out


def field_elements(n, curve, stream, seed=SEED):
    """(n,4) uniform canonical field elements (limbs drawn uniformly, top bits masked for
    Ed25519, rejected-by-resampling if >= p)."""
    p = PRIME[curve]
    a = splitmix64(4 * n, seed, stream).reshape(n, 4).copy()
    if curve == 2:
        a[:, 3] &= np.uint64(0x7FFFFFFFFFFFFFFF)
    bad = ~_lt(a, p)
    k = 1
    while bad.any():
        r = splitmix64(4 * int(bad.sum()), seed, stream + 7919 * k).reshape(-1, 4)
        if curve == 2:
            r[:, 3] &= np.uint64(0x7FFFFFFFFFFFFFFF)
        a[bad] = r
        bad = ~_lt(a, p)
        k += 1
    return a


def scalars(n, curve, stream, seed=SEED):
    """(n,4) scalars, canonical as Scalar::random leaves them, never zero."""
    a = splitmix64(4 * n, seed, stream).reshape(n, 4).copy()
    if curve == 2:
        # ed25519.rs:939-964: field-style reduce() once -> value < 2^255-19, possibly >= l
        a[:, 3] &= np.uint64(0x7FFFFFFFFFFFFFFF)
        ge = ~_lt(a, PRIME[2])
        a[ge] = _sub_const(a[ge], PRIME[2])
    else:
        ge = ~_lt(a, ORDER[curve])
        a[ge] = _sub_const(a[ge], ORDER[curve])
    zero = (a == 0).all(axis=1)
    a[zero, 0] = np.uint64(1)
    return a


def points(n, curve, stream, seed=SEED):
    """(n, limbs) points with independent uniform canonical coordinates."""
    nc = POINT_LIMBS[curve] // 4
    cols = [field_elements(n, curve, stream * 16 + c + 1, seed) for c in range(nc)]
    return np.ascontiguousarray(np.concatenate(cols, axis=1))


def edge_field_values(curve):
    p = PRIME[curve]
    vals = [0, 1, 2, 3, 8, 19, 38, 977, p - 2, p - 1, p, p + 1, (1 << 256) - 1, 1 << 255, (1 << 255) - 19,
            (1 << 255) - 1, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, 1 << 192, 1 << 224, (1 << 256) - p,
            (1 << 256) - p - 1, (1 << 256) - 2, 0xFFFFFFFF00000000FFFFFFFF00000000FFFFFFFF00000000FFFFFFFF00000000,
            0xFFFFFFFFFFFFFFFF0000000000000000FFFFFFFFFFFFFFFF0000000000000000,
            0x00000000FFFFFFFF00000000FFFFFFFF00000000FFFFFFFF00000000FFFFFFFF,
            # operands whose sums / differences send the carry of the short chains (+- c on words 0..1, +- 19 on
            # word 0: limbs.hpp) through the remaining words
            12, (1 << 255) + 5, (1 << 255) + 7, (1 << 64) - 2, (1 << 32) - 19, (1 << 32) - 3, (1 << 255) - 20,
            (1 << 256) - (1 << 64), (1 << 256) - (1 << 32) - 5]
    return [v for v in vals if 0 <= v < (1 << 256)]
