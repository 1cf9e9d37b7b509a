"""
crosscheck.py -- agreement of the two independent restatements (C vs Python).

  python -m oracle.crosscheck [n_field] [n_mul]

Field ops: n_field random pairs per op per curve, plus edge operands (0, 1, p-1, p, p+1,
2^256-1, single-limb values) so the non-canonical paths (P-256 Sub, while-loops) are hit.
Scalar-muls: n_mul random (point, scalar) per curve, plus small scalars.
"""
import random
import sys

from oracle import c_oracle as C
from oracle import py_model as M

MASK = (1 << 64) - 1


def limbs(x):
    return [(x >> (64 * i)) & MASK for i in range(4)]


def edge_values(p):
    vals = [0, 1, 2, 3, 8, 19, 38, 977, p - 2, p - 1, p, p + 1, (1 << 256) - 1, (1 << 255), (1 << 255) - 19,
            (1 << 255) - 1, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, 1 << 192, (1 << 224), (1 << 256) - p,
            (1 << 256) - p - 1, 0xFFFFFFFF00000000FFFFFFFF00000000FFFFFFFF00000000FFFFFFFF00000000]
    return [v for v in vals if 0 <= v < (1 << 256)]


PRIMES = {0: (1 << 256) - (1 << 32) - 977, 1: (1 << 256) - (1 << 224) + (1 << 192) + (1 << 96) - 1,
          2: (1 << 255) - 19}


def main():
    n_field = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    n_mul = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rng = random.Random(0xF0E1D2C3)
    bad = 0
    for curve, F in M.CURVES.items():
        p = PRIMES[curve]
        edges = edge_values(p)
        pairs = [(a, b) for a in edges for b in edges]
        for _ in range(n_field):
            kind = rng.random()
            if kind < 0.6:
                a, b = rng.randrange(p), rng.randrange(p)
            elif kind < 0.8:
                a, b = rng.randrange(1 << 256), rng.randrange(1 << 256)   # non-canonical operands
            else:
                a = rng.choice(edges) ^ rng.randrange(1 << 16)
                b = rng.randrange(1 << 256) if rng.random() < 0.5 else rng.choice(edges)
                a &= (1 << 256) - 1
            pairs.append((a, b))
        for a, b in pairs:
            la, lb = limbs(a), limbs(b)
            for op, fn in (("add", F.add), ("sub", F.sub), ("mul", F.mul)):
                # the reference's u128 sums in secp256k1 Mul only fit for any 64-bit limbs; all inputs are legal
                want = fn(la, lb)
                got = [int(v) for v in C.field_op(curve, op, la, lb)]
                if want != got:
                    bad += 1
                    print("MISMATCH", curve, op, hex(a), hex(b), want, got)
            for op, fn in (("sqr", F.sqr), ("neg", F.neg)):
                want = fn(la)
                got = [int(v) for v in C.field_op(curve, op, la)]
                if want != got:
                    bad += 1
                    print("MISMATCH", curve, op, hex(a), want, got)
        for a in [1, 2, 5, p - 1, rng.randrange(p), rng.randrange(p)]:
            want = F.inv(limbs(a))
            got = [int(v) for v in C.field_op(curve, "inv", limbs(a))]
            if want != got:
                bad += 1
                print("MISMATCH inv", curve, hex(a))
        print("curve %d: field ops agree on %d operand pairs" % (curve, len(pairs)), flush=True)

        g = F.generator()
        assert M.flat(g) == [int(v) for v in C.generator(curve)]
        ncoord = 3 if curve != 2 else 4
        cases = [(g, limbs(k)) for k in (1, 2, 3, 5, 1 << 255, (1 << 256) - 1)]
        for _ in range(n_mul):
            pt = tuple(limbs(rng.randrange(p)) for _ in range(ncoord))
            cases.append((pt, limbs(rng.randrange(1, 1 << 256))))
        for pt, k in cases:
            want = M.flat(F.multiply(pt, k))
            got = [int(v) for v in C.multiply(curve, M.flat(pt), k)]
            if want != got:
                bad += 1
                print("MISMATCH multiply", curve, pt, k)
        # point add / double incl. degenerate operands
        p2 = F.double(g)
        pts = [g, p2, F.identity(), F.padd(g, p2)]
        for a in pts:
            for b in pts:
                want = M.flat(F.padd(a, b))
                got = [int(v) for v in C.point_add(curve, M.flat(a), M.flat(b))]
                if want != got:
                    bad += 1
                    print("MISMATCH padd", curve)
            want = M.flat(F.double(a))
            got = [int(v) for v in C.point_double(curve, M.flat(a))]
            if want != got:
                bad += 1
                print("MISMATCH double", curve)
        print("curve %d: %d scalar-muls + point ops agree" % (curve, len(cases)), flush=True)
    print("TOTAL MISMATCHES:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
