"""
canon_model.py -- big-integer model of the REAL secp256k1 group (TEST INFRASTRUCTURE ONLY).

The checker for the canonical-math mode (include/fecgpu_canon.h).  It has nothing to do with the
reference: forge-ec's secp256k1 arithmetic is not the curve, so this mode
is pinned instead by the public standard:
  * curve parameters and G: SEC 2 v2, section 2.4.1 (secp256k1);
  * multiples of G: 2G, 3G quoted below are the widely published values (3G.x is also the public
    key of BIP-340 test vector 0, secret key 3 -- the key the reference's own
    test_forge_ec/src/bin/test_standard_vectors.rs uses for its BIP-340 case);
  * self-consistency: on-curve checks, n*G = infinity, group-law identities.
Affine textbook formulas over Python integers; slow and obviously correct.
"""

P = 2**256 - 2**32 - 977
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
B = 7
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
G = (GX, GY)
INF = None

# published multiples of G (x, y)
KNOWN_MULTIPLES = {
    1: (GX, GY),
    2: (0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5,
        0x1AE168FEA63DC339A3C58419466CEAEEF7F632653266D0E1236431A950CFE52A),
    3: (0xF9308A019258C31049344F85F89D5229B531C845836F99B08601F113BCE036F9,
        0x388F7B0F632DE8140FE337E62A37F3566500A99934C2231B6CB9FD7584B8E672),
}


def on_curve(pt):
    if pt is INF:
        return True
    x, y = pt
    return 0 <= x < P and 0 <= y < P and (y * y - x * x * x - B) % P == 0


def neg(pt):
    return INF if pt is INF else (pt[0], (-pt[1]) % P)


def add(p1, p2):
    if p1 is INF:
        return p2
    if p2 is INF:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return INF
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def mul(k, pt):
    acc = INF
    while k:
        if k & 1:
            acc = add(acc, pt)
        pt = add(pt, pt)
        k >>= 1
    return acc


def field_op(op, a, b=0):
    if op == "add":
        return (a + b) % P
    if op == "sub":
        return (a - b) % P
    if op == "mul":
        return a * b % P
    if op == "sqr":
        return a * a % P
    if op == "neg":
        return (-a) % P
    if op == "inv":
        return pow(a, P - 2, P)
    raise ValueError(op)


def limbs(v):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def unlimbs(l):
    return sum(int(l[i]) << (64 * i) for i in range(4))


def xy_limbs(pt):
    return [0] * 8 if pt is INF else limbs(pt[0]) + limbs(pt[1])
