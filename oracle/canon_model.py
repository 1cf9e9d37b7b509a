"""
canon_model.py -- big-integer model of the REAL curves (TEST INFRASTRUCTURE ONLY).

The checker for the canonical-math mode (include/fecgpu_canon.h).  It has nothing to do with the
reference: forge-ec's curve arithmetic is not the curves, so this mode is
pinned instead by public standards:
  * secp256k1: parameters and G from SEC 2 v2 section 2.4.1; 2G, 3G are the widely published
    values (3G.x is also the public key of BIP-340 test vector 0, secret key 3 -- the key the
    reference's own test_forge_ec/src/bin/test_standard_vectors.rs uses for its BIP-340 case);
  * P-256: parameters from FIPS 186-4 D.1.2.3; the key pair of RFC 6979 A.2.5 (private key
    C9AFA9D8..., the one test_standard_vectors.rs quotes for its FIPS 186-4 ECDSA case) with its
    public key (Ux, Uy) as printed in the RFC;
  * self-consistency: on-curve checks, n*G = infinity, group-law identities.
Affine textbook formulas over Python integers; slow and obviously correct.
"""

INF = None


class Weierstrass:
    def __init__(self, name, p, a, b, gx, gy, n, known):
        self.name, self.P, self.A, self.B, self.N = name, p, a % p, b, n
        self.G = (gx, gy)
        self.KNOWN_MULTIPLES = known

    def on_curve(self, pt):
        if pt is INF:
            return True
        x, y = pt
        return 0 <= x < self.P and 0 <= y < self.P and (y * y - x * x * x - self.A * x - self.B) % self.P == 0

    def neg(self, pt):
        return INF if pt is INF else (pt[0], (-pt[1]) % self.P)

    def add(self, p1, p2):
        P = self.P
        if p1 is INF:
            return p2
        if p2 is INF:
            return p1
        x1, y1 = p1
        x2, y2 = p2
        if x1 == x2:
            if (y1 + y2) % P == 0:
                return INF
            lam = (3 * x1 * x1 + self.A) * pow(2 * y1, -1, P) % P
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
        x3 = (lam * lam - x1 - x2) % P
        return (x3, (lam * (x1 - x3) - y1) % P)

    def mul(self, k, pt):
        acc = INF
        while k:
            if k & 1:
                acc = self.add(acc, pt)
            pt = self.add(pt, pt)
            k >>= 1
        return acc

    def field_op(self, op, a, b=0):
        P = self.P
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


SECP256K1 = Weierstrass(
    "secp256k1", 2**256 - 2**32 - 977, 0, 7,
    0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
    0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8,
    0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
    {
        1: (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
            0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8),
        2: (0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5,
            0x1AE168FEA63DC339A3C58419466CEAEEF7F632653266D0E1236431A950CFE52A),
        3: (0xF9308A019258C31049344F85F89D5229B531C845836F99B08601F113BCE036F9,
            0x388F7B0F632DE8140FE337E62A37F3566500A99934C2231B6CB9FD7584B8E672),
    })

P256 = Weierstrass(
    "p256", 2**256 - 2**224 + 2**192 + 2**96 - 1, -3,
    0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B,
    0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296,
    0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5,
    0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551,
    {
        1: (0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296,
            0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5),
        # RFC 6979 A.2.5: private key x, public key (Ux, Uy)
        0xC9AFA9D845BA75166B5C215767B1D6934E50C3DB36E89B127B8A622B120F6721:
            (0x60FED4BA255A9D31C961EB74C6356D68C049B8923B61FA6CE669622E60F29FB6,
             0x7903FE1008B8BC99A41AE9E95628BC64F2F1B20C2D7E9F5177A3C294D4462299),
    })



class TwistedEdwards:
    """-x^2 + y^2 = 1 + d x^2 y^2 over p (RFC 8032 section 5.1); complete affine addition law."""

    def __init__(self, name, p, d, gx, gy, order):
        self.name, self.P, self.D, self.N = name, p, d, order
        self.G = (gx, gy)
        self.IDENTITY = (0, 1)

    def on_curve(self, pt):
        x, y = pt
        P = self.P
        return 0 <= x < P and 0 <= y < P and (-x * x + y * y - 1 - self.D * x * x * y * y) % P == 0

    def neg(self, pt):
        return ((-pt[0]) % self.P, pt[1])

    def add(self, p1, p2):
        P, d = self.P, self.D
        x1, y1 = p1
        x2, y2 = p2
        k = d * x1 * x2 * y1 * y2 % P
        x3 = (x1 * y2 + x2 * y1) * pow(1 + k, -1, P) % P
        y3 = (y1 * y2 + x1 * x2) * pow(1 - k, -1, P) % P
        return (x3, y3)

    def mul(self, k, pt):
        acc = self.IDENTITY
        while k:
            if k & 1:
                acc = self.add(acc, pt)
            pt = self.add(pt, pt)
            k >>= 1
        return acc

    def field_op(self, op, a, b=0):
        return Weierstrass.field_op(self, op, a, b)

    def encode(self, pt):
        """RFC 8032 section 5.1.2: 32 bytes, y little-endian with the sign of x in the top bit."""
        x, y = pt
        return (y | ((x & 1) << 255)).to_bytes(32, "little")

    def secret_scalar(self, seed32):
        """RFC 8032 section 5.1.5: the clamped lower half of SHA-512(seed)."""
        import hashlib
        h = bytearray(hashlib.sha512(seed32).digest()[:32])
        h[0] &= 248
        h[31] &= 127
        h[31] |= 64
        return int.from_bytes(bytes(h), "little")


_EDP = 2**255 - 19
ED25519 = TwistedEdwards(
    "ed25519", _EDP, (-121665 * pow(121666, -1, _EDP)) % _EDP,
    0x216936D3CD6E53FEC0A4E231FDD6DC5C692CC7609525A7B2C9562D608F25D51A,
    0x6666666666666666666666666666666666666666666666666666666666666658,
    2**252 + 27742317777372353535851937790883648493)
# RFC 8032 section 7.1, TEST 1 (the vector test_forge_ec/src/bin/test_standard_vectors.rs quotes): secret seed -> public key
ED25519_RFC8032_TEST1 = (bytes.fromhex("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60"),
                         bytes.fromhex("d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a"))
# TEST 2
ED25519_RFC8032_TEST2 = (bytes.fromhex("4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb"),
                         bytes.fromhex("3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c"))

CURVES = {"secp256k1": SECP256K1, "p256": P256}


def limbs(v):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def unlimbs(l):
    return sum(int(l[i]) << (64 * i) for i in range(4))


def xy_limbs(pt):
    return [0] * 8 if pt is INF else limbs(pt[0]) + limbs(pt[1])


# ---- standard SIGNATURE vectors, reduced to the point equation u1*G + u2*P the GPU evaluates ----
# Each helper returns (curve, u1, u2, P, check) with check(R) -> bool for the affine result R.
def ecdsa_p256_rfc6979_sample():
    """RFC 6979 A.2.5, P-256 + SHA-256, message "sample" (the key of test_standard_vectors.rs's FIPS case;
    that file's own r/s strings are truncated to 63/62 hex digits, so the RFC's are used)."""
    import hashlib
    C = P256
    d = 0xC9AFA9D845BA75166B5C215767B1D6934E50C3DB36E89B127B8A622B120F6721
    r = 0xEFD48B2AACB6A8FD1140DD9CD45E81D69D2C877B56AAF991C34D0EA84EAF3716
    s = 0xF7CB1C942D657C41D436C7A1B6E29F65F3E900DBB9AFF4064DC4AB2F843ACDA8
    z = int.from_bytes(hashlib.sha256(b"sample").digest(), "big")
    w = pow(s, -1, C.N)
    return C, z * w % C.N, r * w % C.N, C.KNOWN_MULTIPLES[d], lambda R: R is not INF and R[0] % C.N == r


def schnorr_bip340_vector0():
    """BIP-340 test vector 0 (secret key 3, message 0^32): the signature test_standard_vectors.rs quotes."""
    import hashlib
    C = SECP256K1
    pkx = 0xF9308A019258C31049344F85F89D5229B531C845836F99B08601F113BCE036F9
    sig = bytes.fromhex("E907831F80848D1069A5371B402410364BDF1C5F8307B0084C55F1CE2DCA8215"
                        "25F66A4A85EA8B71E482A74F382D2CE5EBEEE8FDB2172F477DF4900D310536C0")
    r, s = int.from_bytes(sig[:32], "big"), int.from_bytes(sig[32:], "big")
    y2 = (pow(pkx, 3, C.P) + 7) % C.P
    y = pow(y2, (C.P + 1) // 4, C.P)
    assert y * y % C.P == y2
    P = (pkx, y if y % 2 == 0 else C.P - y)      # lift_x
    t = hashlib.sha256(b"BIP0340/challenge").digest()
    e = int.from_bytes(hashlib.sha256(t + t + sig[:32] + pkx.to_bytes(32, "big") + bytes(32)).digest(), "big") % C.N
    return C, s, (C.N - e) % C.N, P, lambda R: R is not INF and R[1] % 2 == 0 and R[0] == r


def ed25519_decode(b):
    """RFC 8032 section 5.1.3"""
    E = ED25519
    v = int.from_bytes(b, "little")
    sign, y = v >> 255, v & ((1 << 255) - 1)
    p = E.P
    x2 = (y * y - 1) * pow(E.D * y * y + 1, -1, p) % p
    x = pow(x2, (p + 3) // 8, p)
    if (x * x - x2) % p:
        x = x * pow(2, (p - 1) // 4, p) % p
    assert (x * x - x2) % p == 0 and y < p
    if x & 1 != sign:
        x = p - x
    return (x, y)


def eddsa_rfc8032_test1():
    """RFC 8032 section 7.1 TEST 1 (empty message): the signature test_standard_vectors.rs quotes.
    S*B == R + h*A  <=>  S*B + (l - h)*A == R."""
    import hashlib
    E = ED25519
    _, pk = ED25519_RFC8032_TEST1
    sig = bytes.fromhex("e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e06522490155"
                        "5fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b")
    A, Rp, S = ed25519_decode(pk), ed25519_decode(sig[:32]), int.from_bytes(sig[32:], "little")
    h = int.from_bytes(hashlib.sha512(sig[:32] + pk + b"").digest(), "little") % E.N
    return E, S, (E.N - h) % E.N, A, lambda R: R == Rp


SIGNATURE_VECTORS = {"p256": ecdsa_p256_rfc6979_sample, "secp256k1": schnorr_bip340_vector0,
                     "ed25519": eddsa_rfc8032_test1}
