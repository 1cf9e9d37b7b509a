"""
py_model.py -- second, independently written CPU restatement (TEST INFRASTRUCTURE ONLY).

Written from the Rust sources (forge-ec-curves/src/{secp256k1,p256,ed25519}.rs), not
from oracle/forge_ec_oracle.c, so that agreement between the two pins the
restatement ("agreement of two independently written
transliterations").  Pure Python integers stand in for u64/u128/i128 with explicit
masks; loops mirror the reference's loops.  Slow (about 50 ms per secp256k1
scalar-mul): used for small cases and to emit tests/golden/*.json.

Nothing under forge_ec_amd/ may import this module.
"""
M64 = (1 << 64) - 1
M128 = (1 << 128) - 1

SECP256K1, P256, ED25519 = 0, 1, 2


def _is_zero(a):
    return a[0] == 0 and a[1] == 0 and a[2] == 0 and a[3] == 0


# ======================================================================================
# secp256k1 (secp256k1.rs)
# ======================================================================================
class Secp:
    P = [0xFFFFFFFEFFFFFC2F, M64, M64, M64]  # :22-23
    N0 = 0xD838091DD2253531  # :468

    @staticmethod
    def cmp_p(l):  # :47-75
        res = 0
        for i in (3, 2, 1, 0):
            if res != 0:
                continue
            if l[i] < Secp.P[i]:
                res = -1
            elif l[i] > Secp.P[i]:
                res = 1
        return res

    @staticmethod
    def _minus_p(l):
        out, borrow = [0] * 4, 0
        for i in range(4):
            d1 = (l[i] - Secp.P[i]) & M64
            b1 = 1 if l[i] < Secp.P[i] else 0
            d2 = (d1 - borrow) & M64
            b2 = 1 if d1 < borrow else 0
            out[i] = d2
            borrow = b1 | b2
        return out

    @staticmethod
    def reduce(l):  # :78-102
        return Secp._minus_p(l) if Secp.cmp_p(l) >= 0 else list(l)

    @staticmethod
    def add(a, b):  # :353-393
        r, carry = list(a), 0
        for i in range(4):
            s1 = (r[i] + b[i]) & M64
            s2 = (s1 + carry) & M64
            c1 = 1 if r[i] > (b[i] ^ M64) else 0
            c2 = 1 if s1 > (M64 - carry) else 0
            r[i] = s2
            carry = c1 | c2
        red = Secp._minus_p(r)
        return red if (carry > 0 or Secp.cmp_p(r) >= 0) else r

    @staticmethod
    def sub(a, b):  # :395-440
        r, borrow = list(a), 0
        for i in range(4):
            d1 = (r[i] - b[i]) & M64
            d2 = (d1 - borrow) & M64
            b1 = 1 if r[i] < b[i] else 0
            b2 = 1 if d1 < borrow else 0
            r[i] = d2
            borrow = b1 | b2
        wp, carry = list(r), 0
        for i in range(4):
            s1 = (wp[i] + Secp.P[i]) & M64
            s2 = (s1 + carry) & M64
            c1 = 1 if wp[i] > (Secp.P[i] ^ M64) else 0
            c2 = 1 if s1 > (M64 - carry) else 0
            wp[i] = s2
            carry = c1 | c2
        return wp if borrow > 0 else r

    @staticmethod
    def mul(a, b):  # :442-507
        t = [0] * 8
        for i in range(4):
            carry = 0
            for j in range(4):
                prod = a[i] * b[j] + t[i + j] + carry
                assert prod <= M128
                t[i + j] = prod & M64
                carry = prod >> 64
            t[i + 4] = carry
        carry = 0  # declared once, outside the rounds (:470)
        for i in range(4):
            m = (t[i] * Secp.N0) & M64
            s = t[i] + m * Secp.P[0] + carry
            assert s <= M128
            carry = s >> 64
            for j in range(1, 4):
                s = t[i + j] + m * Secp.P[j] + carry
                assert s <= M128
                t[i + j] = s & M64
                carry = s >> 64
            j = i + 4
            while j < 8 and carry > 0:
                s = t[j] + carry
                t[j] = s & M64
                carry = s >> 64
                j += 1
        r = t[4:8]
        if Secp.cmp_p(r) >= 0:
            r = Secp.reduce(r)
        return r

    @staticmethod
    def neg(a):  # :509-539
        r, borrow = [0] * 4, 0
        for i in range(4):
            d1 = (Secp.P[i] - a[i]) & M64
            d2 = (d1 - borrow) & M64
            b1 = 1 if Secp.P[i] < a[i] else 0
            b2 = 1 if d1 < borrow else 0
            r[i] = d2
            borrow = b1 | b2
        return list(a) if _is_zero(a) else r

    @staticmethod
    def sqr(a):  # :634-713
        product = [0] * 8
        for i in range(4):
            s = a[i] * a[i]
            product[2 * i] = s & M64
            product[2 * i + 1] = s >> 64
        for i in range(4):
            for j in range(i + 1, 4):
                cross = ((a[i] * a[j]) * 2) & M128  # u128::wrapping_mul(2)
                lo, hi = cross & M64, cross >> 64
                s = product[i + j] + lo
                carry = s > M64
                product[i + j] = s & M64
                s = product[i + j + 1] + hi
                carry2 = s > M64
                product[i + j + 1] = s & M64
                if carry or carry2:
                    k = i + j + 2
                    while k < 8:
                        product[k] = (product[k] + 1) & M64
                        if product[k] != 0:
                            break
                        k += 1
        result = product[0:4]
        carry = 0
        for i in range(4, 8):
            m = (product[i] * 0x1000003D1) & M64
            t = (result[0] + m) & M64
            t = (t + carry) & M64
            result[0] = t
            carry = (1 if t < m else 0) | ((1 if t < carry else 0) & (1 if m != 0 else 0))
            for j in range(1, 4):
                t2 = (result[j] + carry) & M64
                result[j] = t2
                carry = 1 if t2 < carry else 0
        return Secp.reduce(result)

    @staticmethod
    def inv(a):  # :599-632
        if _is_zero(a):
            return [0, 0, 0, 0]
        e = [0xFFFFFFFEFFFFFC2D, M64, M64, M64]
        result = [1, 0, 0, 0]
        for i in range(4):
            for j in range(63, -1, -1):
                result = Secp.sqr(result)
                if (e[i] >> j) & 1:
                    result = Secp.mul(result, a)
        return result

    # ---- points: tuples (x, y, z) of limb lists ----
    @staticmethod
    def identity():
        return ([0, 0, 0, 0], [1, 0, 0, 0], [0, 0, 0, 0])

    @staticmethod
    def is_identity(p):  # :1326-1340
        if _is_zero(p[0]) and _is_zero(p[1]) and _is_zero(p[2]):
            return True
        return _is_zero(p[2])

    @staticmethod
    def double(p):  # inherent, :1502-1540
        F = Secp
        if F.is_identity(p):
            return F.identity()
        x, y, z = p
        a = F.sqr(x)
        b = F.sqr(y)
        c = F.sqr(b)
        xpb2 = F.sqr(F.add(x, b))
        t = F.sub(F.sub(xpb2, a), c)
        d = F.add(t, t)
        e = F.mul(a, [3, 0, 0, 0])
        f = F.sqr(e)
        x3 = F.sub(f, F.add(d, d))
        y3 = F.sub(F.mul(e, F.sub(d, x3)), F.mul(c, [8, 0, 0, 0]))
        yz = F.mul(y, z)
        z3 = F.add(yz, yz)
        return (x3, y3, z3)

    @staticmethod
    def double_trait(p):  # :1375-1418
        F = Secp
        if F.is_identity(p):
            return F.identity()
        x, y, z = p
        xx = F.sqr(x)
        yy = F.sqr(y)
        yyyy = F.sqr(yy)
        xy2 = F.sqr(F.add(x, yy))
        w = F.sub(F.sub(xy2, xx), yyyy)
        d = F.add(w, w)
        e = F.mul([3, 0, 0, 0], xx)
        ee = F.sqr(e)
        x3 = F.sub(F.sub(ee, d), d)
        y3 = F.sub(F.mul(e, F.sub(d, x3)), F.mul([8, 0, 0, 0], yyyy))
        z3 = F.add(y, y)
        if z != [1, 0, 0, 0]:
            z3 = F.mul(z3, z)
        return (x3, y3, z3)

    @staticmethod
    def padd(p, q):  # :1444-1498
        F = Secp
        if F.is_identity(p):
            return q
        if F.is_identity(q):
            return p
        x1, y1, z1 = p
        x2, y2, z2 = q
        z1s, z2s = F.sqr(z1), F.sqr(z2)
        u1, u2 = F.mul(x1, z2s), F.mul(x2, z1s)
        z1c, z2c = F.mul(z1s, z1), F.mul(z2s, z2)
        s1, s2 = F.mul(y1, z2c), F.mul(y2, z1c)
        if u1 == u2:
            return F.double(p) if s1 == s2 else F.identity()
        h = F.sub(u2, u1)
        r = F.sub(s2, s1)
        h2 = F.sqr(h)
        h3 = F.mul(h2, h)
        u1h2 = F.mul(u1, h2)
        x3 = F.sub(F.sub(F.sub(F.sqr(r), h3), u1h2), u1h2)
        y3 = F.sub(F.mul(r, F.sub(u1h2, x3)), F.mul(s1, h3))
        z3 = F.mul(F.mul(h, z1), z2)
        return (x3, y3, z3)

    @staticmethod
    def generator():  # :2608-2625 with to_montgomery's R_SQUARED (:225-230)
        r2 = [0x000E9F61, 0x07A20000, 0x00000100, 0]
        gx = [0x59F2815B16F81798, 0x029BFCDB2DCE28D9, 0x55A06295CE870B07, 0x79BE667EF9DCBBAC]
        gy = [0x9C47D08FFB10D4B8, 0xFD17B448A6855419, 0x5DA4FBFC0E1108A8, 0x483ADA7726A3C465]
        return (Secp.mul(gx, r2), Secp.mul(gy, r2), [1, 0, 0, 0])

    @staticmethod
    def to_affine(p):  # :1342-1363
        F = Secp
        if F.is_identity(p):
            return ([0, 0, 0, 0], [0, 0, 0, 0], True)
        zi = F.inv(p[2])
        zi2 = F.sqr(zi)
        zi3 = F.mul(zi2, zi)
        return (F.mul(p[0], zi2), F.mul(p[1], zi3), False)

    @staticmethod
    def multiply(point, k):  # :2635-2692
        F = Secp
        if F.is_identity(point) or _is_zero(k):
            return F.identity()
        by = []
        for i in range(4):  # inherent Scalar::to_bytes, little-endian (:1924-1933)
            for j in range(8):
                by.append((k[i] >> (8 * j)) & 0xFF)
        r0, r1 = F.identity(), point
        for i in range(256):
            bit = (by[i // 8] >> (7 - (i % 8))) & 1
            s = F.padd(r0, r1)
            d0 = F.double(r0)
            d1 = F.double(r1)
            r0 = s if bit else d0
            r1 = d1 if bit else s
        return r0


# ======================================================================================
# P-256 (p256.rs)
# ======================================================================================
class P256c:
    P = [M64, 0x00000000FFFFFFFF, 0, 0xFFFFFFFF00000001]  # :18-19

    @staticmethod
    def cmp(a, b):  # :70-80
        for i in (3, 2, 1, 0):
            if a[i] < b[i]:
                return -1
            if a[i] > b[i]:
                return 1
        return 0

    @staticmethod
    def _sub_limbs(a, b):
        out, borrow = [0] * 4, 0
        for i in range(4):
            d1 = (a[i] - b[i]) & M64
            b1 = 1 if a[i] < b[i] else 0
            d2 = (d1 - borrow) & M64
            b2 = 1 if d1 < borrow else 0
            out[i] = d2
            borrow = b1 + b2
        return out

    @staticmethod
    def _add_limbs(a, b):
        out, c = [0] * 4, 0
        for i in range(4):
            s1 = a[i] + b[i]
            o1 = s1 >> 64
            s1 &= M64
            s2 = s1 + c
            o2 = s2 >> 64
            out[i] = s2 & M64
            c = o1 + o2
        return out, c

    @staticmethod
    def reduce(a):  # :88-99
        a = list(a)
        while P256c.cmp(a, P256c.P) >= 0:
            a = P256c._sub_limbs(a, P256c.P)
        return a

    @staticmethod
    def add(a, b):  # :416-468
        r, carry = P256c._add_limbs(a, b)
        red = [1, 0xFFFFFFFF00000000, M64, 0x00000000FFFFFFFE]
        while carry > 0:
            r, ac = P256c._add_limbs(r, red)
            carry = carry - 1 + ac
        return P256c.reduce(r)

    @staticmethod
    def sub(a, b):  # :470-496
        r = list(a)
        if P256c.cmp(a, b) < 0:
            r = P256c.add(r, P256c.P)
        return P256c._sub_limbs(r, b)

    @staticmethod
    def reduce_wide(w):  # :544-704
        c = []
        for i in range(8):
            c.append(w[i] & 0xFFFFFFFF)
            c.append(w[i] >> 32)
        acc = [c[i] for i in range(8)]  # s1
        for idx, k in ((3, 11), (4, 12), (5, 13), (6, 14), (7, 15)):  # 2*s2
            acc[idx] += 2 * c[k]
        for idx, k in ((3, 12), (4, 13), (5, 14), (6, 15)):  # 2*s3
            acc[idx] += 2 * c[k]
        for idx, k in ((0, 8), (1, 9), (2, 10), (6, 14), (7, 15)):  # s4
            acc[idx] += c[k]
        for idx, k in ((0, 9), (1, 10), (2, 11), (3, 13), (4, 14), (5, 15), (6, 13), (7, 8)):  # s5
            acc[idx] += c[k]
        for idx, k in ((0, 11), (1, 12), (2, 13), (6, 8), (7, 10)):  # s6
            acc[idx] -= c[k]
        for idx, k in ((0, 12), (1, 13), (2, 14), (3, 15), (6, 9), (7, 11)):  # s7
            acc[idx] -= c[k]
        for idx, k in ((0, 13), (1, 14), (2, 15), (3, 8), (4, 9), (5, 10), (7, 12)):  # s8
            acc[idx] -= c[k]
        for idx, k in ((0, 14), (1, 15), (3, 9), (4, 10), (5, 11), (7, 13)):  # s9
            acc[idx] -= c[k]
        for i in range(7):
            carry = acc[i] >> 32  # Python >> on negative ints floors, like i128 >>
            acc[i] &= 0xFFFFFFFF
            acc[i + 1] += carry
        carry = acc[7] >> 32
        acc[7] &= 0xFFFFFFFF
        r = [acc[0] | (acc[1] << 32), acc[2] | (acc[3] << 32), acc[4] | (acc[5] << 32), acc[6] | (acc[7] << 32)]
        while carry > 0:
            r = P256c._sub_limbs(r, P256c.P)
            carry -= 1
        while carry < 0:
            r, _ = P256c._add_limbs(r, P256c.P)
            carry += 1
        return P256c.reduce(r)

    @staticmethod
    def mul(a, b):  # :498-534
        wide = [0] * 8
        for i in range(4):
            carry = 0
            for j in range(4):
                prod = a[i] * b[j] + wide[i + j] + carry
                wide[i + j] = prod & M64
                carry = prod >> 64
            s = wide[i + 4] + carry
            wide[i + 4] = s & M64
            if (s >> 64) != 0:
                for k in range(i + 5, 8):
                    nv = wide[k] + 1
                    wide[k] = nv & M64
                    if nv <= M64:
                        break
        return P256c.reduce_wide(wide)

    @staticmethod
    def sqr(a):
        return P256c.mul(a, a)

    @staticmethod
    def neg(a):  # :707-729
        if _is_zero(a):
            return list(a)
        return P256c._sub_limbs(P256c.P, a)

    @staticmethod
    def pow(a, e):  # :376-393
        result, base = [1, 0, 0, 0], list(a)
        for w in e:
            for _ in range(64):
                if w & 1:
                    result = P256c.mul(result, base)
                base = P256c.sqr(base)
                w >>= 1
        return result

    @staticmethod
    def inv(a):  # :343-370
        if _is_zero(a):
            return [0, 0, 0, 0]
        e, borrow = list(P256c.P), 2
        for i in range(4):
            did = e[i] < borrow
            e[i] = (e[i] - borrow) & M64
            if not did:
                break
            borrow = 1
        return P256c.pow(a, e)

    @staticmethod
    def identity():
        return ([0, 0, 0, 0], [1, 0, 0, 0], [0, 0, 0, 0])

    @staticmethod
    def is_identity(p):
        return _is_zero(p[2])

    @staticmethod
    def double(p):  # :1869-1912
        F = P256c
        if F.is_identity(p):
            return F.identity()
        x, y, z = p
        xx = F.sqr(x)
        yy = F.sqr(y)
        yyyy = F.sqr(yy)
        xy2 = F.sqr(F.add(x, yy))
        w = F.sub(F.sub(xy2, xx), yyyy)
        d = F.add(w, w)
        e = F.mul([3, 0, 0, 0], xx)
        ee = F.sqr(e)
        x3 = F.sub(F.sub(ee, d), d)
        y3 = F.sub(F.mul(e, F.sub(d, x3)), F.mul([8, 0, 0, 0], yyyy))
        z3 = F.add(y, y)
        if z != [1, 0, 0, 0]:
            z3 = F.mul(z3, z)
        return (x3, y3, z3)

    @staticmethod
    def pt_eq(p, q):  # :2034-2068
        F = P256c
        if F.is_identity(p) and F.is_identity(q):
            return True
        if F.is_identity(p) or F.is_identity(q):
            return False
        z1z1, z2z2 = F.sqr(p[2]), F.sqr(q[2])
        u1, u2 = F.mul(p[0], z2z2), F.mul(q[0], z1z1)
        s1 = F.mul(F.mul(p[1], q[2]), z2z2)
        s2 = F.mul(F.mul(q[1], p[2]), z1z1)
        return u1 == u2 and s1 == s2

    @staticmethod
    def padd(p, q):  # :1938-2007
        F = P256c
        if F.is_identity(p):
            return q
        if F.is_identity(q):
            return p
        if F.pt_eq(p, q):
            return F.double(p)
        z1z1, z2z2 = F.sqr(p[2]), F.sqr(q[2])
        u1, u2 = F.mul(p[0], z2z2), F.mul(q[0], z1z1)
        s1 = F.mul(F.mul(p[1], q[2]), z2z2)
        s2 = F.mul(F.mul(q[1], p[2]), z1z1)
        if u1 == u2 and s1 == F.neg(s2):
            return F.identity()
        h = F.sub(u2, u1)
        i = F.sqr(F.add(h, h))
        j = F.mul(h, i)
        r = F.add(F.sub(s2, s1), F.sub(s2, s1))
        v = F.mul(u1, i)
        x3 = F.sub(F.sub(F.sub(F.sqr(r), j), v), v)
        y3 = F.sub(F.mul(r, F.sub(v, x3)), F.mul(F.add(s1, s1), j))
        z3 = F.mul(F.sub(F.sub(F.sqr(F.add(p[2], q[2])), z1z1), z2z2), h)
        return (x3, y3, z3)

    @staticmethod
    def generator():  # :2092-2110
        return ([0xF4A13945D898C296, 0x77037D812DEB33A0, 0xF8BCE6E563A440F2, 0x6B17D1F2E12C4247],
                [0xCBB6406837BF51F5, 0x2BCE33576B315ECE, 0x8EE7EB4A7C0F9E16, 0x4FE342E2FE1A7F9B],
                [1, 0, 0, 0])

    @staticmethod
    def to_affine(p):  # :1835-1857
        F = P256c
        if F.is_identity(p):
            return ([0, 0, 0, 0], [0, 0, 0, 0], True)
        zi = F.inv(p[2])
        zi2 = F.sqr(zi)
        zi3 = F.mul(zi2, zi)
        return (F.mul(p[0], zi2), F.mul(p[1], zi3), False)

    @staticmethod
    def multiply(point, k):  # :2120-2156
        F = P256c
        if F.is_identity(point) or _is_zero(k):
            return F.identity()
        by = [0] * 32
        for i in range(4):  # inherent Scalar::to_bytes, big-endian (:1026-1038)
            for j in range(8):
                by[31 - (i * 8 + j)] = (k[i] >> (8 * j)) & 0xFF
        result = F.identity()
        for i in range(256):
            bit = (by[i // 8] >> (7 - (i % 8))) & 1
            result = F.double(result)
            if bit == 1:
                result = F.padd(result, point)
        return result


# ======================================================================================
# Ed25519 (ed25519.rs)
# ======================================================================================
class Ed:
    P = [0xFFFFFFFFFFFFFFED, M64, M64, 0x7FFFFFFFFFFFFFFF]  # :64-69
    D = [0x75EB4DCA135EDEFF, 0x00E0149A8283B156, 0x198E80F2EEF3D130, 0x2406875CC61A8E3C]  # :86-91

    @staticmethod
    def reduce(a):  # :214-247
        a = list(a)
        top = a[3] >> 63
        a[3] &= 0x7FFFFFFFFFFFFFFF
        carry = top * 19
        for i in range(4):
            s = a[i] + carry
            a[i] = s & M64
            carry = s >> 64
        diff, borrow = [0] * 4, 0
        for i in range(4):
            d1 = (a[i] - Ed.P[i]) & M64
            b1 = 1 if a[i] < Ed.P[i] else 0
            d2 = (d1 - borrow) & M64
            b2 = 1 if d1 < borrow else 0
            diff[i] = d2
            borrow = b1 + b2
        return diff if borrow == 0 else a

    @staticmethod
    def reduce_wide(l):  # :260-289
        low = list(l[0:4])
        carry = 0
        for i in range(4):
            prod = l[4 + i] * 38 + low[i] + carry
            low[i] = prod & M64
            carry = prod >> 64
        fc = ((carry & M64) * 19) & M64  # (carry as u64) * 19
        for i in range(4):
            s = low[i] + fc
            low[i] = s & M64
            fc = s >> 64
        return Ed.reduce(low)

    @staticmethod
    def add(a, b):  # :458-488
        r, carry = [0] * 4, 0
        for i in range(4):
            s = a[i] + b[i] + carry
            r[i] = s & M64
            carry = s >> 64
        if carry > 0:
            ec = carry * 19
            for i in range(4):
                s = r[i] + ec
                r[i] = s & M64
                ec = s >> 64
        return Ed.reduce(r)

    @staticmethod
    def sub(a, b):  # :490-520
        r, borrow = [0] * 4, 0
        for i in range(4):
            d1 = (a[i] - b[i]) & M64
            b1 = 1 if a[i] < b[i] else 0
            d2 = (d1 - borrow) & M64
            b2 = 1 if d1 < borrow else 0
            r[i] = d2
            borrow = b1 + b2
        if borrow > 0:
            carry = 0
            for i in range(4):
                s = r[i] + Ed.P[i] + carry
                r[i] = s & M64
                carry = s >> 64
        return r

    @staticmethod
    def mul(a, b):  # :522-545
        prod = [0] * 8
        for i in range(4):
            carry = 0
            for j in range(4):
                p = a[i] * b[j] + prod[i + j] + carry
                prod[i + j] = p & M64
                carry = p >> 64
            prod[i + 4] = carry
        return Ed.reduce_wide(prod)

    @staticmethod
    def sqr(a):
        return Ed.mul(a, a)

    @staticmethod
    def neg(a):  # :547-570
        if _is_zero(a):
            return [0, 0, 0, 0]
        r, borrow = [0] * 4, 0
        for i in range(4):
            d1 = (Ed.P[i] - a[i]) & M64
            b1 = 1 if Ed.P[i] < a[i] else 0
            d2 = (d1 - borrow) & M64
            b2 = 1 if d1 < borrow else 0
            r[i] = d2
            borrow = b1 + b2
        return r

    @staticmethod
    def pow(a, e):  # :410-431
        result, base = [1, 0, 0, 0], list(a)
        for w in e:
            for i in range(64):
                nr = Ed.mul(result, base)
                if (w >> i) & 1:
                    result = nr
                base = Ed.sqr(base)
        return result

    @staticmethod
    def inv(a):  # :603-621
        return Ed.pow(a, [0xFFFFFFFFFFFFFFEB, M64, M64, 0x7FFFFFFFFFFFFFFF])

    @staticmethod
    def identity():
        return ([0, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0], [0, 0, 0, 0])

    @staticmethod
    def is_identity(p):  # :1785-1791
        return _is_zero(p[0]) and p[1] == p[2] and _is_zero(p[3])

    @staticmethod
    def padd(p, q):  # :1864-1928
        F = Ed
        if F.is_identity(p):
            return q
        if F.is_identity(q):
            return p
        x1, y1, z1, t1 = p
        x2, y2, z2, t2 = q
        if x1 == F.neg(x2) and y1 == y2:
            return F.identity()
        a = F.mul(F.sub(y1, x1), F.sub(y2, x2))
        b = F.mul(F.add(y1, x1), F.add(y2, x2))
        c = F.mul(F.mul(t1, t2), F.D)
        d = F.mul(z1, z2)
        e = F.sub(b, a)
        f = F.sub(d, c)
        g = F.add(d, c)
        h = F.add(b, a)
        return (F.mul(e, f), F.mul(g, h), F.mul(f, g), F.mul(e, h))

    @staticmethod
    def double(p):  # :1828-1832
        return Ed.padd(p, p)

    @staticmethod
    def generator():  # :2015-2052
        y = [0x2DFC9311D90045F9, 0x0A71C760BF38C6A7, 0xA6FB8EEBCEAA2C8D, 0x5FD9C9E6CC3CCCCC]
        x = [0x1A1462FAFB9683F2, 0xD2E8A68B8B30C404, 0xA0C0F3A1E9E71B63, 0x216936D3CD6E53FE]
        return (x, y, [1, 0, 0, 0], Ed.mul(x, y))

    @staticmethod
    def to_affine(p):  # :1793-1811
        if Ed.is_identity(p):
            return ([0, 0, 0, 0], [0, 0, 0, 0], True)
        zi = Ed.inv(p[2])
        return (Ed.mul(p[0], zi), Ed.mul(p[1], zi), False)

    @staticmethod
    def multiply(point, k):  # :2062-2097
        F = Ed
        if F.is_identity(point) or _is_zero(k):
            return F.identity()
        result, addend = F.identity(), point
        for i in range(4):
            for j in range(64):
                bit = (k[i] >> j) & 1
                rpa = F.padd(result, addend)
                if bit:
                    result = rpa
                addend = F.double(addend)
        return result


# ======================================================================================
# secp256k1 scalar field as implemented (secp256k1.rs) and ECDSA verify (ecdsa.rs:213-281)
# ======================================================================================
class SecpScalar:
    N = [0xBFD25E8CD0364141, 0xBAAEDCE6AF48A03B, M64, 0xFFFFFFFFFFFFFFFE]  # :27-28

    @staticmethod
    def ge_n(a):  # the comparison spelled out at :1955-1958
        N = SecpScalar.N
        return (a[3] > N[3] or (a[3] == N[3] and a[2] > N[2]) or (a[3] == N[3] and a[2] == N[2] and a[1] > N[1])
                or (a[3] == N[3] and a[2] == N[2] and a[1] == N[1] and a[0] >= N[0]))

    @staticmethod
    def reduce(a):  # :1953-1969
        a = list(a)
        if SecpScalar.ge_n(a):
            borrow = 0
            for i in range(4):
                d1 = (a[i] - SecpScalar.N[i]) & M64
                b1 = a[i] < SecpScalar.N[i]
                d2 = (d1 - borrow) & M64
                b2 = d1 < borrow
                a[i] = d2
                borrow = 1 if (b1 or b2) else 0
        return a

    @staticmethod
    def add(a, b):  # :2358-2378 -- the carry out of the top limb is dropped, then one reduce()
        v = (sum(x << (64 * i) for i, x in enumerate(a)) + sum(x << (64 * i) for i, x in enumerate(b))) & ((1 << 256) - 1)
        return SecpScalar.reduce([(v >> (64 * i)) & M64 for i in range(4)])

    @staticmethod
    def mul(a, b):  # :2410-2456 -- keeps only t[0..4] of the 512-bit product
        t = [0] * 8
        for i in range(4):
            carry = 0
            for j in range(4):
                product = a[i] * b[j]
                lo, hi = product & M64, product >> 64
                res1 = t[i + j] + lo
                c1 = res1 >> 64
                res1 &= M64
                res2 = res1 + carry
                c2 = res2 >> 64
                t[i + j] = res2 & M64
                carry = hi + c1 + c2
                assert carry <= M64
                if j == 3:
                    t[i + j + 1] = carry
        r = SecpScalar.reduce(t[0:4])
        while SecpScalar.ge_n(r):
            r = SecpScalar.reduce(r)
        return r

    @staticmethod
    def inv(a):  # :2162-2195
        if _is_zero(a):
            return None
        e = [0xBFD25E8CD036413F, 0xBAAEDCE6AF48A03B, M64, 0xFFFFFFFFFFFFFFFE]
        result = [1, 0, 0, 0]
        for i in range(4):
            for j in range(63, -1, -1):
                result = SecpScalar.mul(result, result)
                if (e[i] >> j) & 1:
                    result = SecpScalar.mul(result, a)
        return result

    @staticmethod
    def from_bytes_be(b):  # trait Scalar::from_bytes :2270-2297
        l = [0] * 4
        for i in range(4):
            for j in range(8):
                l[i] |= b[31 - (i * 8 + j)] << (8 * j)
        return l, not SecpScalar.ge_n(l)


def secp256k1_ecdsa_verify(digest, r, s, pk_xy, pk_inf=False):
    """ecdsa.rs:213-281 with the digest supplied.  1 valid, 0 invalid, 2 = the reference panics."""
    S, F = SecpScalar, Secp
    if _is_zero(r) or _is_zero(s):
        return 0
    if S.ge_n(r) or S.ge_n(s):
        return 0
    h, ok = S.from_bytes_be(list(digest))
    if not ok:
        return 2
    s_inv = S.inv(s)
    if s_inv is None:
        return 0
    u1 = S.mul(h, s_inv)
    u2 = S.mul(r, s_inv)
    q = F.identity() if pk_inf else (list(pk_xy[0:4]), list(pk_xy[4:8]), [1, 0, 0, 0])
    rp = F.padd(F.multiply(F.generator(), u1), F.multiply(q, u2))
    if F.is_identity(rp):
        return 0
    x, _, _ = F.to_affine(rp)
    xr = F.mul(x, [1, 0, 0, 0])  # FieldElement::to_bytes = mont_reduce (:138-178), then Scalar::from_bytes
    if S.ge_n(xr):
        return 2
    return 1 if xr == list(r) else 0


class P256Scalar:
    """p256.rs Scalar (875-1038, 1409-1432).  Whole-integer arithmetic where the reference's u128 limb
    loops are exact (no u128 column overflows: the constant's limbs are 60, 63, 0 and 32 bits wide), limb
    loops where they are not an integer identity."""
    N = 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551  # :23-24
    C = [0x0C46353D039CDAAF, 0x4319055258E8617B, 0, 0x00000000FFFFFFFF]  # TWO_256_MINUS_N :932-937
    M256 = (1 << 256) - 1

    @staticmethod
    def val(l):
        return l[0] | (l[1] << 64) | (l[2] << 128) | (l[3] << 192)

    @staticmethod
    def limbs(v):
        return [(v >> (64 * i)) & M64 for i in range(4)]

    @staticmethod
    def reduce_wide(w):  # :924-1020; w an integer below 2^512
        S = P256Scalar
        cv = S.val(S.C)
        first = (w & S.M256) + (w >> 256) * cv  # product + low, carries propagated: exact, below 2^481
        low2, high2 = first & S.M256, first >> 256
        if high2 != 0:
            # 993-998: only product2[0..4] is added -- the high half of high2 * C is dropped
            t = low2 + ((high2 * cv) & S.M256)
            low2, carry = t & S.M256, t >> 256
            if carry > 0:  # 1000-1007: limb i receives c * C[i] with c the running carry, not C itself
                l = S.limbs(low2)
                c = carry
                for i in range(4):
                    sm = l[i] + c * S.C[i]
                    assert sm <= M128
                    l[i] = sm & M64
                    c = sm >> 64
                low2 = S.val(l)
        while low2 >= S.N:  # 1010-1019 (a borrow chain that is exact for low2 >= n)
            low2 -= S.N
        return low2

    @staticmethod
    def mul(a, b):  # :1409-1432 -- the schoolbook product is exact
        S = P256Scalar
        return S.limbs(S.reduce_wide(S.val(a) * S.val(b)))

    @staticmethod
    def add(a, b):  # :1352-1375 -- on a carry out, reduce() still sees only the low 256 bits
        S = P256Scalar
        t = S.val(a) + S.val(b)
        low = t & S.M256
        if (t >> 256) > 0 or low >= S.N:
            while low >= S.N:
                low -= S.N
        return S.limbs(low)

    @staticmethod
    def inv(a):  # :1057-1080 with pow :1083-1100 (LSB first; square() = s * s)
        S = P256Scalar
        if _is_zero(a):
            return None
        e = S.N - 2
        assert S.limbs(e) == [0xF3B9CAC2FC63254F, 0xBCE6FAADA7179E84, M64, 0xFFFFFFFF00000000]
        result, base = [1, 0, 0, 0], list(a)
        for k in range(256):
            if (e >> k) & 1:
                result = S.mul(result, base)
            base = S.mul(base, base)
        return result

    @staticmethod
    def from_bytes_be(b):  # :1041-1055
        v = int.from_bytes(bytes(b), "big")
        return P256Scalar.limbs(v), v < P256Scalar.N

    @staticmethod
    def ct_lt_default(a, b):  # forge-ec-core/src/lib.rs:497-531 (P-256 keeps the trait default)
        ab = P256Scalar.val(a).to_bytes(32, "big")
        bb = P256Scalar.val(b).to_bytes(32, "big")
        result, eq_so_far = False, True
        for i in range(32):
            borrow1 = bb[i] < ab[i]  # other_byte.overflowing_sub(self_byte)
            result = result or (eq_so_far and not borrow1)
            eq_so_far = eq_so_far and ab[i] == bb[i]
        return result


def p256_ecdsa_verify(digest, r, s, pk_xy, pk_inf=False):
    """ecdsa.rs:213-281 for C = P256 with the digest supplied.  1 valid, 0 invalid, 2 = the reference
    panics (unwrap of a None CtOption at :239 or :271)."""
    S, F = P256Scalar, P256c
    if _is_zero(r) or _is_zero(s):
        return 0
    order = S.limbs(S.N)
    if not (S.ct_lt_default(r, order) and S.ct_lt_default(s, order)):
        return 0
    h, ok = S.from_bytes_be(list(digest))
    if not ok:
        return 2
    s_inv = S.inv(s)
    if s_inv is None:
        return 0
    u1 = S.mul(h, s_inv)
    u2 = S.mul(r, s_inv)
    q = F.identity() if pk_inf else (list(pk_xy[0:4]), list(pk_xy[4:8]), [1, 0, 0, 0])  # from_affine :1859-1867
    rp = F.padd(F.multiply(F.generator(), u1), F.multiply(q, u2))
    if F.is_identity(rp):
        return 0
    x, _, _ = F.to_affine(rp)
    # field_to_bytes = FieldElement::to_bytes (:288-300): raw limbs, big-endian; Scalar::from_bytes of them
    if S.val(x) >= S.N:
        return 2
    return 1 if list(x) == list(r) else 0


def ecdsa_batch_verify(curve, digests, r, s, pk_xy, pk_inf, a):
    """Ecdsa::<C, D>::batch_verify (ecdsa.rs:287-391) for C = Secp256k1 (curve 0) / P256 (curve 1) with the
    digests and the weights a_i supplied.  Returns (status, r_sum, r_scalar_sum): status 1 true, 0 false,
    2 = the reference panics; the sums are None when the loop returned early."""
    n = len(digests)
    if n == 0:
        return 0, None, None  # 289-291
    S, F = (SecpScalar, Secp) if curve == SECP256K1 else (P256Scalar, P256c)
    order = SecpScalar.N if curve == SECP256K1 else P256Scalar.limbs(P256Scalar.N)
    r_sum = F.identity()
    for i in range(n):
        ri, si, ai = list(r[i]), list(s[i]), list(a[i])
        if _is_zero(ri) or _is_zero(si):
            return 0, None, None
        if curve == SECP256K1:
            if S.ge_n(ri) or S.ge_n(si):
                return 0, None, None
        elif not (S.ct_lt_default(ri, order) and S.ct_lt_default(si, order)):
            return 0, None, None
        h, ok = S.from_bytes_be(list(digests[i]))
        if not ok:
            return 2, None, None  # 334
        s_inv = S.inv(si)
        if s_inv is None:
            return 0, None, None
        au1 = S.mul(ai, S.mul(h, s_inv))  # 345-350
        au2 = S.mul(ai, S.mul(ri, s_inv))
        inf = bool(pk_inf[i]) if pk_inf is not None else False
        q = F.identity() if inf else (list(pk_xy[i][0:4]), list(pk_xy[i][4:8]), [1, 0, 0, 0])
        r_i = F.padd(F.multiply(F.generator(), au1), F.multiply(q, au2))  # 353-355
        r_sum = F.padd(r_sum, r_i)  # 358
    total = [0, 0, 0, 0]
    for i in range(n):  # 368-372
        total = S.add(total, S.mul(list(a[i]), list(r[i])))
    if F.is_identity(r_sum):
        return 0, r_sum, total  # 361-364
    x, _, _ = F.to_affine(r_sum)
    if curve == SECP256K1:
        xs = F.mul(x, [1, 0, 0, 0])  # FieldElement::to_bytes = mont_reduce
        if S.ge_n(xs):
            return 2, r_sum, total
    else:
        xs = list(x)
        if S.val(xs) >= S.N:
            return 2, r_sum, total
    return (1 if xs == total else 0), r_sum, total


def ed25519_eddsa_verify(r_xy, r_inf, pk_xy, pk_inf, s, k):
    """Eddsa::<Ed25519, D>::verify (eddsa.rs:174-211) / Ed25519::verify (430-447) from the point computation
    on, s and k = from_bytes_reduced(hash) given.  1 true, 0 false, 2 = the reference panics
    (to_affine unwraps the inverse of a zero z, ed25519.rs:1805)."""
    F = Ed

    def from_affine(xy, inf):  # ed25519.rs:1813-1826
        if inf:
            return F.identity()
        x, y = list(xy[0:4]), list(xy[4:8])
        return (x, y, [1, 0, 0, 0], F.mul(x, y))

    def affine_or_panic(p):  # 1793-1811
        if F.is_identity(p):
            return ([0] * 8, True)
        if _is_zero(p[2]):
            raise ZeroDivisionError
        x, y, _ = F.to_affine(p)
        return (list(x) + list(y), False)

    if r_inf:
        return 0  # eddsa.rs:174-177
    s_g = F.multiply(F.generator(), list(s))
    k_a = F.multiply(from_affine(pk_xy, pk_inf), list(k))
    rk = F.padd(from_affine(r_xy, False), k_a)
    try:
        a1, i1 = affine_or_panic(s_g)
        a2, i2 = affine_or_panic(rk)
    except ZeroDivisionError:
        return 2
    p2 = from_affine(a2, i2)
    neg = (F.neg(p2[0]), p2[1], p2[2], F.neg(p2[3]))  # negate 1834-1841; Sub = self + rhs.negate() 1936-1947
    return 1 if F.is_identity(F.padd(from_affine(a1, i1), neg)) else 0


def secp256k1_schnorr_batch_verify(pk_xy, pk_inf, r_xy, r_inf, s, a, e):
    """schnorr.rs:194-290 with challenges e and weights a supplied.  -> (result, sides, sides_inf):
    sides = x, y of to_affine(s_g) then x, y of to_affine(r_e_p) (what line 286 compares)."""
    S, F = SecpScalar, Secp
    n = len(s)
    zero = [0, 0, 0, 0]
    if n == 0:
        return 0, [zero] * 4, [0, 0]
    for i in range(n):  # 204-225: the is_on_curve tests are `!u8 == 1`, never true; identity tests reject
        if (pk_inf is not None and pk_inf[i]) or (r_inf is not None and r_inf[i]):
            return 0, [zero] * 4, [0, 0]
    g = F.generator()
    s_g, r_e_p = F.identity(), F.identity()
    one = [1, 0, 0, 0]
    for i in range(n):
        s_g = F.padd(s_g, F.multiply(g, S.mul(list(s[i]), list(a[i]))))
        ep = F.multiply((list(pk_xy[i][0:4]), list(pk_xy[i][4:8]), one), list(e[i]))
        rp = F.padd((list(r_xy[i][0:4]), list(r_xy[i][4:8]), one), ep)
        r_e_p = F.padd(r_e_p, F.multiply(rp, list(a[i])))
    x1, y1, i1 = F.to_affine(s_g)
    x2, y2, i2 = F.to_affine(r_e_p)
    ok = (x1 == x2 and y1 == y2) or (i1 and i2)
    return (1 if ok else 0), [x1, y1, x2, y2], [int(i1), int(i2)]


def p256_schnorr_batch_verify(pk_xy, pk_inf, r_xy, r_inf, s, a, e):
    """schnorr.rs:194-290 for C = P256 (Scalar Mul p256.rs:1409-1432): as secp256k1_schnorr_batch_verify."""
    S, F = P256Scalar, P256c
    n = len(s)
    zero = [0, 0, 0, 0]
    if n == 0:
        return 0, [zero] * 4, [0, 0]
    for i in range(n):
        if (pk_inf is not None and pk_inf[i]) or (r_inf is not None and r_inf[i]):
            return 0, [zero] * 4, [0, 0]
    g = F.generator()
    s_g, r_e_p = F.identity(), F.identity()
    one = [1, 0, 0, 0]
    for i in range(n):
        s_g = F.padd(s_g, F.multiply(g, S.mul(list(s[i]), list(a[i]))))
        ep = F.multiply((list(pk_xy[i][0:4]), list(pk_xy[i][4:8]), one), list(e[i]))
        rp = F.padd((list(r_xy[i][0:4]), list(r_xy[i][4:8]), one), ep)
        r_e_p = F.padd(r_e_p, F.multiply(rp, list(a[i])))
    x1, y1, i1 = F.to_affine(s_g)
    x2, y2, i2 = F.to_affine(r_e_p)
    ok = (x1 == x2 and y1 == y2) or (i1 and i2)
    return (1 if ok else 0), [x1, y1, x2, y2], [int(i1), int(i2)]


class Ed25519Scalar:
    """ed25519.rs Scalar: Add (1193-1239) and Mul (1256-1376) as the RELEASE profile runs them (the reference's
    Cargo.toml:53-58 sets no overflow-checks: u128 `+=` wraps modulo 2^128 where a debug build panics)."""
    ORDER = [0x5812631A5CF5D3ED, 0x14DEF9DEA2F79CD6, 0, 0x1000000000000000]

    @staticmethod
    def ge_order(r):  # the loops at 1215-1226 / 1303-1312 / 1354-1363
        for i in (3, 2, 1, 0):
            if r[i] < Ed25519Scalar.ORDER[i]:
                return False
            if r[i] > Ed25519Scalar.ORDER[i]:
                return True
        return True

    @staticmethod
    def sub_order(r):  # 1229-1236: limb-wise with a borrow, `diff as u64`
        out, borrow = [], 0
        for i in range(4):
            diff = r[i] - Ed25519Scalar.ORDER[i] - borrow
            out.append(diff & M64)
            borrow = 1 if diff < 0 else 0
        return out

    @staticmethod
    def add(a, b):
        r, carry = [], 0
        for i in range(4):
            total = a[i] + b[i] + carry
            r.append(total & M64)
            carry = total >> 64
        return Ed25519Scalar.sub_order(r) if Ed25519Scalar.ge_order(r) else r   # (the last carry is dropped)

    @staticmethod
    def mul_release(a, b):
        """-> (product limbs, overflowed): overflowed = some u128 sum of 1268-1272 / 1278 passed 2^128."""
        m128 = (1 << 128) - 1
        product, ovf = [0] * 8, False
        for i in range(4):
            for j in range(4):
                total = product[i + j] + a[i] * b[j]
                ovf = ovf or total > m128
                product[i + j] = total & m128
        carry = 0
        for i in range(8):
            total = product[i] + carry
            ovf = ovf or total > m128
            total &= m128
            carry = total >> 64
            product[i] = total & M64
        result, high = product[0:4], product[4:8]
        if any(high) or Ed25519Scalar.ge_order(result):
            if any(high):
                for _ in range(256):                      # 1347-1349: result += high_bits
                    result = Ed25519Scalar.add(result, high)
            if Ed25519Scalar.ge_order(result):
                result = Ed25519Scalar.sub_order(result)
        return result, ovf


def ed25519_schnorr_batch_verify(pk_xy, pk_inf, r_xy, r_inf, s, a, e):
    """schnorr.rs:194-290 for C = Ed25519 under the release profile.  -> (result, sides, sides_inf, debug_build_panics):
    result 1 true / 0 false / 2 = the reference panics in to_affine (ed25519.rs:1805); debug_build_panics = some
    s_i * a_i wrapped a u128 sum."""
    F = Ed
    n = len(s)
    zero = [0, 0, 0, 0]
    if n == 0:
        return 0, [zero] * 4, [0, 0], 0
    for i in range(n):
        if (pk_inf is not None and pk_inf[i]) or (r_inf is not None and r_inf[i]):
            return 0, [zero] * 4, [0, 0], 0

    def from_affine(xy):  # 1813-1826 (the identities were rejected above)
        x, y = list(xy[0:4]), list(xy[4:8])
        return (x, y, [1, 0, 0, 0], F.mul(x, y))

    g = F.generator()
    s_g, r_e_p = F.identity(), F.identity()
    panics = 0
    for i in range(n):
        sa, ovf = Ed25519Scalar.mul_release(list(s[i]), list(a[i]))
        panics |= int(ovf)
        s_g = F.padd(s_g, F.multiply(g, sa))
        ep = F.multiply(from_affine(pk_xy[i]), list(e[i]))
        rp = F.padd(from_affine(r_xy[i]), ep)
        r_e_p = F.padd(r_e_p, F.multiply(rp, list(a[i])))
    for p in (s_g, r_e_p):
        if not F.is_identity(p) and _is_zero(p[2]):
            return 2, [zero] * 4, [0, 0], panics
    x1, y1, i1 = F.to_affine(s_g)
    x2, y2, i2 = F.to_affine(r_e_p)
    ok = (x1 == x2 and y1 == y2) or (i1 and i2)
    return (1 if ok else 0), [x1, y1, x2, y2], [int(i1), int(i2)], panics


def schnorr_verify(curve, pk_xy, pk_inf, r_xy, r_inf, s, e):
    """Schnorr::<C, D>::verify (forge-ec-signature/src/schnorr.rs:90-140) from line 125 on, the challenge
    e = from_bytes_reduced(H(R || P || m)) given.  1 true, 0 false, 2 = the reference panics (Ed25519's to_affine
    unwrapping the inverse of a zero z, ed25519.rs:1805)."""
    if r_inf:
        return 0                                              # 103-105
    one = [1, 0, 0, 0]
    x_p, y_p = list(pk_xy[0:4]), list(pk_xy[4:8])
    if curve in (SECP256K1, P256):
        F = Secp if curve == SECP256K1 else P256c
        s_g = F.multiply(F.generator(), list(s))              # 125
        pk = F.identity() if pk_inf else (x_p, y_p, one)      # from_affine
        e_p = F.multiply(pk, list(e))                         # 126
        x, y, _ = F.to_affine(e_p)                            # 129: (0, 0) for the identity
        ny = F.neg(y)                                         # 130
        on = _secp_on_curve(x, ny) if curve == SECP256K1 else (P256c.sqr(ny) == _p256_rhs(x))
        if not on:
            return 0                                          # 132-134: PointAffine::new is None
        rx, ry, ri = F.to_affine(F.padd(s_g, (x, ny, one)))   # 136-139
    else:
        F = Ed
        s_g = F.multiply(F.generator(), list(s))
        pk = F.identity() if pk_inf else (x_p, y_p, one, F.mul(x_p, y_p))
        e_p = F.multiply(pk, list(e))
        if not F.is_identity(e_p) and _is_zero(e_p[2]):
            return 2
        x, y, _ = F.to_affine(e_p)
        ny = F.neg(y)
        x2, y2 = F.sqr(x), F.sqr(ny)
        if F.add(F.neg(x2), y2) != F.add(one, F.mul(F.D, F.mul(x2, y2))):   # PointAffine::new 1477-1498
            return 0
        rp = F.padd(s_g, (x, ny, one, F.mul(x, ny)))
        if not F.is_identity(rp) and _is_zero(rp[2]):
            return 2
        rx, ry, ri = F.to_affine(rp)
    same = list(rx) == list(r_xy[0:4]) and list(ry) == list(r_xy[4:8])
    return 1 if (same or (ri and r_inf)) else 0                # 142: AffinePoint::ct_eq


def compress(curve, x, y, inf=False):
    """PointAffine::to_bytes (secp256k1.rs:875-896, p256.rs:1558-1578, ed25519.rs:1505-1525)."""
    if inf:
        return bytes(33)

    def fb(a):
        if curve == SECP256K1:
            a = Secp.mul(a, [1, 0, 0, 0])  # FieldElement::to_bytes = mont_reduce, big-endian (138-178)
        elif curve == ED25519:
            a = Ed.reduce(a)               # reduce(), little-endian (295-310)
        v = sum(int(a[i]) << (64 * i) for i in range(4))
        return v.to_bytes(32, "little" if curve == ED25519 else "big")

    yb = fb(list(y))
    return bytes([0x03 if yb[31] & 1 else 0x02]) + fb(list(x))


def to_bytes_field(curve, a):
    """FieldElement::to_bytes: secp256k1 mont_reduce + big-endian (138-178); P-256 raw limbs big-endian
    (288-300); Ed25519 reduce() + little-endian (295-310)."""
    a = list(a)
    if curve == SECP256K1:
        a = Secp.mul(a, [1, 0, 0, 0])
    elif curve == ED25519:
        a = Ed.reduce(a)
    v = sum(int(a[i]) << (64 * i) for i in range(4))
    return v.to_bytes(32, "little" if curve == ED25519 else "big")


def field_from_bytes(curve, b):
    """FieldElement::from_bytes(&[u8; 32]) -> (limbs, valid): secp256k1.rs:182-212, p256.rs:303-317,
    ed25519.rs:315-357."""
    if curve == SECP256K1:
        v = int.from_bytes(b, "big")
        l = [(v >> (64 * i)) & M64 for i in range(4)]
        valid = Secp.cmp_p(l) < 0
        mont = Secp.mul(l, [0x000E9F61, 0x07A20000, 0x00000100, 0])  # to_montgomery (219-235)
        return (mont if valid else [0, 0, 0, 0]), valid
    if curve == P256:
        v = int.from_bytes(b, "big")
        l = [(v >> (64 * i)) & M64 for i in range(4)]
        return l, P256c.cmp(l, P256c.P) < 0
    l = [int.from_bytes(b[8 * i:8 * i + 8], "little") for i in range(4)]
    is_less, is_equal = False, True
    for i in (3, 2, 1, 0):  # 335-349: the `gt` early return fires whatever the higher limbs decided
        lt, eq, gt = l[i] < Ed.P[i], l[i] == Ed.P[i], l[i] > Ed.P[i]
        is_less = is_less or (is_equal and lt)
        is_equal = is_equal and eq
        if gt:
            return [0, 0, 0, 0], False
    return l, is_less


def _secp_pow(a, e):  # trait pow, secp256k1.rs:715-735 (LSB first, result *= base, base = base.square())
    result, base = [1, 0, 0, 0], list(a)
    for w in e:
        for j in range(64):
            if (w >> j) & 1:
                result = Secp.mul(result, base)
            base = Secp.sqr(base)
    return result


def _secp_on_curve(x, y):  # PointAffine::new 856-869 / is_on_curve 978-1004
    seven_m = Secp.mul([7, 0, 0, 0], [0x000E9F61, 0x07A20000, 0x00000100, 0])
    return Secp.sqr(y) == Secp.add(Secp.mul(Secp.sqr(x), x), seven_m)


P256_B = [0x3BCE3C3E27D2604B, 0x651D06B0CC53B0F6, 0xB3EBBD55769886BC, 0x5AC635D8AA3A93E7]  # p256.rs:28-33


def _p256_rhs(x):  # x^3 - 3x + b as spelled at 1536-1543 and 1612-1618
    x3 = P256c.mul(P256c.sqr(x), x)
    return P256c.add(P256c.sub(x3, P256c.mul([3, 0, 0, 0], x)), P256_B)


ED_SQRT_M1 = [0xC4EE1B274A0EA0B0, 0x2F431806AD2FE478, 0x2B4D00993DFBD7A7, 0x2B8324804FC1DF0B]  # ed25519.rs:132-137


def _ed_sqrt(a):  # ed25519.rs:359-402
    leg = Ed.pow(a, [0x7FFFFFFFFFFFFFF6, M64, M64, 0x3FFFFFFFFFFFFFFF])
    if leg != [1, 0, 0, 0] and leg != [0, 0, 0, 0]:
        return None
    cand = Ed.pow(a, [0x1FFFFFFFFFFFFFFF, M64, M64, 0x0FFFFFFFFFFFFFFF])
    ok1 = Ed.sqr(cand) == list(a)
    alt = Ed.mul(cand, ED_SQRT_M1)
    ok2 = Ed.sqr(alt) == list(a)
    return (alt if ok2 else cand) if (ok1 or ok2) else None


def decompress(curve, b):
    """PointAffine::from_bytes(&[u8; 33]) -> None, or (x, y, infinity): secp256k1.rs:896-976,
    p256.rs:1580-1639, ed25519.rs:1526-1582."""
    if b[0] == 0x00:
        return [0, 0, 0, 0], [0, 0, 0, 0], True
    if b[0] not in (2, 3):
        return None
    x, valid = field_from_bytes(curve, b[1:33])
    if not valid:
        return None
    want_odd = b[0] == 3
    if curve == SECP256K1:
        y2 = Secp.add(Secp.mul(Secp.sqr(x), x), [7, 0, 0, 0])     # raw seven (935)
        s = _secp_pow(y2, [0xFF0C, 0xFFFF, 0xFFFE, 0x3FFF])      # inherent sqrt (112-131)
        if Secp.sqr(s) != y2:
            return None
        parity = to_bytes_field(curve, s)[31] & 1
        y = Secp.neg(s) if (want_odd != bool(parity)) else s
        return (x, y, False) if _secp_on_curve(x, y) else None
    if curve == P256:
        y2 = _p256_rhs(x)
        s = P256c.pow(y2, [0xC0000000, 0x40000000, 0x4000000000000000, 0x40000000C0000000])  # 320-339
        if P256c.sqr(s) != y2:
            return None
        parity = to_bytes_field(curve, s)[31] & 1
        return x, (P256c.neg(s) if (bool(parity) != want_odd) else s), False
    x2 = Ed.sqr(x)
    y2 = Ed.add(Ed.add(Ed.mul(x2, x), Ed.mul([0x7FFFFFDA, 0, 0, 0], x2)), x)
    s = _ed_sqrt(y2)
    if s is None:
        return None
    parity = to_bytes_field(curve, s)[31] & 1
    return x, (Ed.neg(s) if (bool(parity) != want_odd) else s), False


def encode_uncompressed(curve, x, y, inf=False):
    """UncompressedPoint::from_affine (forge-ec-encoding/src/point.rs:186-211)."""
    if inf:
        return bytes(65)
    return bytes([4]) + to_bytes_field(curve, x) + to_bytes_field(curve, y)


def decode_uncompressed(curve, b):
    """UncompressedPoint::to_affine (point.rs:214-281) -> None, or (x, y, infinity)."""
    if b[0] == 0x00:
        return [0, 0, 0, 0], [0, 0, 0, 0], True
    if b[0] != 0x04:
        return None
    x, vx = field_from_bytes(curve, b[1:33])
    y, vy = field_from_bytes(curve, b[33:65])
    if not (vx and vy):
        return None
    F = CURVES[curve]
    a = {SECP256K1: [0, 0, 0, 0], P256: [0xFFFFFFFC, 0xFFFFFFFF, 0xFFFFFFFE, 0xFFFFFFFF],
         ED25519: [0x7FFFFFFFFFFFFFED, 0x7FFFFFFFFFFFF, 0, 0]}[curve]          # get_a() as written
    bb = {SECP256K1: [7, 0, 0, 0], P256: P256_B, ED25519: [0, 0, 0, 0]}[curve]  # get_b()
    x3 = F.mul(F.mul(x, x), x)                                                 # `x * x`, then `* x` (251-252)
    if F.mul(y, y) != F.add(F.add(x3, F.mul(a, x)), bb):
        return None
    if curve == SECP256K1:
        on = _secp_on_curve(x, y)
    elif curve == P256:
        on = P256c.sqr(y) == _p256_rhs(x)                                      # PointAffine::new 1535-1552
    else:                                                                      # ed25519.rs:1476-1498
        x2, y2 = Ed.sqr(x), Ed.sqr(y)
        on = Ed.add(Ed.neg(x2), y2) == Ed.add([1, 0, 0, 0], Ed.mul(Ed.D, Ed.mul(x2, y2)))
    return (x, y, False) if on else None


CURVES = {SECP256K1: Secp, P256: P256c, ED25519: Ed}


def flat(pt):
    out = []
    for c in pt:
        out.extend(c)
    return out


def unflat(limbs):
    return tuple(list(limbs[i:i + 4]) for i in range(0, len(limbs), 4))


def double_mul(curve, u1, u2, q):
    """R = multiply(G,u1) + multiply(Q,u2)  (forge-ec-signature/src/ecdsa.rs:254-256)."""
    F = CURVES[curve]
    return F.padd(F.multiply(F.generator(), u1), F.multiply(q, u2))


def ecdh(curve, sk, pk_xy, pk_inf=False):
    """KeyExchange::derive_shared_secret (secp256k1.rs:1884-1904; p256.rs:2281-2302 with validate_public_key
    2304-2312 -> validate_point 2187-2191 -> is_on_curve 1636-1656).  Returns (status, 32 bytes): 0 = Ok(bytes),
    1 = Err(InvalidPublicKey) (P-256 only), 2 = Err because the product is the identity."""
    x, y = list(pk_xy[0:4]), list(pk_xy[4:8])
    if curve == SECP256K1:
        F = Secp
    elif curve == P256:
        F = P256c
        if pk_inf or F.sqr(y) != _p256_rhs(x):
            return 1, bytes(32)
    else:
        raise ValueError("Ed25519 does not implement KeyExchange")
    q = F.identity() if pk_inf else (x, y, [1, 0, 0, 0])
    ax, _, inf = F.to_affine(F.multiply(q, list(sk)))
    if inf:
        return 2, bytes(32)
    return 0, bytes(to_bytes_field(curve, ax))


def validate_point(curve, xy, inf=False):
    """Curve::validate_point: PointAffine::is_on_curve for Secp256k1 (secp256k1.rs:2722-2726 -> 978-1004) and P256
    (p256.rs:2187-2191 -> 1636-1656); the trait default for Ed25519 (forge-ec-core/src/lib.rs:905-925): on the curve
    and multiply(multiply(from_affine(p), 8), L) is the identity (default clear_cofactor 885-897)."""
    x, y = list(xy[0:4]), list(xy[4:8])
    if curve == SECP256K1:
        return 1 if inf or _secp_on_curve(x, y) else 0
    if curve == P256:
        return 1 if inf or P256c.sqr(y) == _p256_rhs(x) else 0
    x2, y2 = Ed.sqr(x), Ed.sqr(y)
    on = inf or Ed.add(Ed.neg(x2), y2) == Ed.add([1, 0, 0, 0], Ed.mul(Ed.D, Ed.mul(x2, y2)))   # 1719-1744
    p = Ed.identity() if inf else (x, y, [1, 0, 0, 0], Ed.mul(x, y))
    order = [0x5812631A5CF5D3ED, 0x14DEF9DEA2F79CD6, 0, 0x1000000000000000]                      # L, 75-80
    sp = Ed.multiply(Ed.multiply(p, [8, 0, 0, 0]), order)
    return 1 if on and Ed.is_identity(sp) else 0

