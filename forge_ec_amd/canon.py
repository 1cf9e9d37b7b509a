"""
Canonical-math mode over include/fecgpu_canon.h -- NOT reference parity.

The real secp256k1 / P-256 / Ed25519 groups (standard public keys / ECDH points), for callers who want results
other libraries agree with; forge-ec's own arithmetic does not produce them.
Arrays are numpy uint64 little-endian limbs of the plain integers: scalars (n,4); affine points
(n,8) = x then y; status (n,) uint8: 0 finite, 1 infinity (xy = 0), 2 input point rejected.
GPU only, like the rest of the package.
"""
import numpy as np

from . import _lib as L
from .curves import Context, _check, _ptr, _u64

FINITE, INFINITY, BAD_POINT = 0, 1, 2


class CanonCurve:
    CURVE = None

    def __init__(self, ctx=None, device=0):
        self.ctx = ctx if ctx is not None else Context(device)
        self._lib = self.ctx._lib
        self._h = self.ctx._h

    def mul_base(self, scalars):
        """(xy, status) with xy[i] = scalars[i] * G."""
        k = _u64(scalars, 4)
        n = k.shape[0]
        xy = np.zeros((n, 8), dtype=np.uint64)
        st = np.zeros(n, dtype=np.uint8)
        _check(self._lib.fec_canon_mul_base(self._h, self.CURVE, _ptr(k), _ptr(xy), _ptr(st), n), "fec_canon_mul_base")
        return xy, st

    def mul(self, scalars, points_xy):
        """(xy, status) with xy[i] = scalars[i] * points_xy[i]; off-curve inputs get status 2."""
        k = _u64(scalars, 4)
        p = _u64(points_xy, 8)
        if p.shape[0] != k.shape[0]:
            raise ValueError("scalars and points differ in length")
        n = k.shape[0]
        xy = np.zeros((n, 8), dtype=np.uint64)
        st = np.zeros(n, dtype=np.uint8)
        _check(self._lib.fec_canon_mul(self._h, self.CURVE, _ptr(k), _ptr(p), _ptr(xy), _ptr(st), n), "fec_canon_mul")
        return xy, st

    def double_mul(self, u1, u2, points_xy):
        """(xy, status) with xy[i] = u1[i] * G + u2[i] * points_xy[i]  (signature-verification point)."""
        a, b = _u64(u1, 4), _u64(u2, 4)
        p = _u64(points_xy, 8)
        if not (a.shape[0] == b.shape[0] == p.shape[0]):
            raise ValueError("inputs differ in length")
        n = a.shape[0]
        xy = np.zeros((n, 8), dtype=np.uint64)
        st = np.zeros(n, dtype=np.uint8)
        _check(self._lib.fec_canon_double_mul(self._h, self.CURVE, _ptr(a), _ptr(b), _ptr(p), _ptr(xy), _ptr(st), n),
               "fec_canon_double_mul")
        return xy, st

    def double_mul_dev(self, d_u1, d_u2, d_points_xy, d_out_xy, d_status, n, stream=None):
        _check(self._lib.fec_canon_double_mul_dev(self._h, self.CURVE, d_u1, d_u2, d_points_xy, d_out_xy, d_status, n,
                                                  stream), "fec_canon_double_mul_dev")

    def ecdsa_verify(self, z, r, s, pk_xy):
        """Standard ECDSA verification (secp256k1, P-256): (n,) uint8, 1 = valid.  z = digest as an integer."""
        zz, rr, ss = _u64(z, 4), _u64(r, 4), _u64(s, 4)
        pk = _u64(pk_xy, 8)
        n = zz.shape[0]
        if not (rr.shape[0] == ss.shape[0] == pk.shape[0] == n):
            raise ValueError("inputs differ in length")
        out = np.zeros(n, dtype=np.uint8)
        _check(self._lib.fec_canon_ecdsa_verify(self._h, self.CURVE, _ptr(zz), _ptr(rr), _ptr(ss), _ptr(pk), _ptr(out), n),
               "fec_canon_ecdsa_verify")
        return out

    def ecdsa_verify_dev(self, d_z, d_r, d_s, d_pk_xy, d_result, n, stream=None):
        _check(self._lib.fec_canon_ecdsa_verify_dev(self._h, self.CURVE, d_z, d_r, d_s, d_pk_xy, d_result, n, stream),
               "fec_canon_ecdsa_verify_dev")

    def scalar_muladd(self, a, b, c):
        """a * b + c modulo the group order, element-wise, any 256-bit inputs."""
        aa, bb, cc = _u64(a, 4), _u64(b, 4), _u64(c, 4)
        if not (aa.shape == bb.shape == cc.shape):
            raise ValueError("operands differ in shape")
        out = np.empty_like(aa)
        _check(self._lib.fec_canon_scalar_op(self._h, self.CURVE, 0, _ptr(aa), _ptr(bb), _ptr(cc), _ptr(out), aa.shape[0]),
               "fec_canon_scalar_op")
        return out

    def scalar_inv(self, a):
        """a^-1 modulo the group order (0 for a = 0)."""
        aa = _u64(a, 4)
        out = np.empty_like(aa)
        _check(self._lib.fec_canon_scalar_op(self._h, self.CURVE, 1, _ptr(aa), None, None, _ptr(out), aa.shape[0]),
               "fec_canon_scalar_op")
        return out

    def ecdsa_sign(self, z, d, k):
        """ECDSA signing with caller-supplied nonces k (e.g. RFC 6979): r = x(k G) mod n, s = k^-1 (z + r d).
        -> (r, s, ok) with ok[i] = 0 where r or s came out 0 (the caller picks another nonce).  All arithmetic
        on the GPU: one comb pass and three scalar-field passes.

        NOT FOR PRODUCTION SECRETS: the comb indexes its table by digits of k, skips zero digits and takes
        wave-level branches on exceptional cases (include/fecgpu_canon.h) -- the nonce and the key are not
        handled in constant time.  This helper exists to check the arithmetic against RFC 6979 / FIPS
        vectors (test keys).  The ctx staging that held k and d is zeroed before returning."""
        kk = _u64(k, 4)
        one = np.zeros_like(kk)
        one[:, 0] = 1
        zero = np.zeros_like(kk)
        xy, st = self.mul_base(kk)
        r = self.scalar_muladd(xy[:, :4], one, zero)                       # x mod n
        s = self.scalar_muladd(self.scalar_inv(kk), self.scalar_muladd(r, d, z), zero)
        ok = ((st == 0) & r.any(axis=1) & s.any(axis=1)).astype(np.uint8)
        self._lib.fec_ctx_wipe(self._h)
        return r, s, ok

    def mul_base_dev(self, d_scalars, d_out_xy, d_status, n, stream=None):
        _check(self._lib.fec_canon_mul_base_dev(self._h, self.CURVE, d_scalars, d_out_xy, d_status, n, stream),
               "fec_canon_mul_base_dev")

    def mul_dev(self, d_scalars, d_points_xy, d_out_xy, d_status, n, stream=None):
        _check(self._lib.fec_canon_mul_dev(self._h, self.CURVE, d_scalars, d_points_xy, d_out_xy, d_status, n, stream),
               "fec_canon_mul_dev")

    def field_op(self, op, a, b=None):
        x = _u64(a, 4)
        y = _u64(b, 4) if b is not None else None
        if y is not None and y.shape != x.shape:
            raise ValueError("operands differ in shape")
        out = np.empty_like(x)
        _check(self._lib.fec_canon_field_op(self._h, self.CURVE, op, _ptr(x), _ptr(y), _ptr(out), x.shape[0]),
               "fec_canon_field_op")
        return out


def _four(lib_fn, h, a, b, c, d, what):
    aa, bb, cc, dd = _u64(a, 4), _u64(b, 4), _u64(c, 4), _u64(d, 4)
    n = aa.shape[0]
    if not (bb.shape[0] == cc.shape[0] == dd.shape[0] == n):
        raise ValueError("inputs differ in length")
    out = np.zeros(n, dtype=np.uint8)
    _check(lib_fn(h, _ptr(aa), _ptr(bb), _ptr(cc), _ptr(dd), _ptr(out), n), what)
    return out


class CanonSecp256k1(CanonCurve):
    CURVE = L.SECP256K1

    def bip340_verify(self, pk_x, r, s, e):
        """BIP-340 Schnorr verification; e = int(tagged_hash("BIP0340/challenge", r || pk || m)).  (n,) uint8."""
        return _four(self._lib.fec_canon_bip340_verify, self._h, pk_x, r, s, e, "fec_canon_bip340_verify")

    def bip340_verify_dev(self, d_pk_x, d_r, d_s, d_e, d_result, n, stream=None):
        _check(self._lib.fec_canon_bip340_verify_dev(self._h, d_pk_x, d_r, d_s, d_e, d_result, n, stream),
               "fec_canon_bip340_verify_dev")


class CanonP256(CanonCurve):
    CURVE = L.P256


class CanonEd25519(CanonCurve):
    """Ed25519 (RFC 8032): affine (x, y) of k*B / k*P; there is no point at infinity, status is 0 or 2."""
    CURVE = L.ED25519

    def eddsa_verify(self, a_enc, r_enc, s, h):
        """RFC 8032 verification; encodings as little-endian 256-bit integers, h = SHA-512(R||A||M) mod l."""
        return _four(self._lib.fec_canon_eddsa_verify, self._h, a_enc, r_enc, s, h, "fec_canon_eddsa_verify")

    def eddsa_sign_finish(self, h, a, r):
        """second half of RFC 8032 signing: S = h * a + r (mod l); the first half is R = mul_base(r).
        NOT FOR PRODUCTION SECRETS (see ecdsa_sign): vector-checking only; the ctx staging is zeroed after."""
        out = self.scalar_muladd(h, a, r)
        self._lib.fec_ctx_wipe(self._h)
        return out

    def eddsa_verify_dev(self, d_a_enc, d_r_enc, d_s, d_h, d_result, n, stream=None):
        _check(self._lib.fec_canon_eddsa_verify_dev(self._h, d_a_enc, d_r_enc, d_s, d_h, d_result, n, stream),
               "fec_canon_eddsa_verify_dev")


CANON_CURVES = {"secp256k1": CanonSecp256k1, "p256": CanonP256, "ed25519": CanonEd25519}
