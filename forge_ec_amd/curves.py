"""
Host-side mirror of the forge-ec-core trait surface for the batched scalar-multiplication path.

Names follow the reference (forge-ec-core/src/lib.rs): `Curve::multiply` (832), `Curve::generator`
/ `identity` (784-830), `PointProjective::{add,double,negate,is_identity}` (699-748),
`FieldElement::{add,sub,mul,square,neg}` (173-241).  The batched forms (`batch_multiply*`) are what
this backend adds: each replaces a caller-side loop over `Curve::multiply` (ecdsa.rs:313-361,
schnorr.rs:268-284, core lib.rs:944-948).

Everything runs on the GPU through libfecgpu.so (include/fecgpu.h); there is no CPU path here.
Arrays are numpy uint64, little-endian limbs: scalars/field elements (n,4); points (n,12) for
secp256k1/P-256 (X,Y,Z) and (n,16) for Ed25519 (X,Y,Z,T) -- the reference's `to_raw()` layout.
"""
import ctypes

import numpy as np

from . import _lib as L
from ._lib import ED25519, P256, SECP256K1, FecError


def _u64(a, cols=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.uint64))
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


def _check(rc, what=""):
    if rc != 0:
        raise FecError(rc, what)


class Context:
    """One fec_ctx.  Context(device) = one GPU, one stream pair (use one per process per GPU).
    Context(devices=[...]) = the multi-device ctx of fec_ctx_create_multi: the element-wise host-pointer
    calls shard the batch contiguously over the listed devices (an ordinal may repeat: several shard
    workers on one GPU) and write straight into the caller's output; the *_dev calls are not available."""

    def __init__(self, device=0, devices=None):
        self._lib = L.lib()
        h = ctypes.c_void_p()
        if devices is not None:
            devs = [int(d) for d in devices]
            arr = (ctypes.c_int * len(devs))(*devs)
            _check(self._lib.fec_ctx_create_multi(ctypes.byref(h), arr, len(devs)), "fec_ctx_create_multi")
            self.device = devs[0] if devs else 0
        else:
            _check(self._lib.fec_ctx_create(ctypes.byref(h), int(device)), "fec_ctx_create")
            self.device = int(device)
        self._h = h

    def wipe(self):
        """fec_ctx_wipe: zero every ctx-owned device buffer that can hold copies of caller data."""
        _check(self._lib.fec_ctx_wipe(self._h), "fec_ctx_wipe")

    def check(self):
        """fec_ctx_check: synchronise, then raise FecError(FEC_E_LAUNCH) if a kernel launched through this ctx since
        the last check reported a fault (the outputs of those launches must not be used).  For the *_dev callers;
        the host-pointer calls check by themselves."""
        _check(self._lib.fec_ctx_check(self._h), "fec_ctx_check")

    def debug_force_fault(self, enabled):
        """fec_ctx_debug_force_fault: test hook -- scheduler kernels raise their fault word at once."""
        _check(self._lib.fec_ctx_debug_force_fault(self._h, 1 if enabled else 0), "fec_ctx_debug_force_fault")

    def set_fixed_prefix_bits(self, bits):
        """fec_ctx_set_fixed_prefix_bits: size of the generator's fixed-base prefix tables (0 = off, default 24)."""
        _check(self._lib.fec_ctx_set_fixed_prefix_bits(self._h, int(bits)), "fec_ctx_set_fixed_prefix_bits")

    def build_fixed_prefix(self, curve):
        """fec_ctx_build_fixed_prefix: attach or build the generator's prefix table of `curve` now (synchronous)."""
        _check(self._lib.fec_ctx_build_fixed_prefix(self._h, int(curve)), "fec_ctx_build_fixed_prefix")

    def set_fixed_prefix_after(self, elements):
        """fec_ctx_set_fixed_prefix_after: a ctx left to its defaults builds a table after this many multiplications by G."""
        _check(self._lib.fec_ctx_set_fixed_prefix_after(self._h, int(elements)), "fec_ctx_set_fixed_prefix_after")

    def set_fixed_prefix_budget(self, percent_of_free_memory):
        """fec_ctx_set_fixed_prefix_budget: share of the device's FREE memory a table may take (default 25 %)."""
        _check(self._lib.fec_ctx_set_fixed_prefix_budget(self._h, int(percent_of_free_memory)), "fec_ctx_set_fixed_prefix_budget")

    def set_side_stream_max(self, elements):
        """fec_ctx_set_side_stream_max: u1*G runs beside u2*Q on the second stream up to this many elements."""
        _check(self._lib.fec_ctx_set_side_stream_max(self._h, int(elements)), "fec_ctx_set_side_stream_max")

    def fixed_prefix_bits(self, curve):
        """fec_ctx_fixed_prefix_bits: bits of the prefix table `curve` has now (0 = none)."""
        r = self._lib.fec_ctx_fixed_prefix_bits(self._h, int(curve))
        _check(min(r, 0), "fec_ctx_fixed_prefix_bits")
        return r

    def device_count(self):
        return int(self._lib.fec_ctx_device_count(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fec_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- host-pointer entry points ----
    def batch_mul(self, curve, scalars, points):
        pl = L.POINT_LIMBS[curve]
        s, p = _u64(scalars, 4), _u64(points, pl)
        if s.shape[0] != p.shape[0]:
            raise ValueError("scalars and points differ in length")
        out = np.empty_like(p)
        _check(self._lib.fec_batch_mul(self._h, curve, _ptr(s), _ptr(p), _ptr(out), s.shape[0]), "fec_batch_mul")
        return out

    def batch_mul_fixed(self, curve, scalars, base):
        pl = L.POINT_LIMBS[curve]
        s, b = _u64(scalars, 4), _u64(base).reshape(pl)
        out = np.empty((s.shape[0], pl), dtype=np.uint64)
        _check(self._lib.fec_batch_mul_fixed(self._h, curve, _ptr(s), _ptr(b), _ptr(out), s.shape[0]),
               "fec_batch_mul_fixed")
        return out

    def batch_double_mul(self, curve, u1, u2, q):
        pl = L.POINT_LIMBS[curve]
        a, b, p = _u64(u1, 4), _u64(u2, 4), _u64(q, pl)
        if not (a.shape[0] == b.shape[0] == p.shape[0]):
            raise ValueError("u1, u2 and q differ in length")
        out = np.empty_like(p)
        _check(self._lib.fec_batch_double_mul(self._h, curve, _ptr(a), _ptr(b), _ptr(p), _ptr(out), a.shape[0]),
               "fec_batch_double_mul")
        return out

    def batch_to_affine(self, curve, points):
        """(xy, inf): xy (n,8) affine limbs, inf (n,) uint8 -- Curve::to_affine per element."""
        pl = L.POINT_LIMBS[curve]
        p = _u64(points, pl)
        xy = np.empty((p.shape[0], 8), dtype=np.uint64)
        inf = np.empty(p.shape[0], dtype=np.uint8)
        _check(self._lib.fec_batch_to_affine(self._h, curve, _ptr(p), _ptr(xy), _ptr(inf), p.shape[0]),
               "fec_batch_to_affine")
        return xy, inf

    def multi_scalar_mul(self, curve, scalars, points):
        """Curve::multi_scalar_multiply: sum_i multiply(points[i], scalars[i]) in the reference's order."""
        pl = L.POINT_LIMBS[curve]
        s, p = _u64(scalars, 4), _u64(points, pl)
        if s.shape[0] != p.shape[0]:
            raise ValueError("scalars and points differ in length")
        out = np.empty(pl, dtype=np.uint64)
        _check(self._lib.fec_multi_scalar_mul(self._h, curve, _ptr(s), _ptr(p), _ptr(out), s.shape[0]),
               "fec_multi_scalar_mul")
        return out

    def _ecdsa_verify(self, fn, what, digests, r, s, pk_xy, pk_inf):
        d = np.ascontiguousarray(np.asarray(digests, dtype=np.uint8)).reshape(-1, 32)
        rr, ss, pk = _u64(r, 4), _u64(s, 4), _u64(pk_xy, 8)
        n = d.shape[0]
        if not (rr.shape[0] == ss.shape[0] == pk.shape[0] == n):
            raise ValueError("inputs differ in length")
        inf = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)).reshape(-1) if pk_inf is not None else None
        if inf is not None and inf.shape[0] != n:
            raise ValueError("pk_inf and the signatures differ in length")  # the C side reads n bytes
        out = np.empty(n, dtype=np.uint8)
        _check(fn(self._h, _ptr(d), _ptr(rr), _ptr(ss), _ptr(pk), _ptr(inf), _ptr(out), n), what)
        return out

    def ecdsa_verify_secp256k1(self, digests, r, s, pk_xy, pk_inf=None):
        """Ecdsa::<Secp256k1, D>::verify per signature with the digests supplied (ecdsa.rs:213-281).
        digests (n,32) uint8; r, s (n,4); pk_xy (n,8) raw limbs; pk_inf (n,) uint8 or None.
        Returns (n,) uint8: 1 valid, 0 invalid, 2 = the reference panics."""
        return self._ecdsa_verify(self._lib.fec_ecdsa_verify_secp256k1, "fec_ecdsa_verify_secp256k1", digests, r, s,
                                  pk_xy, pk_inf)

    def ecdsa_verify_p256(self, digests, r, s, pk_xy, pk_inf=None):
        """Ecdsa::<P256, D>::verify per signature, same conventions, in the reference's P-256 scalar
        arithmetic (p256.rs:924-1020, 1409-1432) -- not standard ECDSA: see include/fecgpu.h."""
        return self._ecdsa_verify(self._lib.fec_ecdsa_verify_p256, "fec_ecdsa_verify_p256", digests, r, s, pk_xy,
                                  pk_inf)

    def batch_validate_point(self, curve, xy, inf=None):
        """Curve::validate_point per affine point (x, y, infinity): (n,) uint8, 1 valid / 0 not."""
        p = _u64(xy, 8)
        n = p.shape[0]
        fl = np.ascontiguousarray(np.asarray(inf, dtype=np.uint8)).reshape(-1) if inf is not None else None
        if fl is not None and fl.shape[0] != n:
            raise ValueError("flags and points differ in length")  # the C side reads n bytes
        ok = np.zeros(n, dtype=np.uint8)
        _check(self._lib.fec_batch_validate_point(self._h, curve, _ptr(p), _ptr(fl), _ptr(ok), n), "fec_batch_validate_point")
        return ok

    def batch_ecdh(self, curve, private_keys, pk_xy, pk_inf=None):
        """KeyExchange::derive_shared_secret per element (secp256k1.rs:1884-1904, p256.rs:2281-2312).  Returns
        (secrets (n,32) uint8, status (n,) uint8): 0 Ok, 1 Err(InvalidPublicKey) (P-256), 2 Err (identity).
        NOT FOR PRODUCTION SECRETS: parity mode reproduces reference behaviour; see include/fecgpu.h."""
        kk, pk = _u64(private_keys, 4), _u64(pk_xy, 8)
        n = kk.shape[0]
        if pk.shape[0] != n:
            raise ValueError("inputs differ in length")
        inf = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)).reshape(-1) if pk_inf is not None else None
        if inf is not None and inf.shape[0] != n:
            raise ValueError("pk_inf and the keys differ in length")  # the C side reads n bytes
        out = np.zeros((n, 32), dtype=np.uint8)
        st = np.zeros(n, dtype=np.uint8)
        _check(self._lib.fec_batch_ecdh(self._h, curve, _ptr(kk), _ptr(pk), _ptr(inf), _ptr(out), _ptr(st), n), "fec_batch_ecdh")
        return out, st

    def ecdsa_batch_verify(self, curve, digests, r, s, pk_xy, pk_inf, a):
        """Ecdsa::<C, D>::batch_verify (ecdsa.rs:287-391) for secp256k1 / P-256 with the digests and the weights
        a (n,4) supplied.  Returns (result, detail): result 1 true, 0 false, 2 = the reference panics; detail
        (16,) uint64 = r_sum (12 limbs) and r_scalar_sum (4), zero when the loop returned early."""
        d = np.ascontiguousarray(np.asarray(digests, dtype=np.uint8)).reshape(-1, 32)
        rr, ss, pk, aa = _u64(r, 4), _u64(s, 4), _u64(pk_xy, 8), _u64(a, 4)
        n = d.shape[0]
        if not (rr.shape[0] == ss.shape[0] == pk.shape[0] == aa.shape[0] == n):
            raise ValueError("inputs differ in length")
        inf = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)).reshape(-1) if pk_inf is not None else None
        if inf is not None and inf.shape[0] != n:
            raise ValueError("pk_inf and the signatures differ in length")  # the C side reads n bytes
        res = np.zeros(1, dtype=np.uint8)
        detail = np.zeros(16, dtype=np.uint64)
        _check(self._lib.fec_ecdsa_batch_verify(self._h, curve, _ptr(d), _ptr(rr), _ptr(ss), _ptr(pk), _ptr(inf), _ptr(aa),
                                                n, _ptr(res), _ptr(detail)), "fec_ecdsa_batch_verify")
        return int(res[0]), detail

    def eddsa_verify_ed25519(self, r_xy, r_inf, pk_xy, pk_inf, s, k):
        """Eddsa::<Ed25519, D>::verify / Ed25519::verify from the point computation on (eddsa.rs:174-211,
        430-447): r_xy, pk_xy (n,8) raw limbs; r_inf, pk_inf (n,) uint8 or None; s, k (n,4) scalars
        (k = from_bytes_reduced(hash)).  Returns (n,) uint8: 1 true, 0 false, 2 = the reference panics."""
        rr, pk, ss, kk = _u64(r_xy, 8), _u64(pk_xy, 8), _u64(s, 4), _u64(k, 4)
        n = ss.shape[0]
        if not (rr.shape[0] == pk.shape[0] == kk.shape[0] == n):
            raise ValueError("inputs differ in length")
        flags = []
        for f in (r_inf, pk_inf):
            a = np.ascontiguousarray(np.asarray(f, dtype=np.uint8)).reshape(-1) if f is not None else None
            if a is not None and a.shape[0] != n:
                raise ValueError("flags and signatures differ in length")  # the C side reads n bytes
            flags.append(a)
        out = np.empty(n, dtype=np.uint8)
        _check(self._lib.fec_eddsa_verify_ed25519(self._h, _ptr(rr), _ptr(flags[0]), _ptr(pk), _ptr(flags[1]), _ptr(ss),
                                                  _ptr(kk), _ptr(out), n), "fec_eddsa_verify_ed25519")
        return out

    def batch_compress(self, curve, xy, inf=None):
        """PointAffine::to_bytes of each affine point (x, y, infinity) -> (n, 33) uint8."""
        p = _u64(xy, 8)
        n = p.shape[0]
        fl = np.ascontiguousarray(np.asarray(inf, dtype=np.uint8)) if inf is not None else None
        if fl is not None and fl.shape[0] != n:
            raise ValueError("flags and points differ in length")
        out = np.zeros((n, 33), dtype=np.uint8)
        _check(self._lib.fec_batch_compress(self._h, curve, _ptr(p), _ptr(fl), _ptr(out), n), "fec_batch_compress")
        return out

    def _decode(self, fn, what, curve, data, width):
        b = np.ascontiguousarray(np.asarray(data, dtype=np.uint8)).reshape(-1, width)
        n = b.shape[0]
        xy = np.zeros((n, 8), dtype=np.uint64)
        inf, ok = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        _check(fn(self._h, curve, _ptr(b), _ptr(xy), _ptr(inf), _ptr(ok), n), what)
        return xy, inf, ok

    def batch_decompress(self, curve, data33):
        """PointAffine::from_bytes of each 33-byte encoding -> (xy (n,8), infinity (n,), ok (n,)); ok = 0 where
        the reference returns None."""
        return self._decode(self._lib.fec_batch_decompress, "fec_batch_decompress", curve, data33, 33)

    def batch_decode_uncompressed(self, curve, data65):
        """forge-ec-encoding UncompressedPoint::to_affine of each 65-byte encoding."""
        return self._decode(self._lib.fec_batch_decode_uncompressed, "fec_batch_decode_uncompressed", curve, data65, 65)

    def batch_encode_uncompressed(self, curve, xy, inf=None):
        """UncompressedPoint::from_affine -> (n, 65) uint8."""
        p = _u64(xy, 8)
        n = p.shape[0]
        fl = np.ascontiguousarray(np.asarray(inf, dtype=np.uint8)).reshape(-1) if inf is not None else None
        if fl is not None and fl.shape[0] != n:
            raise ValueError("flags and points differ in length")
        out = np.zeros((n, 65), dtype=np.uint8)
        _check(self._lib.fec_batch_encode_uncompressed(self._h, curve, _ptr(p), _ptr(fl), _ptr(out), n),
               "fec_batch_encode_uncompressed")
        return out

    def schnorr_batch_verify_secp256k1(self, pk_xy, r_xy, s, a, e, pk_inf=None, r_inf=None):
        """schnorr::batch_verify::<Secp256k1, D> (schnorr.rs:194-290), challenges e and weights a supplied.
        -> (result bool, sides (16,) uint64 = x,y of both affine sums, sides_inf (2,) uint8)."""
        pk, rr = _u64(pk_xy, 8), _u64(r_xy, 8)
        ss, aa, ee = _u64(s, 4), _u64(a, 4), _u64(e, 4)
        n = ss.shape[0]
        if not (pk.shape[0] == rr.shape[0] == aa.shape[0] == ee.shape[0] == n):
            raise ValueError("inputs differ in length")
        pi = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)).reshape(-1) if pk_inf is not None else None
        ri = np.ascontiguousarray(np.asarray(r_inf, dtype=np.uint8)).reshape(-1) if r_inf is not None else None
        for flags in (pi, ri):
            if flags is not None and flags.shape[0] != n:
                raise ValueError("infinity flags and the signatures differ in length")  # the C side reads n bytes
        res = np.zeros(1, dtype=np.uint8)
        sides = np.zeros(16, dtype=np.uint64)
        sinf = np.zeros(2, dtype=np.uint8)
        _check(self._lib.fec_schnorr_batch_verify_secp256k1(self._h, _ptr(pk), _ptr(pi), _ptr(rr), _ptr(ri), _ptr(ss),
                                                            _ptr(aa), _ptr(ee), n, _ptr(res), _ptr(sides), _ptr(sinf)),
               "fec_schnorr_batch_verify_secp256k1")
        return bool(res[0]), sides, sinf

    def schnorr_batch_verify(self, curve, pk_xy, r_xy, s, a, e, pk_inf=None, r_inf=None):
        """schnorr::batch_verify::<C, D> (schnorr.rs:194-290) for any curve; as the secp256k1 form.  (ED25519: the release
        profile's behaviour; schnorr_batch_verify_ed25519 also says whether a debug build would have panicked.)"""
        pk, rr = _u64(pk_xy, 8), _u64(r_xy, 8)
        ss, aa, ee = _u64(s, 4), _u64(a, 4), _u64(e, 4)
        n = ss.shape[0]
        if not (pk.shape[0] == rr.shape[0] == aa.shape[0] == ee.shape[0] == n):
            raise ValueError("inputs differ in length")
        pi = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)).reshape(-1) if pk_inf is not None else None
        ri = np.ascontiguousarray(np.asarray(r_inf, dtype=np.uint8)).reshape(-1) if r_inf is not None else None
        for flags in (pi, ri):
            if flags is not None and flags.shape[0] != n:
                raise ValueError("infinity flags and the signatures differ in length")
        res = np.zeros(1, dtype=np.uint8)
        sides = np.zeros(16, dtype=np.uint64)
        sinf = np.zeros(2, dtype=np.uint8)
        _check(self._lib.fec_schnorr_batch_verify(self._h, curve, _ptr(pk), _ptr(pi), _ptr(rr), _ptr(ri), _ptr(ss), _ptr(aa),
                                                  _ptr(ee), n, _ptr(res), _ptr(sides), _ptr(sinf)), "fec_schnorr_batch_verify")
        return bool(res[0]), sides, sinf

    def schnorr_batch_verify_ed25519(self, pk_xy, r_xy, s, a, e, pk_inf=None, r_inf=None):
        """fec_schnorr_batch_verify_ed25519: schnorr::batch_verify::<Ed25519, D> with the scalar Mul as the reference's
        RELEASE profile runs it (u128 sums wrap).  -> (result 0 / 1 / 2 = the reference panics in to_affine, sides (16,),
        sides_inf (2,), debug_build_panics: a debug build panics on these inputs instead)."""
        pk, rr = _u64(pk_xy, 8), _u64(r_xy, 8)
        ss, aa, ee = _u64(s, 4), _u64(a, 4), _u64(e, 4)
        n = ss.shape[0]
        if not (pk.shape[0] == rr.shape[0] == aa.shape[0] == ee.shape[0] == n):
            raise ValueError("inputs differ in length")
        pi = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)).reshape(-1) if pk_inf is not None else None
        ri = np.ascontiguousarray(np.asarray(r_inf, dtype=np.uint8)).reshape(-1) if r_inf is not None else None
        for flags in (pi, ri):
            if flags is not None and flags.shape[0] != n:
                raise ValueError("infinity flags and the signatures differ in length")
        res = np.zeros(1, dtype=np.uint8)
        sides = np.zeros(16, dtype=np.uint64)
        sinf = np.zeros(2, dtype=np.uint8)
        dbg = np.zeros(1, dtype=np.uint8)
        _check(self._lib.fec_schnorr_batch_verify_ed25519(self._h, _ptr(pk), _ptr(pi), _ptr(rr), _ptr(ri), _ptr(ss), _ptr(aa),
                                                          _ptr(ee), n, _ptr(res), _ptr(sides), _ptr(sinf), _ptr(dbg)),
               "fec_schnorr_batch_verify_ed25519")
        return int(res[0]), sides, sinf, bool(dbg[0])

    def schnorr_verify(self, curve, pk_xy, r_xy, s, e, pk_inf=None, r_inf=None):
        """Schnorr::<C, D>::verify per signature (schnorr.rs:90-140) from the point computation on, the challenges
        e = from_bytes_reduced(hash) supplied: (n,) uint8 -- 1 true, 0 false, 2 = the reference panics."""
        pk, rr, ss, ee = _u64(pk_xy, 8), _u64(r_xy, 8), _u64(s, 4), _u64(e, 4)
        n = ss.shape[0]
        if not (pk.shape[0] == rr.shape[0] == ee.shape[0] == n):
            raise ValueError("inputs differ in length")
        pi = np.ascontiguousarray(np.asarray(pk_inf, dtype=np.uint8)).reshape(-1) if pk_inf is not None else None
        ri = np.ascontiguousarray(np.asarray(r_inf, dtype=np.uint8)).reshape(-1) if r_inf is not None else None
        for flags in (pi, ri):
            if flags is not None and flags.shape[0] != n:
                raise ValueError("infinity flags and the signatures differ in length")
        out = np.empty(n, dtype=np.uint8)
        _check(self._lib.fec_schnorr_verify(self._h, curve, _ptr(pk), _ptr(pi), _ptr(rr), _ptr(ri), _ptr(ss), _ptr(ee),
                                            _ptr(out), n), "fec_schnorr_verify")
        return out

    def schnorr_verify_dev(self, curve, d_pk_xy, d_pk_inf, d_r_xy, d_r_inf, d_s, d_e, d_status, n, stream=None):
        _check(self._lib.fec_schnorr_verify_dev(self._h, curve, d_pk_xy, d_pk_inf, d_r_xy, d_r_inf, d_s, d_e, d_status, n,
                                                stream), "fec_schnorr_verify_dev")

    def field_op(self, curve, op, a, b=None):
        x = _u64(a, 4)
        y = _u64(b, 4) if b is not None else None
        if y is not None and y.shape != x.shape:
            raise ValueError("operands differ in shape")
        out = np.empty_like(x)
        _check(self._lib.fec_field_op(self._h, curve, op, _ptr(x), _ptr(y), _ptr(out), x.shape[0]), "fec_field_op")
        return out

    def point_op(self, curve, op, p, q=None):
        pl = L.POINT_LIMBS[curve]
        x = _u64(p, pl)
        y = _u64(q, pl) if q is not None else None
        if y is not None and y.shape != x.shape:
            raise ValueError("operands differ in shape")
        out = np.empty_like(x)
        _check(self._lib.fec_point_op(self._h, curve, op, _ptr(x), _ptr(y), _ptr(out), x.shape[0]), "fec_point_op")
        return out

    # ---- device-pointer entry points (raw addresses, e.g. torch.Tensor.data_ptr()) ----
    def batch_mul_dev(self, curve, d_scalars, d_points, d_out, n, stream=None):
        _check(self._lib.fec_batch_mul_dev(self._h, curve, d_scalars, d_points, d_out, n, stream), "fec_batch_mul_dev")

    def batch_mul_fixed_dev(self, curve, d_scalars, d_base, d_out, n, stream=None):
        _check(self._lib.fec_batch_mul_fixed_dev(self._h, curve, d_scalars, d_base, d_out, n, stream),
               "fec_batch_mul_fixed_dev")

    def batch_double_mul_dev(self, curve, d_u1, d_u2, d_q, d_out, n, stream=None):
        _check(self._lib.fec_batch_double_mul_dev(self._h, curve, d_u1, d_u2, d_q, d_out, n, stream),
               "fec_batch_double_mul_dev")

    # ---- device-resident shards of a multi-device ctx (fec_multi_batch_*_dev): lists with one raw device pointer /
    # count per shard worker; `gathered` a raw device pointer on the consumer-th device or None ----
    @staticmethod
    def _ptr_array(ptrs, n):
        if ptrs is None:
            return None
        if len(ptrs) != n:
            raise ValueError("one entry per device of the ctx")
        return (ctypes.c_void_p * n)(*[ctypes.c_void_p(int(p) if p else 0) for p in ptrs])

    def _multi_dev(self, fn, what, curve, inputs, d_out, counts, gathered, consumer, streams):
        n = self.device_count()
        if len(counts) != n:
            raise ValueError("one count per device of the ctx")
        cnt = (ctypes.c_size_t * n)(*[int(c) for c in counts])
        args = [self._ptr_array(a, n) for a in inputs] + [self._ptr_array(d_out, n), cnt,
                                                          ctypes.c_void_p(int(gathered)) if gathered else None, int(consumer),
                                                          self._ptr_array(streams, n)]
        _check(fn(self._h, int(curve), *args), what)

    def multi_batch_mul_dev(self, curve, d_scalars, d_points, d_out, counts, gathered=None, consumer=0, streams=None):
        self._multi_dev(self._lib.fec_multi_batch_mul_dev, "fec_multi_batch_mul_dev", curve, [d_scalars, d_points], d_out,
                        counts, gathered, consumer, streams)

    def multi_batch_mul_fixed_dev(self, curve, d_scalars, d_bases, d_out, counts, gathered=None, consumer=0, streams=None):
        self._multi_dev(self._lib.fec_multi_batch_mul_fixed_dev, "fec_multi_batch_mul_fixed_dev", curve, [d_scalars, d_bases],
                        d_out, counts, gathered, consumer, streams)

    def multi_batch_double_mul_dev(self, curve, d_u1, d_u2, d_q, d_out, counts, gathered=None, consumer=0, streams=None):
        self._multi_dev(self._lib.fec_multi_batch_double_mul_dev, "fec_multi_batch_double_mul_dev", curve, [d_u1, d_u2, d_q],
                        d_out, counts, gathered, consumer, streams)

    def batch_to_affine_dev(self, curve, d_points, d_xy, d_inf, n, stream=None):
        _check(self._lib.fec_batch_to_affine_dev(self._h, curve, d_points, d_xy, d_inf, n, stream),
               "fec_batch_to_affine_dev")

    def batch_compress_dev(self, curve, d_xy, d_inf, d_out, n, stream=None):
        _check(self._lib.fec_batch_compress_dev(self._h, curve, d_xy, d_inf, d_out, n, stream), "fec_batch_compress_dev")

    def ecdsa_verify_secp256k1_dev(self, d_digests, d_r, d_s, d_pk_xy, d_pk_inf, d_status, n, stream=None):
        _check(self._lib.fec_ecdsa_verify_secp256k1_dev(self._h, d_digests, d_r, d_s, d_pk_xy, d_pk_inf, d_status, n,
                                                        stream), "fec_ecdsa_verify_secp256k1_dev")

    def batch_validate_point_dev(self, curve, d_xy, d_inf, d_ok, n, stream=None):
        _check(self._lib.fec_batch_validate_point_dev(self._h, curve, d_xy, d_inf, d_ok, n, stream), "fec_batch_validate_point_dev")

    def batch_ecdh_dev(self, curve, d_private_keys, d_pk_xy, d_pk_inf, d_secrets, d_status, n, stream=None):
        _check(self._lib.fec_batch_ecdh_dev(self._h, curve, d_private_keys, d_pk_xy, d_pk_inf, d_secrets, d_status, n, stream),
               "fec_batch_ecdh_dev")

    def eddsa_verify_ed25519_dev(self, d_r_xy, d_r_inf, d_pk_xy, d_pk_inf, d_s, d_k, d_status, n, stream=None):
        _check(self._lib.fec_eddsa_verify_ed25519_dev(self._h, d_r_xy, d_r_inf, d_pk_xy, d_pk_inf, d_s, d_k, d_status, n,
                                                      stream), "fec_eddsa_verify_ed25519_dev")

    def ecdsa_verify_p256_dev(self, d_digests, d_r, d_s, d_pk_xy, d_pk_inf, d_status, n, stream=None):
        _check(self._lib.fec_ecdsa_verify_p256_dev(self._h, d_digests, d_r, d_s, d_pk_xy, d_pk_inf, d_status, n,
                                                   stream), "fec_ecdsa_verify_p256_dev")

    def generator(self, curve):
        out = np.empty(L.POINT_LIMBS[curve], dtype=np.uint64)
        _check(self._lib.fec_generator(self._h, curve, _ptr(out)), "fec_generator")
        return out

    def generator_dev(self, curve):
        """Device address of the ctx's generator (pass it to batch_mul_fixed_dev)."""
        return self._lib.fec_generator_dev(self._h, curve)

    def set_chunk(self, elements):
        """Elements per pipeline chunk of the host-pointer entry points (default 2^18)."""
        _check(self._lib.fec_ctx_set_chunk(self._h, int(elements)))

    # ---- measurement ----
    def set_timing(self, enabled=True):
        _check(self._lib.fec_ctx_set_timing(self._h, 1 if enabled else 0))

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        name = ctypes.c_char_p()
        _check(self._lib.fec_ctx_last_kernel_ms(self._h, ctypes.byref(ms), ctypes.byref(name)), "last_kernel_ms")
        return float(ms.value), (name.value or b"").decode()

    def measure_peak_mad32(self):
        v = ctypes.c_double()
        _check(self._lib.fec_measure_peak_mad32(self._h, ctypes.byref(v)), "fec_measure_peak_mad32")
        return float(v.value)

    def device_info(self):
        buf = ctypes.create_string_buffer(256)
        cus, khz = ctypes.c_int(), ctypes.c_int()
        _check(self._lib.fec_ctx_device_info(self._h, buf, 256, ctypes.byref(cus), ctypes.byref(khz)))
        return {"name": buf.value.decode(), "compute_units": cus.value, "clock_khz": khz.value}


class _Curve:
    """Common `Curve` trait surface.  Subclasses fix ID, NAME and the reference's constants."""
    ID = None
    NAME = None
    POINT_LIMBS = 12
    _IDENTITY = None

    def __init__(self, ctx=None, device=0):
        self.ctx = ctx if ctx is not None else Context(device)

    # Curve::identity / generator (as the reference builds them, raw limbs)
    @classmethod
    def identity(cls):
        return np.array(cls._IDENTITY, dtype=np.uint64)

    def generator(self):
        return self.ctx.generator(self.ID)

    # Curve::multiply
    def multiply(self, point, scalar):
        return self.ctx.batch_mul(self.ID, _u64(scalar).reshape(1, 4), _u64(point).reshape(1, -1))[0]

    # batched forms
    def batch_multiply(self, points, scalars):
        return self.ctx.batch_mul(self.ID, scalars, points)

    def batch_multiply_fixed(self, base, scalars):
        return self.ctx.batch_mul_fixed(self.ID, scalars, base)

    def batch_double_multiply(self, u1, u2, q):
        """R[i] = multiply(G, u1[i]) + multiply(q[i], u2[i])  (ecdsa.rs:254-256)."""
        return self.ctx.batch_double_mul(self.ID, u1, u2, q)

    def multi_scalar_multiply(self, points, scalars):
        """Curve::multi_scalar_multiply (core lib.rs:934-951): identity for empty or mismatched input."""
        pts, ks = _u64(points, self.POINT_LIMBS), _u64(scalars, 4)
        if pts.shape[0] != ks.shape[0] or pts.shape[0] == 0:
            return self.identity()
        return self.ctx.multi_scalar_mul(self.ID, ks, pts)

    # Curve::to_affine, batched
    def batch_to_affine(self, points):
        return self.ctx.batch_to_affine(self.ID, points)

    # PointProjective
    def add(self, p, q):
        return self.ctx.point_op(self.ID, L.P_ADD, p, q)

    def double(self, p):
        return self.ctx.point_op(self.ID, L.P_DOUBLE, p)

    def negate(self, p):
        return self.ctx.point_op(self.ID, L.P_NEGATE, p)

    # FieldElement
    def fe_add(self, a, b):
        return self.ctx.field_op(self.ID, L.F_ADD, a, b)

    def fe_sub(self, a, b):
        return self.ctx.field_op(self.ID, L.F_SUB, a, b)

    def fe_mul(self, a, b):
        return self.ctx.field_op(self.ID, L.F_MUL, a, b)

    def fe_square(self, a):
        return self.ctx.field_op(self.ID, L.F_SQR, a)

    def fe_neg(self, a):
        return self.ctx.field_op(self.ID, L.F_NEG, a)


class Secp256k1(_Curve):
    ID, NAME, POINT_LIMBS = SECP256K1, "secp256k1", 12
    _IDENTITY = [0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0]  # secp256k1.rs:1322-1324

    def double_trait(self, p):
        """trait PointProjective::double (secp256k1.rs:1375-1418), not the ladder's inherent one."""
        return self.ctx.point_op(self.ID, L.P_DOUBLE_TRAIT, p)


class P256Curve(_Curve):
    ID, NAME, POINT_LIMBS = P256, "p256", 12
    _IDENTITY = [0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0]  # p256.rs:1827-1829


class Ed25519Curve(_Curve):
    ID, NAME, POINT_LIMBS = ED25519, "ed25519", 16
    _IDENTITY = [0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0]  # ed25519.rs:1776-1783


CURVES = {SECP256K1: Secp256k1, P256: P256Curve, ED25519: Ed25519Curve}
